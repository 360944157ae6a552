#!/usr/bin/env python3
"""Match leg alone (1M x 1M by default), for A/B runs of kernel variants (SIFT_HIP_LIB=<lib>):
    python tools/match_rate.py [n] [runs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sift_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = sift_amd.SiftHip(0)
q, r = sift_amd.planted_descriptors(n, n)
dq, dr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
best = torch.empty(n, dtype=torch.int32, device="cuda")
d1 = torch.empty_like(best)
d2 = torch.empty_like(best)
ctx.match_u8_dev(dq, n, dr, n, 0.5, 0, best, d1, d2)
ctx.synchronize()
res = []
for _ in range(runs):
    t0 = time.perf_counter()
    ctx.match_u8_dev(dq, n, dr, n, 0.5, 0, best, d1, d2)
    ctx.synchronize()
    res.append(time.perf_counter() - t0)
t = min(res)
print("%s: %d x %d: %s ms -> %.1f Gpairs/s = %.3f of 5 POP/s, accepted %d, checksum %d" % (
    os.path.basename(os.environ.get("SIFT_HIP_LIB", "libsift_hip.so")), n, n, " ".join("%.2f" % (x * 1e3) for x in res),
    n * float(n) / t / 1e9, n * float(n) * 256 / t / 5e15, int((best >= 0).sum()), int(d1.long().sum() % 1000003)), flush=True)
