#!/usr/bin/env python3
"""Wall time of single 4K extracts (and of a pipelined batch) under values of one launch-shape switch, interleaved so
that a drift of the box shows:  python tools/tuning_ab.py blur_rstream -1 0 [--rounds 3]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("key")
ap.add_argument("values", type=int, nargs="+")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--frames", type=int, default=100)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--blur-mode", type=int, default=0)
a = ap.parse_args()
ctx = sift_amd.SiftHip(0)
W, H = a.width, a.height
d = torch.from_numpy(sift_amd.blob_image(W, H, seed=1)).cuda()
out = ctx.pinned_array(1 << 17, sift_amd.FEAT_DTYPE)
prm = SiftParams(octaves=4, blur_mode=a.blur_mode)
res = {v: [] for v in a.values}
nfeat = {}
for r in range(a.rounds):
    for v in a.values:
        ctx.set_tuning(a.key, v)
        for _ in range(10):
            n = ctx.extract_dev(d, W, H, prm, out)
        nfeat[v] = n if isinstance(n, int) else len(n)
        ts = []
        for _ in range(a.frames):
            t0 = time.perf_counter()
            ctx.extract_dev(d, W, H, prm, out)
            ts.append((time.perf_counter() - t0) * 1e3)
        res[v].append(float(np.median(ts)))
for v in a.values:
    print("%s = %d: median ms per frame by round %s, features %s" % (a.key, v, " ".join("%.3f" % x for x in res[v]), nfeat[v]), flush=True)
