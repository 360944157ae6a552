#!/usr/bin/env python3
"""Runs the two-stream extract with extra streams alive in the process (an RCCL
communicator, several torch streams with work): more streams than hardware queues.
Prints progress; meant to be run under `timeout`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
t = torch.ones(1 << 20, device="cuda")
dist.all_reduce(t)
streams = [torch.cuda.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4)]
ctx = sift_amd.SiftHip(0)
W, H = 1920, 1080
frame = sift_amd.blob_image(W, H, seed=1)
d = torch.from_numpy(frame).cuda()
out = np.zeros(1 << 15, sift_amd.FEAT_DTYPE)
prm = SiftParams(octaves=4)
t0 = time.time()
for it in range(600):
    for s in streams:
        with torch.cuda.stream(s):
            t.mul_(1.0001)
    if it % 50 == 0:
        dist.all_reduce(t)
    n = ctx.extract_dev(d, W, H, prm, out)
    if it % 100 == 0:
        print("iter", it, "features", n, "t=%.1fs" % (time.time() - t0), flush=True)
torch.cuda.synchronize()
print("ok", time.time() - t0, flush=True)
dist.destroy_process_group()
