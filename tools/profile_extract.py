#!/usr/bin/env python3
"""Runs a few 4K extracts (and optionally a match) so that rocprofv3 can trace the kernels:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/profile_extract.py
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--octaves", type=int, default=4)
ap.add_argument("--blur-mode", type=int, default=0)
ap.add_argument("--match-n", type=int, default=0)
ap.add_argument("--pinned", action="store_true", help="feature array in page-locked memory")
a = ap.parse_args()
ctx = sift_amd.SiftHip(0)
frame = sift_amd.blob_image(a.width, a.height, seed=1)
d = torch.from_numpy(frame).cuda()
out = ctx.pinned_array(1 << 16, sift_amd.FEAT_DTYPE) if a.pinned else np.zeros(1 << 16, sift_amd.FEAT_DTYPE)
n = 0
for _ in range(a.steps):
    n = ctx.extract_dev(d, a.width, a.height, SiftParams(octaves=a.octaves, blur_mode=a.blur_mode), out)
print("features", n)
if a.match_n:
    q, r = sift_amd.planted_descriptors(a.match_n, a.match_n)
    dq, dr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    best = torch.empty(a.match_n, dtype=torch.int32, device="cuda")
    for _ in range(2):
        ctx.match_u8_dev(dq, a.match_n, dr, a.match_n, 0.5, 0, best)
        ctx.synchronize()
    print("accepted", int((best >= 0).sum()))
