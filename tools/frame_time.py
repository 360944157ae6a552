#!/usr/bin/env python3
"""Wall time per frame of sift_hip_extract_dev (frame resident, page-locked feature array), for A/B runs of
environment switches:  python tools/frame_time.py [--width W --height H --octaves N --blur-mode M --steps K]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--octaves", type=int, default=4)
ap.add_argument("--blur-mode", type=int, default=0)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
ctx = sift_amd.SiftHip(0)
d = torch.from_numpy(sift_amd.blob_image(a.width, a.height, seed=1)).cuda()
out = ctx.pinned_array(1 << 18, sift_amd.FEAT_DTYPE)
prm = SiftParams(octaves=a.octaves, blur_mode=a.blur_mode)
for _ in range(25):
    n = ctx.extract_dev(d, a.width, a.height, prm, out)
res = []
for _ in range(a.reps):
    t0 = time.perf_counter()
    for _ in range(a.steps):
        n = ctx.extract_dev(d, a.width, a.height, prm, out)
    res.append((time.perf_counter() - t0) * 1e3 / a.steps)
print("%dx%d octaves %d blur_mode %d: %d features, ms per frame %s" % (
    a.width, a.height, a.octaves, a.blur_mode, n, " ".join("%.3f" % r for r in res)))
