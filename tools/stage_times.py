#!/usr/bin/env python3
"""Per-kernel-class time of one extract (event-bracketed, serialised): python tools/stage_times.py [--width W --height H]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--octaves", type=int, default=4)
ap.add_argument("--blur-mode", type=int, default=0)
a = ap.parse_args()
ctx = sift_amd.SiftHip(0)
frame = sift_amd.blob_image(a.width, a.height, seed=1)
d = torch.from_numpy(frame).cuda()
out = np.zeros(1 << 17, sift_amd.FEAT_DTYPE)
prm = SiftParams(octaves=a.octaves, blur_mode=a.blur_mode)
n = ctx.extract_dev(d, a.width, a.height, prm, out)
ctx.profile(True)
ctx.profile_read(reset=True)
for _ in range(a.steps):
    n = ctx.extract_dev(d, a.width, a.height, prm, out)
prof = ctx.profile_read(reset=True)
ctx.profile(False)
print("features", n, " ".join(f"{k}={v[0] / a.steps:.3f}" for k, v in prof.items() if v[1]),
      "total=%.3f" % (sum(v[0] for v in prof.values()) / a.steps))
