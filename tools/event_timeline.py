#!/usr/bin/env python3
"""When each launch group of one extract began and ended, per stream, from HIP events with the
streams left concurrent (no tracer: the host launches at its normal pace):
    python tools/event_timeline.py [--width W --height H --octaves N --blur-mode M]
Lines "timeline q<stream> <begin us> + <duration us> <class>" on stderr, then the frame's wall time."""
import argparse
import os
import sys
import time

import numpy as np

os.environ["SIFT_HIP_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--octaves", type=int, default=4)
ap.add_argument("--blur-mode", type=int, default=0)
a = ap.parse_args()
ctx = sift_amd.SiftHip(0)
frame = sift_amd.blob_image(a.width, a.height, seed=1)
d = torch.from_numpy(frame).cuda()
out = ctx.pinned_array(1 << 17, sift_amd.FEAT_DTYPE)
prm = SiftParams(octaves=a.octaves, blur_mode=a.blur_mode)
for _ in range(20):
    n = ctx.extract_dev(d, a.width, a.height, prm, out)
t0 = time.perf_counter()
for _ in range(10):
    n = ctx.extract_dev(d, a.width, a.height, prm, out)
ms = (time.perf_counter() - t0) * 100.0
sys.stderr.flush()
ctx.profile_read(reset=True)  # prints the last frame's groups
print("features %d, %.3f ms per frame with the events recorded" % (n, ms))
