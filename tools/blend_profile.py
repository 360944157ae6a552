#!/usr/bin/env python3
"""One multi-band blend of two 4K RGB frames (overlap a quarter of the width) for a kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_blend -- python3 tools/blend_profile.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402

W, H = 3840, 2160
ctx = sift_amd.SiftHip(0)
rng = np.random.default_rng(11)
left = rng.integers(1, 255, (H, W, 3)).astype(np.uint8)
right = rng.integers(1, 255, (H, W, 3)).astype(np.uint8)
for _ in range(3):
    l2, r2 = left.copy(), right.copy()
    t0 = time.perf_counter()
    ctx.multiband_blend(l2, r2, W // 4, H - 7)
    print("blend call %.2f ms" % ((time.perf_counter() - t0) * 1e3))
