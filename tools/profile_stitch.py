#!/usr/bin/env python3
"""Runs the consumers behind the matcher a few times so that rocprofv3 can trace their kernels:
RANSAC over 18 k planted matches and the two warps of a 3840x2160 RGB frame.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stitch -- python3 tools/profile_stitch.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402
from tests.test_gpu_ransac import as_feats  # noqa: E402
from tests.test_ransac_host import HOMS, planted  # noqa: E402

ctx = sift_amd.SiftHip(0)
d = planted(1, 20000, HOMS["persp"], outlier_frac=0.6, unmatched_frac=0.1, w=3840, h=2160)
q, r = as_feats(d["x2"], d["y2"], d["size2"]), as_feats(d["x1"], d["y1"], d["size1"])
for _ in range(3):
    H, idx, iters = ctx.ransac_homog(q, r, d["match"], seed=5)
print("consensus", len(idx), "samples", iters)
rng = np.random.default_rng(0)
src = rng.integers(1, 255, (2160, 3840, 3)).astype(np.uint8)
dst = np.zeros((int(2160 * 1.1), int(3840 * 1.6), 3), np.uint8)
Hw = np.array([1.01, 0.02, 900.0, -0.015, 0.99, 40.0, 2e-6, -1e-6, 1], np.float32)
for _ in range(3):
    ctx.warp_perspective(src, Hw, dst)
    ctx.warp_cylinder(src, 4000.0)
print("warps done")
