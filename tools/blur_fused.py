#!/usr/bin/env python3
"""Per-level time of the blur, two launches (row pass, column pass) against the one-launch form, in both
arithmetic modes, with a bit-equality check:  python tools/blur_fused.py [--width 7680 --height 4320 --coef 1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=7680)
ap.add_argument("--height", type=int, default=4320)
ap.add_argument("--coef", type=float, default=1.0, help="octave coefficient (1 = octave 0, 2 = octave 1 ...)")
ap.add_argument("--levels", default="0,1,2,3,4")
a = ap.parse_args()
ctx = sift_amd.SiftHip(0)
rng = np.random.default_rng(3)
img = rng.random((a.height, a.width), dtype=np.float32)


def timed(p, mode):
    ctx.gau_blur(img, p, blur_mode=mode)
    ctx.profile(True)
    ctx.profile_read(reset=True)
    out = ctx.gau_blur(img, p, blur_mode=mode)
    pr = ctx.profile_read(reset=True)
    ctx.profile(False)
    row = pr["blur_row"][0] + pr["blur_row_small"][0]
    col = pr["blur_col"][0] + pr["blur_col_small"][0] + pr["blur_fused"][0]
    return out, (row + col) * 1e3


for mode in (0, 1):
    for lvl in [int(x) for x in a.levels.split(",")]:
        p = float(np.float32(1.6) * np.float32(np.sqrt(2.0) ** lvl) * np.float32(a.coef))
        K = len(sift_amd.gau_matrix(p))
        ctx.set_tuning("blur_fused", 0)
        ref, t2 = timed(p, mode)
        ctx.set_tuning("blur_fused", 99)
        out, t1 = timed(p, mode)
        same = np.array_equal(out.view(np.uint32), ref.view(np.uint32))
        err = float(np.abs(out - ref).max())
        print("blur_mode %d level %d K=%3d  two launches %.1f us | one launch %.1f us  %s (max diff %.2e)" % (
            mode, lvl, K, t2, t1, "same bits" if same else "DIFFERENT", err), flush=True)
