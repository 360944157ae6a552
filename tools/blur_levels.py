#!/usr/bin/env python3
"""Per-level times of the separable blur (row pass, column pass) on one octave-sized image,
streaming column pass on and off, with a bit-equality check between the two:
    python tools/blur_levels.py [--width 7680 --height 4320] [--segs 64,128,256]"""
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
ap.add_argument("--segs", default="")
ap.add_argument("--levels", default="0,1,2,3,4")
a = ap.parse_args()
ctx = sift_amd.SiftHip(0)
rng = np.random.default_rng(3)
img = rng.random((a.height, a.width), dtype=np.float32)
segs = [int(s) for s in a.segs.split(",") if s] or [0]


def timed(p):
    ctx.gau_blur(img, p)
    ctx.profile(True)
    ctx.profile_read(reset=True)
    out = ctx.gau_blur(img, p)
    pr = ctx.profile_read(reset=True)
    ctx.profile(False)
    row = pr["blur_row"][0] + pr["blur_row_small"][0]
    col = pr["blur_col"][0] + pr["blur_col_small"][0] + pr["blur_fused"][0]
    return out, row * 1e3, col * 1e3


for lvl in [int(x) for x in a.levels.split(",")]:
    p = float(np.float32(1.6) * np.float32(np.sqrt(2.0) ** lvl) * np.float32(a.coef))
    K = len(sift_amd.gau_matrix(p))
    ctx.set_tuning("blur_stream", 0)
    ctx.set_tuning("blur_seg", 0)
    ref, row0, col0 = timed(p)
    line = "level %d K=%3d  two-pass: row %.1f us col %.1f us |" % (lvl, K, row0, col0)
    ctx.set_tuning("blur_stream", 1)
    for sg in segs:
        ctx.set_tuning("blur_seg", sg or 0)
        out, row1, col1 = timed(p)
        same = np.array_equal(out.view(np.uint32), ref.view(np.uint32))
        line += " seg %s: row %.1f col %.1f %s |" % (sg or "auto", row1, col1, "same" if same else "DIFFERENT")
    print(line, flush=True)
