#!/usr/bin/env python3
"""Float blur (blur_mode 1), per level: the sweeping one-launch form against the tiled one-launch form and the
two-launch form, with a bit-equality check:
  python tools/blur_sweep.py [--width 7680 --height 4320 --coef 1 --segs 0,128,176,256]"""
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
ap.add_argument("--levels", default="2,3,4")
ap.add_argument("--segs", default="0", help="rows per segment to try (0 = the library's choice)")
a = ap.parse_args()
ctx = sift_amd.SiftHip(0)
rng = np.random.default_rng(3)
img = rng.random((a.height, a.width), dtype=np.float32)


def timed(p):
    ctx.gau_blur(img, p, blur_mode=1)
    ctx.profile(True)
    ctx.profile_read(reset=True)
    best = 1e9
    for _ in range(3):
        out = ctx.gau_blur(img, p, blur_mode=1)
        pr = ctx.profile_read(reset=True)
        t = sum(pr[k][0] for k in ("blur_row", "blur_row_small", "blur_col", "blur_col_small", "blur_fused"))
        best = min(best, t)
    ctx.profile(False)
    return out, best * 1e3


for lvl in [int(x) for x in a.levels.split(",")]:
    p = float(np.float32(1.6) * np.float32(np.sqrt(2.0) ** lvl) * np.float32(a.coef))
    K = len(sift_amd.gau_matrix(p))
    ctx.set_tuning("blur_sweep", 0)
    ctx.set_tuning("blur_fused", 0)
    ref, t2 = timed(p)
    ctx.set_tuning("blur_fused", 99)
    out1, t1 = timed(p)
    line = "level %d K=%3d  two launches %.1f us | tiled one-launch %.1f us (%s)" % (
        lvl, K, t2, t1, "same bits" if np.array_equal(out1.view(np.uint32), ref.view(np.uint32)) else "DIFFERENT")
    for sg in [int(x) for x in a.segs.split(",")]:
        ctx.set_tuning("blur_sweep", sg if sg > 0 else -1)
        out, ts = timed(p)
        same = np.array_equal(out.view(np.uint32), ref.view(np.uint32))
        line += " | sweep seg %d: %.1f us (%s)" % (sg, ts, "same bits" if same else "DIFFERENT, max diff %.2e" % float(np.abs(out - ref).max()))
    ctx.set_tuning("blur_sweep", -1)
    ctx.set_tuning("blur_fused", -1)
    print(line, flush=True)
