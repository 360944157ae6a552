#!/usr/bin/env python3
"""Per-level table of the exact blur for the octaves of a 4K frame: measured time of each launch form against
the bound max(VALU floor, HBM floor) of the two-pass form.
    python tools/level_table.py [blur_mode] > profiles/rNN_blur_levels.txt
VALU floor: 4.56 ns per wave-tap per SIMD (half a v_pk_mul_f32 + v_cvt_f64_f32 + v_add_f64), 1024 SIMDs, two passes
(blur_mode 1: 0.87 ns, one fused multiply-add at two cycles per 64 lanes);
HBM floor: 8 B per pixel per pass at 6.3 TB/s (what a copy kernel reaches on this part)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402

ctx = sift_amd.SiftHip(0)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ns_per_wave_tap = 4.56e-3 if mode == 0 else 0.87e-3
rng = np.random.default_rng(3)
print("octave level  K    row us   col us  one-launch us   total us   bound us (valu / hbm)   efficiency")
tot_t = tot_b = 0.0
for o, (w, h) in enumerate(((7680, 4320), (3840, 2160), (1920, 1080), (960, 540))):
    img = rng.random((h, w), dtype=np.float32)
    for lvl in range(5):
        p = float(np.float32(1.6) * np.float32(np.sqrt(2.0) ** lvl) * np.float32(2 ** o))
        K = len(sift_amd.gau_matrix(p))
        ctx.gau_blur(img, p, mode)
        best = None
        for _ in range(3):
            ctx.profile(True)
            ctx.profile_read(reset=True)
            ctx.gau_blur(img, p, mode)
            pr = ctx.profile_read(reset=True)
            ctx.profile(False)
            row = (pr["blur_row"][0] + pr["blur_row_small"][0]) * 1e3
            col = (pr["blur_col"][0] + pr["blur_col_small"][0]) * 1e3
            fus = pr["blur_fused"][0] * 1e3
            if best is None or row + col + fus < sum(best):
                best = (row, col, fus)
        row, col, fus = best
        px = w * h
        valu = 2 * px * K / 64.0 / 1024.0 * ns_per_wave_tap
        hbm = 2 * px * 8 / 6.3e6
        bound = max(valu, hbm)
        t = row + col + fus
        tot_t += t
        tot_b += bound
        print("%4d %5d %4d %8.1f %8.1f %12.1f %12.1f %10.1f (%.1f / %.1f) %10.2f" % (
            o, lvl, K, row, col, fus, t, bound, valu, hbm, bound / t))
print("sum: %.1f us measured, %.1f us bound, efficiency %.2f" % (tot_t, tot_b, tot_b / tot_t))
