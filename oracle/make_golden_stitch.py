#!/usr/bin/env python3
"""Generates tests/golden/stitch_oracle.npz: outputs of the ORACLES of the steps behind the
matcher (oracle/ransac_oracle.py, oracle/stitch_oracle.cpp) on small seeded inputs.  Those
steps' reference translation units cannot be built here (OpenCV, timer.hpp), so this file
pins the oracles and the product against regressions, not against the reference.
    python oracle/make_golden_stitch.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ransac_oracle  # noqa: E402
from tests.oracle_lib import Oracle  # noqa: E402


def inputs():
    rng = np.random.default_rng(2024)
    n = 260
    H = np.array([0.98, 0.05, 40.0, -0.04, 1.02, 15.0, 1.5e-5, -1.0e-5, 1.0])
    x, y = np.floor(rng.uniform(20, 1900, n)), np.floor(rng.uniform(20, 1060, n))
    den = H[6] * x + H[7] * y + H[8]
    u, v = (H[0] * x + H[1] * y + H[2]) / den, (H[3] * x + H[4] * y + H[5]) / den
    bad = rng.random(n) < 0.4
    u, v = np.where(bad, rng.uniform(0, 1920, n), u), np.where(bad, rng.uniform(0, 1080, n), v)
    xyuv = np.stack([x, y, np.rint(u), np.rint(v)], axis=1).astype(np.float32)
    src = rng.integers(0, 256, (64, 96, 3)).astype(np.uint8)
    canvas = rng.integers(0, 256, (70, 150, 3)).astype(np.uint8)
    Hw = np.array([1.02, 0.03, 20.0, -0.02, 0.98, 3.0, 2e-4, -1e-4, 1], np.float32)
    return xyuv, src, canvas, Hw


def main():
    xyuv, src, canvas, Hw = inputs()
    orc = Oracle()
    Hr, fin, in_min = ransac_oracle.ransac_xform(xyuv[:, :2], xyuv[:, 2:], 4, 0.005, 3.0, 77)
    warped = orc.warp_perspective(src, Hw, canvas)
    cyl, up, down = orc.warp_cylinder(src, 120.0)
    # multiBandBlend / alignMatch / getFocal (oracle/blend_oracle.cpp) on a small smooth pair
    from scipy import ndimage
    rng = np.random.default_rng(99)
    wide = ndimage.gaussian_filter(rng.random((72, 200, 3)).astype(np.float32), (3, 3, 0))
    wide = ((wide - wide.min()) / (wide.max() - wide.min()) * 235 + 15).astype(np.uint8)
    bl, br = np.ascontiguousarray(wide[:, :120]), np.ascontiguousarray(wide[:, 69:189])
    bl[:4], br[-5:] = 0, 0
    blend_l, blend_r = orc.multiband_blend(bl, br, 51, 71)
    Hb = np.array([0.99, 0.01, 60.0, -0.02, 1.01, 2.0, 3e-5, 2e-5, 1.0], np.float32)
    a_xy = np.stack([rng.uniform(5, 60, 12), rng.uniform(5, 65, 12)], axis=1).astype(np.float32)
    b_xy = (a_xy + np.array([69.0, 0.5], np.float32)).astype(np.float32)
    bfocal = np.float32(orc.get_focal(Hb, 120 * 72 * 3, 120 * 72 * 3))
    balign = np.array(orc.align_match((120, 72), (120, 72), a_xy, b_xy, float(bfocal)), np.int32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stitch_oracle.npz"), xyuv=xyuv, ransac_seed=77,
                        ransac_H=Hr, ransac_inliers=fin.astype(np.int32), ransac_in_min=in_min, src=src, canvas=canvas,
                        Hw=Hw, warped=warped, focal=np.float32(120.0), cyl=cyl, cyl_up=up, cyl_down=down,
                        blend_in_l=bl, blend_in_r=br, blend_dx=51, blend_dy=71, blend_l=blend_l, blend_r=blend_r,
                        blend_H=Hb, blend_a=a_xy, blend_b=b_xy, blend_focal=bfocal, blend_align=balign)
    print("consensus", len(fin), "of", len(xyuv), "in_min", in_min)


if __name__ == "__main__":
    main()
