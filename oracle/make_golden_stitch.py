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
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stitch_oracle.npz"), xyuv=xyuv, ransac_seed=77,
                        ransac_H=Hr, ransac_inliers=fin.astype(np.int32), ransac_in_min=in_min, src=src, canvas=canvas,
                        Hw=Hw, warped=warped, focal=np.float32(120.0), cyl=cyl, cyl_up=up, cyl_down=down)
    print("consensus", len(fin), "of", len(xyuv), "in_min", in_min)


if __name__ == "__main__":
    main()
