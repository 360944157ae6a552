"""Kernel time and achieved bandwidth of the stitcher's warps on a 3840x2160 RGB frame
(perspective warp onto the 1.6 x 1.1 canvas blen2img makes, Blend.cpp:920; cylinder warp).
Algorithmic bytes: 3 written + 3 read per output pixel that has a source (perspective),
3 + 3 per pixel (cylinder).  Usage: python tools/warp_rate.py  (GPU box)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402


def main():
    ctx = sift_amd.SiftHip(0)
    rng = np.random.default_rng(0)
    src = rng.integers(1, 255, (2160, 3840, 3)).astype(np.uint8)
    dst = np.zeros((int(2160 * 1.1), int(3840 * 1.6), 3), np.uint8)
    H = np.array([1.01, 0.02, 900.0, -0.015, 0.99, 40.0, 2e-6, -1e-6, 1], np.float32)
    ctx.warp_perspective(src, H, dst)
    ctx.warp_cylinder(src, 4000.0)
    ctx.profile(True)
    reps = 10
    for _ in range(reps):
        out = ctx.warp_perspective(src, H, np.zeros_like(dst))
    ms, n = ctx.profile_read()["warp"]
    covered = int(out.reshape(-1, 3).any(axis=1).sum())
    t = ms / n
    print(json.dumps({"kernel": "warp_perspective_kernel", "canvas": list(dst.shape[:2]), "covered_px": covered,
                      "ms": round(t, 4), "GBps": round(covered * 6 / t / 1e6, 1)}))
    for _ in range(reps):
        ctx.warp_cylinder(src, 4000.0)
    ms, n = ctx.profile_read()["warp"]
    t = ms / n
    print(json.dumps({"kernel": "warp_cylinder_kernel", "px": 3840 * 2160, "ms": round(t, 4),
                      "GBps": round(3840 * 2160 * 6 / t / 1e6, 1)}))
    ctx.close()


if __name__ == "__main__":
    main()
