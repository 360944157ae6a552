#!/usr/bin/env python3
"""Frames/s of sift_hip_extract_batch_dev on a batch of 1080p frames resident in HBM:
    python tools/batch_rate.py [--frames 32 --width 1920 --height 1080]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--octaves", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--cap", type=int, default=1 << 14, help="feature capacity per frame")
a = ap.parse_args()
ctx = sift_amd.SiftHip(0)
base = [sift_amd.blob_image(a.width, a.height, seed=40 + k) for k in range(4)]
frames = np.stack([base[k % 4] for k in range(a.frames)])
d = torch.from_numpy(frames).cuda()
cap = a.cap
out = np.zeros((a.frames, cap), sift_amd.FEAT_DTYPE)
n = np.zeros(a.frames, np.int32)
prm = SiftParams(octaves=a.octaves)._c()


def run():
    rc = ctx.lib.sift_hip_extract_batch_dev(ctx.h, C.c_void_p(d.data_ptr()), a.width, a.height, a.frames, C.byref(prm),
                                            C.c_void_p(out.ctypes.data), cap, n.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0, rc


run()
best = 1e9
for _ in range(a.reps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    run()
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t)
print(f"{a.frames} frames {a.width}x{a.height}: {best / a.frames * 1e3:.3f} ms/frame, {a.frames / best:.1f} frames/s, "
      f"features/frame {n.mean():.0f}")
