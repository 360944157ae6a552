#!/usr/bin/env python3
"""Median wall time of single 4K extracts on the blob frame and on the textured frame, with the serial per-class kernel
times of the latter; run with SIFT_HIP_LIB=<variant build> to compare libraries:  python tools/frame_pair_time.py [rounds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = sift_amd.SiftHip(0)
W, H = 3840, 2160
frames = {"blob": torch.from_numpy(sift_amd.blob_image(W, H, seed=1)).cuda(),
          "textured": torch.from_numpy(sift_amd.textured_image(W, H, seed=3)).cuda()}
out = ctx.pinned_array(1 << 18, sift_amd.FEAT_DTYPE)
prm = SiftParams(octaves=4)
for r in range(rounds):
    for name, d in frames.items():
        for _ in range(10):
            n = ctx.extract_dev(d, W, H, prm, out)
        ts = []
        for _ in range(60):
            t0 = time.perf_counter()
            ctx.extract_dev(d, W, H, prm, out)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%s round %d: median %.3f ms, features %s" % (name, r, float(np.median(ts)), n if isinstance(n, int) else len(n)), flush=True)
ctx.profile(True)
ctx.profile_read(reset=True)
for _ in range(3):
    ctx.extract_dev(frames["textured"], W, H, prm, out)
pr = ctx.profile_read(reset=True)
ctx.profile(False)
print("textured, serial ms per class:", {k: round(v[0] / 3, 4) for k, v in pr.items() if v[1]})
