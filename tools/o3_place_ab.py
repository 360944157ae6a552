#!/usr/bin/env python3
"""Frame time of the extract for placements of octave 3 on the side streams (sift_hip_set_tuning
"o3_place"), results compared byte for byte:  python tools/o3_place_ab.py [width height octaves]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

W, H, O = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (3840, 2160, 4)
ctx = sift_amd.SiftHip(0)
d = torch.from_numpy(sift_amd.blob_image(W, H, seed=1)).cuda()
out = ctx.pinned_array(1 << 17, sift_amd.FEAT_DTYPE)
prm = SiftParams(octaves=O)
ref = None
for rep in range(3):
    for place in (0, 3):
        ctx.set_tuning("o3_place", place)
        for _ in range(15):
            n = ctx.extract_dev(d, W, H, prm, out)
        got = out[:n].tobytes()
        ref = ref or got
        assert got == ref
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.extract_dev(d, W, H, prm, out)
            ts.append((time.perf_counter() - t0) / 20 * 1e3)
        print("o3_place %d: %s ms per frame" % (place, " ".join("%.3f" % t for t in ts)), flush=True)
