#!/usr/bin/env python3
"""One level of the blur on an octave-sized image, for rocprofv3 counter passes:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \\
        --kernel-trace --output-format csv -d gpurun_out/pmc_level -- python3 tools/profile_blur_level.py [sigma [width height [blur_mode]]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402

p = float(sys.argv[1]) if len(sys.argv) > 1 else 6.4
w = int(sys.argv[2]) if len(sys.argv) > 2 else 7680
h = int(sys.argv[3]) if len(sys.argv) > 3 else 4320
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = sift_amd.SiftHip(0)
img = np.random.default_rng(3).random((h, w), dtype=np.float32)
for _ in range(3):
    out = ctx.gau_blur(img, p, mode)
print("taps", len(sift_amd.gau_matrix(p)), "checksum", float(out.sum(dtype=np.float64)))
