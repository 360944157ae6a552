#!/usr/bin/env python3
"""One 1M x 1M match (plus a warm-up) for a rocprofv3 counter pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE \\
        --output-format csv -d gpurun_out/pmc_match -- python3 tools/profile_match.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = sift_amd.SiftHip(0)
q, r = sift_amd.planted_descriptors(n, n)
dq, dr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
best = torch.empty(n, dtype=torch.int32, device="cuda")
for _ in range(2):
    ctx.match_u8_dev(dq, n, dr, n, 0.5, 0, best)
    ctx.synchronize()
print("accepted", int((best >= 0).sum()))
