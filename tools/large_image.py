"""The large-image configuration: one synthetic 8192x8192 frame, 6 octaves, resident in HBM.
Reports ms per frame, device memory the context holds, features, and the stage times."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
octaves = 6
free0, total = torch.cuda.mem_get_info()
ctx = sift_amd.SiftHip(0)
frame = sift_amd.blob_image(W, H, seed=1)
d = torch.from_numpy(frame).cuda()
cap = 1 << 18
out = ctx.pinned_array(cap, sift_amd.FEAT_DTYPE)
prm = SiftParams(octaves=octaves)
for _ in range(8):
    n = ctx.extract_dev(d, W, H, prm, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    n = ctx.extract_dev(d, W, H, prm, out)
ms = (time.perf_counter() - t0) / reps * 1e3
free1, _ = torch.cuda.mem_get_info()
ctx.profile(True)
ctx.profile_read()
ctx.extract_dev(d, W, H, prm, out)
prof = ctx.profile_read()
ctx.profile(False)
blur_ms = sum(prof[k][0] for k in ("blur_row", "blur_col", "blur_row_small", "blur_col_small", "blur_fused"))
px = sum((int(W * s) * int(H * s)) for s in (2, 1, .5, .25, .125, .0625)[:octaves])
print(f"{W}x{H}, {octaves} octaves: {ms:.2f} ms per frame, {n} features, device memory in use "
      f"{(free0 - free1) / 2**30:.1f} GiB (frame included), blur stage {blur_ms:.2f} ms = "
      f"{px * 5 * 2 * 8 / blur_ms / 1e6:.0f} GB/s of its 8 B per pixel per pass; stages "
      + " ".join(f"{k}={v[0]:.2f}" for k, v in prof.items() if v[1]))
