"""4K extract per frame with the feature buffer in pageable vs page-locked host memory
(the first dozen calls on a fresh page-locked buffer are slow: the second round is the steady state)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

W, H = 3840, 2160
ctx = sift_amd.SiftHip(0)
frame = sift_amd.blob_image(W, H, seed=1)
d = torch.from_numpy(frame).cuda()
cap = 1 << 16
prm = SiftParams(octaves=4)
pageable = np.zeros(cap, sift_amd.FEAT_DTYPE)
pin_t = torch.empty(cap * 156, dtype=torch.uint8).pin_memory()
pinned = pin_t.numpy().view(sift_amd.FEAT_DTYPE)
cases = [("pageable", pageable), ("torch-pinned", pinned),
         ("sift_hip_host_alloc", ctx.pinned_array(cap, sift_amd.FEAT_DTYPE))]
for name, out in cases * 2:
    for _ in range(3):
        n = ctx.extract_dev(d, W, H, prm, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        n = ctx.extract_dev(d, W, H, prm, out)
    print(name, n, round((time.perf_counter() - t0) / 50 * 1e3, 4), "ms/frame", flush=True)
for _, o in cases:
    assert np.array_equal(pageable[:n], o[:n])
