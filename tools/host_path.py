"""sift_hip_extract (image handed over as a host buffer) on a 4K frame: ordinary vs page-locked
input and output buffers."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

W, H = 3840, 2160
ctx = sift_amd.SiftHip(0)
frame = sift_amd.blob_image(W, H, seed=1)
cap = 1 << 16
prm = SiftParams(octaves=4)._c()
pin_in = ctx.pinned_array(W * H, np.float32)
pin_in[:] = frame.reshape(-1)
outs = {"pageable": np.zeros(cap, sift_amd.FEAT_DTYPE), "pinned": ctx.pinned_array(cap, sift_amd.FEAT_DTYPE)}
ins = {"pageable": np.ascontiguousarray(frame.reshape(-1)), "pinned": pin_in}
n = C.c_int(0)
for rnd in range(2):
    for ki, vi in ins.items():
        for ko, vo in outs.items():
            def run():
                rc = ctx.lib.sift_hip_extract(ctx.h, C.c_void_p(vi.ctypes.data), W, H, 1, C.byref(prm),
                                              C.c_void_p(vo.ctypes.data), cap, C.byref(n), None)
                assert rc == 0
            for _ in range(15):
                run()
            t0 = time.perf_counter()
            for _ in range(20):
                run()
            print(f"in {ki:8s} out {ko:8s}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/frame, {n.value} features", flush=True)
