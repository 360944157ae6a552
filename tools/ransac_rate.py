"""Times the GPU RANSAC (sift_hip_ransac_homog) against the host implementation
(sift_amd/host/xform.cpp) on planted matches of several sizes and inlier shares.
Usage: python tools/ransac_rate.py  (on a GPU box; prints one JSON line per case)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd  # noqa: E402
from tests.test_gpu_ransac import as_feats  # noqa: E402
from tests.test_ransac_host import HOMS, planted, run_host  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    ctx = sift_amd.SiftHip(0)
    lib = C.CDLL(os.path.join(ROOT, "sift_amd", "lib", "libsift_host.so"))
    vp = C.c_void_p
    lib.sift_host_ransac.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_float,
                                     C.c_uint, vp, vp]
    for n2, out in ((2000, 0.3), (20000, 0.3), (20000, 0.6), (200000, 0.5), (1000000, 0.5)):
        d = planted(1, n2, HOMS["persp"], outlier_frac=out, unmatched_frac=0.1, w=3840, h=2160)
        q = as_feats(d["x2"], d["y2"], d["size2"])
        r = as_feats(d["x1"], d["y1"], d["size1"])
        ctx.ransac_homog(q, r, d["match"], seed=5)  # warm-up (allocations)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            H, idx, iters = ctx.ransac_homog(q, r, d["match"], seed=5)
        t_gpu = (time.perf_counter() - t0) / reps
        sel = np.nonzero(d["match"] >= 0)[0]
        mi = d["match"][sel]
        xyuv = np.stack([d["x2"][sel] / d["size2"][sel], d["y2"][sel] / d["size2"][sel],
                         d["x1"][mi] / d["size1"][mi], d["y1"][mi] / d["size1"][mi]], axis=1).astype(np.float32)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.ransac_homog_pairs(xyuv, seed=5)
        t_pairs = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        n_in, Hh, idxh = run_host(lib, d, 5)
        t_cpu = time.perf_counter() - t0
        same = H is not None and np.array_equal(H.reshape(-1), Hh) and np.array_equal(idx, idxh)
        print(json.dumps({"matches": int((d["match"] >= 0).sum()), "outlier_share": out, "samples": iters,
                          "consensus": int(len(idx)), "gpu_ms": round(t_gpu * 1e3, 3), "gpu_pairs_ms": round(t_pairs * 1e3, 3),
                          "host_ms": round(t_cpu * 1e3, 3), "equal": bool(same)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
