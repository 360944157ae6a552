"""GPU RANSAC homography (sift_hip_ransac_homog) through the C ABI: bit-equal to the
host implementation (sift_amd/host/xform.cpp) on the same libc random stream, equal
consensus to the numpy restatement (oracle/ransac_oracle.py), and right on planted
and extracted data.  Parity with the reference itself is unpinned for this step (it
needs OpenCV and the reference holds no vectors for it)."""
import ctypes as C

import numpy as np
import pytest

import sift_amd
from tests.test_ransac_host import HOMS, host, planted, run_host, run_oracle  # noqa: F401

pytestmark = pytest.mark.gpu


def as_feats(x, y, size):
    f = np.zeros(len(x), sift_amd.FEAT_DTYPE)
    f["x"], f["y"], f["size"] = x, y, size
    return f


def run_gpu(sift, d, seed, m=4, p_bad=0.005, tol=3.0):
    return sift.ransac_homog(as_feats(d["x2"], d["y2"], d["size2"]), as_feats(d["x1"], d["y1"], d["size1"]),
                             d["match"], m, p_bad, tol, seed)


@pytest.mark.parametrize("name", list(HOMS))
@pytest.mark.parametrize("seed,n2,out", [(1, 260, 0.45), (7, 1500, 0.3), (99, 20000, 0.4), (5, 37, 0.1)])
def test_equals_host_and_restatement(sift, host, name, seed, n2, out):  # noqa: F811
    d = planted(300 + seed, n2, HOMS[name], outlier_frac=out)
    H, idx, iters = run_gpu(sift, d, seed)
    n_in, Hh, idxh = run_host(host, d, seed)
    assert H is not None and n_in == len(idx) and np.array_equal(idx, idxh)
    assert np.array_equal(H.reshape(-1).view(np.uint32), Hh.view(np.uint32))  # same arithmetic, same bits
    assert iters >= 1
    if n2 <= 1500:  # the restatement's exchange sort is a Python loop
        Ho, fin, _ = run_oracle(d, seed)
        assert np.array_equal(idx, fin)
        np.testing.assert_allclose(H.reshape(-1), Ho, rtol=2e-4, atol=1e-6)
    good = np.nonzero((d["match"] >= 0) & ~d["bad"])[0]
    assert np.isin(good, idx).mean() > 0.97
    # the form that takes ready-made pairs: same model, indices in pair numbering
    sel = np.nonzero(d["match"] >= 0)[0]
    mi = d["match"][sel]
    xyuv = np.stack([d["x2"][sel] / d["size2"][sel], d["y2"][sel] / d["size2"][sel], d["x1"][mi] / d["size1"][mi],
                     d["y1"][mi] / d["size1"][mi]], axis=1).astype(np.float32)
    Hp, idxp, itp = sift.ransac_homog_pairs(xyuv, seed=seed)
    assert np.array_equal(Hp, H) and np.array_equal(sel[idxp], idx) and itp == iters
    # a second run on the same seed repeats itself
    H2, idx2, it2 = run_gpu(sift, d, seed)
    assert np.array_equal(H, H2) and np.array_equal(idx, idx2) and it2 == iters


def test_sample_sizes_and_many_batches(sift, host):  # noqa: F811
    # m = 6; a low inlier share makes the loop run through several device batches
    d = planted(77, 400, HOMS["persp"], outlier_frac=0.8, unmatched_frac=0.0)
    for m in (4, 6):
        H, idx, iters = run_gpu(sift, d, 21, m=m)
        n_in, Hh, idxh = run_host(host, d, 21, m=m)
        assert (H is None) == (n_in < 0)
        if H is not None:
            assert np.array_equal(idx, idxh) and np.array_equal(H.reshape(-1).view(np.uint32), Hh.view(np.uint32))
    assert iters > 2048  # more than one batch


def test_no_model_cases_and_arguments(sift, host):  # noqa: F811
    d = planted(5, 40, HOMS["shift"], outlier_frac=0.0)
    d["match"][:] = -1
    d["match"][:3] = [0, 1, 2]
    assert run_gpu(sift, d, 3)[0] is None  # fewer matches than the sample size
    d["match"][:] = -1
    assert run_gpu(sift, d, 3)[0] is None
    d = planted(6, 12, HOMS["shift"], outlier_frac=1.0, unmatched_frac=0.0)
    assert run_gpu(sift, d, 3)[0] is None and run_host(host, d, 3)[0] == -1
    # match indices outside the reference array count as "no match"
    d = planted(8, 200, HOMS["shift"], outlier_frac=0.2, unmatched_frac=0.0)
    d["match"][::5] = 10 ** 6
    H, idx, _ = run_gpu(sift, d, 4)
    n_in, Hh, idxh = run_host(host, d, 4)
    assert np.array_equal(idx, idxh) and not np.isin(np.arange(0, 200, 5), idx).any()
    # empty inputs
    e = np.zeros(0, sift_amd.FEAT_DTYPE)
    assert sift.ransac_homog(e, e, np.zeros(0, np.int32))[0] is None
    q = as_feats(d["x2"], d["y2"], d["size2"])
    n_in = C.c_int(0)
    lib, vp = sift.lib, C.c_void_p
    for m, p in ((3, 0.005), (9, 0.005), (4, 0.0), (4, 1.0)):
        rc = lib.sift_hip_ransac_homog(sift.h, vp(q.ctypes.data), len(q), vp(q.ctypes.data), len(q),
                                       vp(d["match"].ctypes.data), m, p, 3.0, 1, None, None, C.byref(n_in), None)
        assert rc == sift_amd.capi.ERR_ARG
    rc = lib.sift_hip_ransac_homog(sift.h, None, len(q), vp(q.ctypes.data), len(q), vp(d["match"].ctypes.data), 4,
                                   0.005, 3.0, 1, None, None, C.byref(n_in), None)
    assert rc == sift_amd.capi.ERR_ARG


def test_repeated_positions(sift, host):  # noqa: F811
    d = planted(9, 120, HOMS["shift"], outlier_frac=0.2, unmatched_frac=0.0)
    for k in ("x2", "y2", "size2", "x1", "y1", "size1"):
        d[k][1::2] = d[k][0::2]
    H, idx, _ = run_gpu(sift, d, 11)
    n_in, Hh, idxh = run_host(host, d, 11)
    assert np.array_equal(idx, idxh) and np.array_equal(H.reshape(-1).view(np.uint32), Hh.view(np.uint32))


class DevMem:
    """Device buffers through the HIP runtime the library itself is linked to (loading a
    second runtime, e.g. torch's own copy, after the first has the device does not work)."""

    def __init__(self):
        path = "libamdhip64.so"
        with open("/proc/self/maps") as f:  # the copy libsift_hip.so brought in
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        self.hip = C.CDLL(path)
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.ptrs = []

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(p), max(nbytes, 1)) == 0
        self.ptrs.append(p)
        return p

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes)
        assert self.hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), a.nbytes, 1) == 0
        return p

    def download(self, p, out):
        assert self.hip.hipMemcpy(C.c_void_p(out.ctypes.data), p, out.nbytes, 2) == 0
        return out

    def free(self):
        for p in self.ptrs:
            self.hip.hipFree(p)
        self.ptrs = []


def test_extract_match_ransac_on_device(sift):
    """detect -> describe -> match -> RANSAC with the features and matches left in device memory."""
    a = sift_amd.blob_image(512, 384, seed=3)
    b = sift_amd.blob_image(512, 384, seed=3, shift_x=-8)
    fa, _ = sift.extract(a)
    fb, _ = sift.extract(b)
    mem = DevMem()
    d_fa, d_fb = mem.upload(fa), mem.upload(fb)
    d_qa, d_ra = mem.upload(fb["descr"]), mem.upload(fa["descr"])
    d_best = mem.alloc(4 * len(fb))
    sift.match_u8_dev(d_qa.value, len(fb), d_ra.value, len(fa), 0.5, 0, d_best.value)
    H = np.zeros(9, np.float32)
    idx = np.full(len(fb), -1, np.int32)
    n_in, iters = C.c_int(0), C.c_int(0)
    vp = C.c_void_p
    rc = sift.lib.sift_hip_ransac_homog_dev(sift.h, d_fb, len(fb), d_fa, len(fa), d_best, 4, 0.005, 3.0, 1234,
                                            vp(H.ctypes.data), vp(idx.ctypes.data), C.byref(n_in), C.byref(iters))
    assert rc == 0 and n_in.value > 20
    best = mem.download(d_best, np.zeros(len(fb), np.int32))
    mem.free()
    H2, idx2, _ = sift.ransac_homog(fb, fa, best, seed=1234)
    assert np.array_equal(H, H2.reshape(-1)) and np.array_equal(idx[: n_in.value], idx2)
    # image 2 is image 1 moved by 8 px: the model is that translation
    H = H.astype(np.float64)
    for cx, cy in ((100.0, 100.0), (256.0, 192.0), (400.0, 300.0)):
        den = H[6] * cx + H[7] * cy + H[8]
        mx, my = (H[0] * cx + H[1] * cy + H[2]) / den - cx, (H[3] * cx + H[4] * cy + H[5]) / den - cy
        assert abs(abs(mx) - 8.0) < 1.0 and abs(my) < 1.0
