"""ctypes binding of the C ABI declared in include/sift_hip.h.

There is no fallback: if the shared library is missing or the device is not a
gfx950 GPU the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

#: numpy view of ``sift_hip_feat`` (156 bytes, no padding)
FEAT_DTYPE = np.dtype(
    [
        ("size", "<f4"),
        ("kai", "<i4"),
        ("sigmaOCT", "<f4"),
        ("x", "<i4"),
        ("y", "<i4"),
        ("mm", "<f4"),
        ("sita", "<i4"),
        ("descr", "u1", (128,)),
    ]
)
assert FEAT_DTYPE.itemsize == 156

KERNEL_CLASSES = ("resample", "blur_row", "blur_col", "detect", "compact", "orient", "emit", "descr", "match", "gray",
                  "blur_row_small", "blur_col_small", "ransac", "warp", "blend", "blur_fused")

OK, ERR_ARG, ERR_NOSPC, ERR_HIP, ERR_NOMEM, ERR_SAME, ERR_NODEVICE, ERR_LIMIT = 0, -1, -2, -3, -4, -5, -6, -7


class _Params(C.Structure):
    _fields_ = [
        ("octaves", C.c_int32),
        ("intvls", C.c_int32),
        ("sigma0", C.c_float),
        ("dx_thr", C.c_float),
        ("subpixel", C.c_int32),
        ("blur_mode", C.c_int32),
        ("contr_thr", C.c_float),
        ("textbook", C.c_int32),
    ]


@dataclass
class SiftParams:
    """Run-time form of the tunables of the reference's Sift.hpp:44-77."""

    octaves: int = 0
    intvls: int = 3
    sigma0: float = 1.6
    dx_thr: float = 0.03
    subpixel: int = 0
    blur_mode: int = 0
    contr_thr: float = 0.05
    textbook: int = 0  # 1 = quality mode (incremental sigmas, radians, normalised descriptor); not the reference's output

    def _c(self) -> _Params:
        return _Params(self.octaves, self.intvls, self.sigma0, self.dx_thr, self.subpixel, self.blur_mode,
                       self.contr_thr, self.textbook)


class SiftHipError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"sift_hip status {status}: {msg}")
        self.status = status


def library_path() -> str:
    """In-tree build product; SIFT_HIP_LIB points at another build (kernel experiments)."""
    return os.environ.get("SIFT_HIP_LIB") or os.path.join(_HERE, "lib", "libsift_hip.so")


_lib = None


def load_library() -> C.CDLL:
    """Loads libsift_hip.so and declares every prototype of sift_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    P = C.POINTER(_Params)
    protos = {
        "sift_hip_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "sift_hip_destroy": (C.c_int, [vp]),
        "sift_hip_default_params": (None, [P]),
        "sift_hip_strerror": (C.c_char_p, [C.c_int]),
        "sift_hip_last_error": (C.c_char_p, [vp]),
        "sift_hip_stream": (vp, [vp]),
        "sift_hip_synchronize": (C.c_int, [vp]),
        "sift_hip_host_alloc": (C.c_int, [vp, C.POINTER(C.c_void_p), C.c_size_t]),
        "sift_hip_host_free": (C.c_int, [vp, vp]),
        "sift_hip_extract": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, P, vp, C.c_int, ip, vp]),
        "sift_hip_extract_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, P, vp, C.c_int, ip, vp]),
        "sift_hip_extract_batch": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, P, vp, C.c_int, ip]),
        "sift_hip_extract_batch_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, P, vp, C.c_int, ip]),
        "sift_hip_last_offsets": (C.c_int, [vp, fp, C.c_int]),
        "sift_hip_gau_matrix": (C.c_int, [C.c_float, C.c_int, fp, C.c_int]),
        "sift_hip_gau_blur": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]),
        "sift_hip_first": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp]),
        "sift_hip_convert_gray": (C.c_int, [vp, vp, C.c_int, vp]),
        "sift_hip_octave_levels": (C.c_int, [vp, vp, C.c_int, C.c_int, P, C.c_int, vp, vp, ip, ip]),
        "sift_hip_pyramid_octaves": (C.c_int, [C.c_int, C.c_int]),
        "sift_hip_match_u8": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp]),
        "sift_hip_match_u8_dev": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp]),
        "sift_hip_match_u8_noalloc": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp]),
        "sift_hip_match_u8_sharded": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp]),
        "sift_hip_match_u8_sharded_ragged": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_float, C.c_int, vp,
                                                       vp, vp]),
        "sift_hip_ragged_offsets": (C.c_int, [vp, C.c_int, vp]),
        "sift_hip_ransac_homog": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, C.c_uint,
                                            vp, vp, ip, ip]),
        "sift_hip_ransac_homog_dev": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_float, C.c_float,
                                                C.c_uint, vp, vp, ip, ip]),
        "sift_hip_ransac_homog_pairs": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint, vp, vp, ip,
                                                  ip]),
        "sift_hip_warp_perspective": (C.c_int, [vp, vp, C.c_int, C.c_int, fp, vp, C.c_int, C.c_int]),
        "sift_hip_warp_perspective_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, fp, vp, C.c_int, C.c_int]),
        "sift_hip_warp_cylinder": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
        "sift_hip_warp_cylinder_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
        "sift_hip_multiband_blend": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "sift_hip_multiband_blend_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "sift_hip_panorama_pair": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
        "sift_hip_panorama_pair_f32": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
        "sift_hip_keep_descriptors": (C.c_int, [vp, C.POINTER(vp), ip]),
        "sift_hip_release_descriptors": (None, [vp, vp]),
        "sift_hip_descriptor_count": (C.c_int, [vp]),
        "sift_hip_match_kept": (C.c_int, [vp, vp, vp, C.c_float, C.c_int, vp, vp, vp]),
        "sift_hip_set_tuning": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "sift_hip_profile": (C.c_int, [vp, C.c_int]),
        "sift_hip_profile_read": (C.c_int, [vp, fp, ip, C.c_int]),
        "sift_hip_kernel_class_name": (C.c_char_p, [C.c_int]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # AttributeError here = the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    lib._protos = tuple(protos)
    _lib = lib
    return lib


def gau_matrix(p: float, mat_len: int = 0) -> np.ndarray:
    """Gaus::gau_matrix (imglib.cpp:127-155); host arithmetic inside the library."""
    lib = load_library()
    buf = np.zeros(8192, np.float32)
    n = lib.sift_hip_gau_matrix(p, mat_len, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size)
    if n < 0:
        raise SiftHipError(n, "gau_matrix")
    return buf[:n].copy()


def pyramid_octaves(w: int, h: int) -> int:
    return load_library().sift_hip_pyramid_octaves(w, h)


def _ptr(a) -> C.c_void_p:
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if isinstance(a, int):
        return C.c_void_p(a)
    return C.c_void_p(a.data_ptr())  # torch tensor (device or host)


def _f32(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class SiftHip:
    """One context = one GPU + one HIP stream (not thread-safe)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.sift_hip_create(C.byref(h), device)
        if rc != OK:
            raise SiftHipError(rc, self.lib.sift_hip_strerror(rc).decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            for p in getattr(self, "_pinned", []):
                self.lib.sift_hip_host_free(self.h, p)
            self._pinned = []
            self.lib.sift_hip_destroy(self.h)
            self.h = None

    def pinned_array(self, count: int, dtype) -> np.ndarray:
        """`count` items of `dtype` in page-locked host memory (sift_hip_host_alloc): result copies
        into it run by direct DMA.  The memory lives until close(); keep no views beyond that."""
        dtype = np.dtype(dtype)
        nbytes = max(1, count * dtype.itemsize)
        p = C.c_void_p()
        self._check(self.lib.sift_hip_host_alloc(self.h, C.byref(p), nbytes))
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(p)
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        a = np.frombuffer(buf, dtype=np.uint8)[: count * dtype.itemsize].view(dtype)
        a[...] = np.zeros((), dtype)
        return a

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, allow=()):
        if rc != OK and rc not in allow:
            detail = self.lib.sift_hip_last_error(self.h).decode()
            raise SiftHipError(rc, self.lib.sift_hip_strerror(rc).decode() + (f" ({detail})" if detail else ""))
        return rc

    def synchronize(self):
        self._check(self.lib.sift_hip_synchronize(self.h))

    # ---- extract ---------------------------------------------------------------
    def extract(self, img: np.ndarray, params: SiftParams | None = None, cap: int = 1 << 16, want_hist=False):
        """img: HxW gray or HxWx3 RGB float32 in [0,1] on the host.
        Returns (features[FEAT_DTYPE], hist or None)."""
        img = _f32(img)
        ch = 1 if img.ndim == 2 else img.shape[2]
        h, w = img.shape[:2]
        prm = (params or SiftParams())._c()
        while True:
            out = np.zeros(cap, FEAT_DTYPE)
            hist = np.zeros((cap, 128), np.float32) if want_hist else None
            n = C.c_int(0)
            rc = self.lib.sift_hip_extract(self.h, _ptr(img), w, h, ch, C.byref(prm), _ptr(out), cap, C.byref(n),
                                           _ptr(hist))
            if rc == ERR_NOSPC:
                cap = n.value
                continue
            self._check(rc)
            return out[: n.value], (hist[: n.value] if want_hist else None)

    def last_offsets(self, n: int) -> np.ndarray:
        """(xc, xr, xi) of the n features the last extract returned (subpixel = 1)."""
        out = np.zeros((max(n, 1), 3), np.float32)
        got = self.lib.sift_hip_last_offsets(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), n)
        if got < 0:
            self._check(got)
        return out[:got]

    def extract_dev(self, d_gray, w: int, h: int, params: SiftParams | None, out: np.ndarray, hist=None) -> int:
        """d_gray: device pointer (int) or torch CUDA tensor; out: preallocated FEAT_DTYPE array."""
        prm = (params or SiftParams())._c()
        n = C.c_int(0)
        rc = self.lib.sift_hip_extract_dev(self.h, _ptr(d_gray), w, h, C.byref(prm), _ptr(out), out.shape[0],
                                           C.byref(n), _ptr(hist))
        self._check(rc, allow=(ERR_NOSPC,))
        return n.value

    def extract_batch(self, frames: np.ndarray, params: SiftParams | None = None, cap: int = 1 << 15):
        frames = _f32(frames)
        nf, h, w = frames.shape
        prm = (params or SiftParams())._c()
        out = np.zeros((nf, cap), FEAT_DTYPE)
        n = np.zeros(nf, np.int32)
        rc = self.lib.sift_hip_extract_batch(self.h, _ptr(frames), w, h, nf, C.byref(prm), _ptr(out), cap,
                                             n.ctypes.data_as(C.POINTER(C.c_int)))
        self._check(rc, allow=(ERR_NOSPC,))
        return out, n

    def extract_batch_dev(self, d_frames, w: int, h: int, nframes: int, params: SiftParams | None, out: np.ndarray,
                          n: np.ndarray):
        prm = (params or SiftParams())._c()
        rc = self.lib.sift_hip_extract_batch_dev(self.h, _ptr(d_frames), w, h, nframes, C.byref(prm), _ptr(out),
                                                 out.shape[1], n.ctypes.data_as(C.POINTER(C.c_int)))
        self._check(rc, allow=(ERR_NOSPC,))
        return rc

    # ---- image ops ----------------------------------------------------------------
    def gau_blur(self, img: np.ndarray, p: float, blur_mode: int = 0, mat_len: int = 0) -> np.ndarray:
        img = _f32(img)
        h, w = img.shape
        out = np.empty_like(img)
        self._check(self.lib.sift_hip_gau_blur(self.h, _ptr(img), _ptr(out), w, h, p, mat_len, blur_mode))
        return out

    def first(self, img: np.ndarray, ratio: float) -> np.ndarray:
        img = _f32(img)
        h, w = img.shape
        nh = int(np.floor(np.float32(h) * np.float32(ratio)))
        nw = int(np.floor(np.float32(w) * np.float32(ratio)))
        out = np.empty((nh, nw), np.float32)
        self._check(self.lib.sift_hip_first(self.h, _ptr(img), w, h, ratio, _ptr(out)))
        return out

    def convert_gray(self, rgb: np.ndarray) -> np.ndarray:
        rgb = _f32(rgb)
        n = rgb.size // 3
        out = np.empty(rgb.shape[:-1], np.float32)
        self._check(self.lib.sift_hip_convert_gray(self.h, _ptr(rgb), n, _ptr(out)))
        return out

    def octave_levels(self, gray: np.ndarray, octave: int, params: SiftParams | None = None):
        gray = _f32(gray)
        h, w = gray.shape
        p = params or SiftParams()
        size = np.float32(2.0) / np.float32(2.0 ** octave)
        ow, oh = int(np.float32(w) * size), int(np.float32(h) * size)
        gau = np.empty((p.intvls + 3, oh, ow), np.float32)
        dog = np.empty((p.intvls + 2, oh, ow), np.float32)
        cw, chh = C.c_int(0), C.c_int(0)
        prm = p._c()
        self._check(self.lib.sift_hip_octave_levels(self.h, _ptr(gray), w, h, C.byref(prm), octave, _ptr(gau),
                                                    _ptr(dog), C.byref(cw), C.byref(chh)))
        assert (cw.value, chh.value) == (ow, oh)
        return gau, dog

    # ---- match --------------------------------------------------------------------
    def match_u8(self, q: np.ndarray, r: np.ndarray, ratio: float = 0.4, ratio_form: int = 0):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 128)
        r = np.ascontiguousarray(r, np.uint8).reshape(-1, 128)
        nq, nr = q.shape[0], r.shape[0]
        best = np.full(nq, -1, np.int32)
        d1 = np.zeros(nq, np.int32)
        d2 = np.zeros(nq, np.int32)
        self._check(self.lib.sift_hip_match_u8(self.h, _ptr(q), nq, _ptr(r), nr, ratio, ratio_form, _ptr(best),
                                               _ptr(d1), _ptr(d2)))
        return best, d1, d2

    def match_u8_dev(self, d_q, nq: int, d_r, nr: int, ratio: float, ratio_form: int, d_best, d_d1=None, d_d2=None):
        self._check(self.lib.sift_hip_match_u8_dev(self.h, _ptr(d_q), nq, _ptr(d_r), nr, ratio, ratio_form,
                                                   _ptr(d_best), _ptr(d_d1), _ptr(d_d2)))

    def match_u8_noalloc(self, d_q, nq: int, d_r, nr: int, ratio: float, ratio_form: int, d_best, d_d1=None,
                         d_d2=None):
        """sift_hip_match_u8_noalloc: device pointers, no allocation inside the call, sizes bounded."""
        self._check(self.lib.sift_hip_match_u8_noalloc(self.h, _ptr(d_q), nq, _ptr(d_r), nr, ratio, ratio_form,
                                                       _ptr(d_best), _ptr(d_d1), _ptr(d_d2)))

    def keep_descriptors(self):
        """sift_hip_keep_descriptors: the descriptors of the last extract's features stay on the device; returns
        (opaque set, count).  Free the set with release_descriptors."""
        h = C.c_void_p()
        n = C.c_int(0)
        self._check(self.lib.sift_hip_keep_descriptors(self.h, C.byref(h), C.byref(n)))
        return h, n.value

    def release_descriptors(self, dset):
        self.lib.sift_hip_release_descriptors(self.h, dset)

    def match_kept(self, qset, rset, ratio: float = 0.4, ratio_form: int = 0):
        """sift_hip_match_kept: match_u8 on two kept descriptor sets, nothing packed or uploaded."""
        nq = self.lib.sift_hip_descriptor_count(qset)
        best = np.full(max(nq, 0), -1, np.int32)
        d1 = np.zeros(max(nq, 0), np.int32)
        d2 = np.zeros(max(nq, 0), np.int32)
        self._check(self.lib.sift_hip_match_kept(self.h, qset, rset, ratio, ratio_form, _ptr(best), _ptr(d1), _ptr(d2)))
        return best, d1, d2

    # ---- RANSAC homography over the matches ------------------------------------------
    def ransac_homog(self, q_feats: np.ndarray, r_feats: np.ndarray, best: np.ndarray, m: int = 4,
                     p_badxform: float = 0.005, err_tol: float = 3.0, seed: int = 0):
        """ransac_xform (ransac/xform.cpp:37-168) with its call site's arguments as defaults
        (Sift.cpp:684).  Returns (H 3x3 float32 or None, query indices of the consensus, samples used)."""
        q_feats = np.ascontiguousarray(q_feats, FEAT_DTYPE)
        r_feats = np.ascontiguousarray(r_feats, FEAT_DTYPE)
        best = np.ascontiguousarray(best, np.int32)
        nq, nr = len(q_feats), len(r_feats)
        assert len(best) == nq
        H = np.zeros(9, np.float32)
        idx = np.full(max(nq, 1), -1, np.int32)
        n_in = C.c_int(0)
        iters = C.c_int(0)
        self._check(self.lib.sift_hip_ransac_homog(self.h, _ptr(q_feats), nq, _ptr(r_feats), nr, _ptr(best), m,
                                                   p_badxform, err_tol, seed, _ptr(H), _ptr(idx), C.byref(n_in),
                                                   C.byref(iters)))
        if n_in.value < 0:
            return None, idx[:0], iters.value
        return H.reshape(3, 3), idx[:n_in.value].copy(), iters.value

    def ransac_homog_pairs(self, xyuv: np.ndarray, m: int = 4, p_badxform: float = 0.005, err_tol: float = 3.0,
                           seed: int = 0):
        """The same over n x (x, y, u, v) pairs; returns (H or None, pair indices of the consensus, samples used)."""
        xyuv = np.ascontiguousarray(xyuv, np.float32).reshape(-1, 4)
        n = len(xyuv)
        H = np.zeros(9, np.float32)
        idx = np.full(max(n, 1), -1, np.int32)
        n_in = C.c_int(0)
        iters = C.c_int(0)
        self._check(self.lib.sift_hip_ransac_homog_pairs(self.h, _ptr(xyuv), n, m, p_badxform, err_tol, seed, _ptr(H),
                                                         _ptr(idx), C.byref(n_in), C.byref(iters)))
        if n_in.value < 0:
            return None, idx[:0], iters.value
        return H.reshape(3, 3), idx[:n_in.value].copy(), iters.value

    # ---- geometric warps of the stitcher ----------------------------------------------
    def warp_perspective(self, src_rgb: np.ndarray, H, dst_rgb: np.ndarray) -> np.ndarray:
        """_WarpPerspective (stitch/stitch.cpp:106-141); dst_rgb (rows, cols, 3) uint8 is updated in place."""
        src = np.ascontiguousarray(src_rgb, np.uint8)
        assert src.ndim == 3 and src.shape[2] == 3 and dst_rgb.dtype == np.uint8 and dst_rgb.flags.c_contiguous
        Hm = np.ascontiguousarray(H, np.float32).reshape(9)
        self._check(self.lib.sift_hip_warp_perspective(self.h, _ptr(src), src.shape[1], src.shape[0],
                                                       Hm.ctypes.data_as(C.POINTER(C.c_float)), _ptr(dst_rgb),
                                                       dst_rgb.shape[1], dst_rgb.shape[0]))
        return dst_rgb

    def warp_cylinder(self, src_rgb: np.ndarray, focal: float):
        """warping (stitch/stitch.cpp:178-221) for one image: (out, upedge (w, 2), downedge (w, 2))."""
        src = np.ascontiguousarray(src_rgb, np.uint8)
        assert src.ndim == 3 and src.shape[2] == 3
        h, w = src.shape[:2]
        out = np.empty_like(src)
        up = np.zeros((w, 2), np.float32)
        down = np.zeros((w, 2), np.float32)
        self._check(self.lib.sift_hip_warp_cylinder(self.h, _ptr(src), w, h, focal, _ptr(out), _ptr(up), _ptr(down)))
        return out, up, down

    def multiband_blend(self, l_rgb: np.ndarray, r_rgb: np.ndarray, dx: int, dy: int):
        """multiBandBlend (stitch/Blend.cpp:638-740): both (rows, cols, 3) uint8 images are updated in place."""
        for a in (l_rgb, r_rgb):
            assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3 and a.flags.c_contiguous
        self._check(self.lib.sift_hip_multiband_blend(self.h, _ptr(l_rgb), l_rgb.shape[1], l_rgb.shape[0], _ptr(r_rgb),
                                                      r_rgb.shape[1], r_rgb.shape[0], dx, dy))

    def panorama_pair(self, img1: np.ndarray, img2: np.ndarray, focal: float, dx: int, blend_dy: int, top0: int, left1: int,
                      top1: int, cols: int, rows: int) -> np.ndarray:
        """blen2img's pixel work (stitch/Blend.cpp:974-1031) in one call: cylinder projections, blend, canvas."""
        for a in (img1, img2):
            assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3 and a.flags.c_contiguous
        canvas = np.empty((rows, cols, 3), np.uint8)
        self._check(self.lib.sift_hip_panorama_pair(self.h, _ptr(img1), img1.shape[1], img1.shape[0], _ptr(img2), img2.shape[1],
                                                    img2.shape[0], focal, dx, blend_dy, top0, left1, top1, _ptr(canvas), cols,
                                                    rows))
        return canvas

    def panorama_pair_f32(self, img1: np.ndarray, img2: np.ndarray, normalised: bool, focal: float, dx: int, blend_dy: int,
                          top0: int, left1: int, top1: int, cols: int, rows: int) -> np.ndarray:
        """panorama_pair on float images as ImgRaw holds them; their 8-bit copies (Imgraw.hpp:59-82) are made on the device."""
        for a in (img1, img2):
            assert a.dtype == np.float32 and a.ndim == 3 and a.shape[2] == 3 and a.flags.c_contiguous
        canvas = np.empty((rows, cols, 3), np.uint8)
        self._check(self.lib.sift_hip_panorama_pair_f32(self.h, _ptr(img1), img1.shape[1], img1.shape[0], _ptr(img2),
                                                        img2.shape[1], img2.shape[0], 1 if normalised else 0, focal, dx,
                                                        blend_dy, top0, left1, top1, _ptr(canvas), cols, rows))
        return canvas

    # ---- measurement ----------------------------------------------------------------
    def set_tuning(self, key: str, value: int):
        """Launch-shape switch of this context (sift_hip_set_tuning); results never depend on it."""
        self._check(self.lib.sift_hip_set_tuning(self.h, key.encode(), int(value)))

    def profile(self, enable: bool):
        self._check(self.lib.sift_hip_profile(self.h, int(enable)))

    def profile_read(self, reset: bool = True):
        ms = np.zeros(len(KERNEL_CLASSES), np.float32)
        cnt = np.zeros(len(KERNEL_CLASSES), np.int32)
        self._check(self.lib.sift_hip_profile_read(self.h, ms.ctypes.data_as(C.POINTER(C.c_float)),
                                                   cnt.ctypes.data_as(C.POINTER(C.c_int)), int(reset)))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(KERNEL_CLASSES)}
