"""ctypes loader for the CPU oracle (oracle/_build/libsift_oracle.so) and, when it
was built in this checkout, the reference's own object code (oracle/_ref).
Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libsift_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsift_ref.so")

FEAT_DTYPE = np.dtype(
    [("size", "<f4"), ("kai", "<i4"), ("sigmaOCT", "<f4"), ("x", "<i4"), ("y", "<i4"), ("mm", "<f4"),
     ("sita", "<i4"), ("descr", "u1", (128,))]
)

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)


def _p(a):
    return C.c_void_p(0) if a is None else C.c_void_p(a.ctypes.data)


def build_oracle():
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/libsift_oracle.so"])


class Oracle:
    def __init__(self):
        build_oracle()
        o = C.CDLL(ORACLE_SO)
        vp = C.c_void_p
        o.orc_gau_matrix.argtypes = [C.c_float, C.c_int, vp, C.c_int]
        o.orc_gau_blur.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float]
        o.orc_first.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp]
        o.orc_first_size.argtypes = [C.c_int, C.c_int, C.c_float, ip, ip]
        o.orc_fast_atan2f.argtypes = [C.c_float, C.c_float]
        o.orc_fast_atan2f.restype = C.c_float
        o.orc_convert_gray.argtypes = [vp, C.c_int, vp]
        o.orc_octave_levels.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, ip, ip]
        o.orc_extract.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]
        o.orc_extract_ex.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int,
                                     C.c_double, vp]
        o.orc_extract_quality.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_double, vp]
        o.orc_force_match.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, vp, vp, vp]
        o.orc_match_u8.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp]
        o.orc_warp_perspective.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int]
        o.orc_warp_cylinder.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp, vp, vp]
        o.orc_multiband_blend.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int]
        o.orc_align_match.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_float, vp]
        o.orc_get_focal.argtypes = [vp, C.c_size_t, C.c_size_t]
        o.orc_get_focal.restype = C.c_float
        o.orc_blend_canvas_size.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        o.orc_blend_compose.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]
        self.o = o

    def multiband_blend(self, l_rgb, r_rgb, dx, dy):
        """multiBandBlend (Blend.cpp:638-740); returns updated copies of both images."""
        l = np.array(l_rgb, np.uint8, copy=True, order="C")
        r = np.array(r_rgb, np.uint8, copy=True, order="C")
        self.o.orc_multiband_blend(_p(l), l.shape[1], l.shape[0], _p(r), r.shape[1], r.shape[0], dx, dy)
        return l, r

    def align_match(self, size1, size2, a_xy, b_xy, FL):
        """alignMatch (Blend.cpp:745-825): a = consensus features (image 2), b = their matches (image 1)."""
        a = np.ascontiguousarray(a_xy, np.float32)
        b = np.ascontiguousarray(b_xy, np.float32)
        ax, ay, bx, by = (np.ascontiguousarray(v) for v in (a[:, 0], a[:, 1], b[:, 0], b[:, 1]))
        out = np.zeros(2, np.int32)
        self.o.orc_align_match(size1[0], size1[1], size2[0], size2[1], _p(ax), _p(ay), _p(bx), _p(by), len(a), FL, _p(out))
        return int(out[0]), int(out[1])

    def get_focal(self, H9, size1, size2):
        H = None if H9 is None else np.ascontiguousarray(H9, np.float32).reshape(9)
        return float(self.o.orc_get_focal(_p(H), size1, size2))

    def blend_compose(self, w0, w1, dx, dy):
        """The tail of blen2img (Blend.cpp:999-1133) on the two projected images: (canvas, w0, w1 after the blend)."""
        a = np.array(w0, np.uint8, copy=True, order="C")
        b = np.array(w1, np.uint8, copy=True, order="C")
        size = np.zeros(2, np.int32)
        self.o.orc_blend_canvas_size(a.shape[1], a.shape[0], b.shape[1], b.shape[0], dx, dy, _p(size))
        canvas = np.zeros((int(size[1]), int(size[0]), 3), np.uint8)
        self.o.orc_blend_compose(_p(a), a.shape[1], a.shape[0], _p(b), b.shape[1], b.shape[0], dx, dy, _p(canvas),
                                 int(size[0]), int(size[1]))
        return canvas, a, b

    def warp_perspective(self, src_rgb, H, dst_rgb):
        """_WarpPerspective (stitch.cpp:106-141); returns the updated copy of dst_rgb."""
        src = np.ascontiguousarray(src_rgb, np.uint8)
        dst = np.array(dst_rgb, np.uint8, copy=True, order="C")
        Hm = np.ascontiguousarray(H, np.float32).reshape(9)
        self.o.orc_warp_perspective(_p(src), src.shape[1], src.shape[0], _p(Hm), _p(dst), dst.shape[1], dst.shape[0])
        return dst

    def warp_cylinder(self, src_rgb, focal):
        """warping (stitch.cpp:178-221) for one image: (out, upedge, downedge)."""
        src = np.ascontiguousarray(src_rgb, np.uint8)
        h, w = src.shape[:2]
        out = np.zeros_like(src)
        up = np.zeros((w, 2), np.float32)
        down = np.zeros((w, 2), np.float32)
        self.o.orc_warp_cylinder(_p(src), w, h, focal, _p(out), _p(up), _p(down))
        return out, up, down

    def gau_matrix(self, p, mat_len=0):
        buf = np.zeros(8192, np.float32)
        n = self.o.orc_gau_matrix(p, mat_len, _p(buf), buf.size)
        assert n > 0
        return buf[:n].copy()

    def gau_blur(self, img, p):
        img = np.ascontiguousarray(img, np.float32)
        out = np.empty_like(img)
        self.o.orc_gau_blur(_p(img), _p(out), img.shape[1], img.shape[0], p)
        return out

    def first(self, img, ratio):
        img = np.ascontiguousarray(img, np.float32)
        h, w = img.shape
        ow, oh = C.c_int(), C.c_int()
        self.o.orc_first_size(w, h, ratio, C.byref(ow), C.byref(oh))
        out = np.empty((oh.value, ow.value), np.float32)
        self.o.orc_first(_p(img), w, h, ratio, _p(out))
        return out

    def fast_atan2f(self, dy, dx):
        return self.o.orc_fast_atan2f(dy, dx)

    def convert_gray(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32)
        out = np.empty(rgb.shape[:-1], np.float32)
        self.o.orc_convert_gray(_p(rgb), out.size, _p(out))
        return out

    def pyramid_octaves(self, w, h):
        return self.o.orc_pyramid_octaves(w, h)

    def octave_levels(self, gray, octave, intvls=3):
        gray = np.ascontiguousarray(gray, np.float32)
        h, w = gray.shape
        size = np.float32(2.0) / np.float32(2.0 ** octave)
        ow, oh = int(np.float32(w) * size), int(np.float32(h) * size)
        gau = np.empty((intvls + 3, oh, ow), np.float32)
        dog = np.empty((intvls + 2, oh, ow), np.float32)
        cw, ch = C.c_int(), C.c_int()
        self.o.orc_octave_levels(_p(gray), w, h, octave, intvls, _p(gau), _p(dog), C.byref(cw), C.byref(ch))
        assert (cw.value, ch.value) == (ow, oh)
        return gau, dog

    def extract(self, gray, octaves=0, intvls=3, cap=1 << 16, describe=True):
        f, h, _ = self.extract_ex(gray, octaves, intvls, cap, describe)
        return f, h

    def extract_ex(self, gray, octaves=0, intvls=3, cap=1 << 16, describe=True, subpixel=0, contr_thr=0.05):
        """-> (features, clamped histograms, sub-pixel offsets n x 3 = (xc, xr, xi))."""
        gray = np.ascontiguousarray(gray, np.float32)
        h, w = gray.shape
        while True:
            out = np.zeros(cap, FEAT_DTYPE)
            hist = np.zeros((cap, 128), np.float32)
            sub = np.zeros((cap, 3), np.float32)
            n = self.o.orc_extract_ex(_p(gray), w, h, octaves, intvls, _p(out), cap, _p(hist), int(describe),
                                      int(subpixel), C.c_double(contr_thr), _p(sub))
            if n > cap:
                cap = n
                continue
            return out[:n], hist[:n], sub[:n]

    def extract_quality(self, gray, octaves=0, intvls=3, cap=1 << 16, contr_thr=0.05):
        """The quality mode (params.textbook = 1 of the product): -> (features, normalised histograms, offsets)."""
        gray = np.ascontiguousarray(gray, np.float32)
        h, w = gray.shape
        while True:
            out = np.zeros(cap, FEAT_DTYPE)
            hist = np.zeros((cap, 128), np.float32)
            sub = np.zeros((cap, 3), np.float32)
            n = self.o.orc_extract_quality(_p(gray), w, h, octaves, intvls, _p(out), cap, _p(hist),
                                           C.c_double(contr_thr), _p(sub))
            if n > cap:
                cap = n
                continue
            return out[:n], hist[:n], sub[:n]

    def force_match(self, q, r, th):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 128)
        r = np.ascontiguousarray(r, np.float32).reshape(-1, 128)
        best = np.zeros(len(q), np.int32)
        d1 = np.zeros(len(q), np.float32)
        d2 = np.zeros(len(q), np.float32)
        self.o.orc_force_match(_p(q), len(q), _p(r), len(r), th, _p(best), _p(d1), _p(d2))
        return best, d1, d2

    def match_u8(self, q, r, th, ratio_form=0):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 128)
        r = np.ascontiguousarray(r, np.uint8).reshape(-1, 128)
        best = np.zeros(len(q), np.int32)
        d1 = np.zeros(len(q), np.int32)
        d2 = np.zeros(len(q), np.int32)
        self.o.orc_match_u8(_p(q), len(q), _p(r), len(r), th, ratio_form, _p(best), _p(d1), _p(d2))
        return best, d1, d2


class Reference:
    """The reference's own imglib / kd-tree object code; exists only where
    /root/reference was mounted at build time."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        r = C.CDLL(REF_SO)
        vp = C.c_void_p
        r.ref_gau_matrix.argtypes = [C.c_float, C.c_int, vp, C.c_int]
        r.ref_gau_blur.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float]
        r.ref_first.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp]
        r.ref_kdtree_match.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp]
        self.r = r

    def gau_matrix(self, p, mat_len=0):
        buf = np.zeros(8192, np.float32)
        n = self.r.ref_gau_matrix(p, mat_len, _p(buf), buf.size)
        return buf[:n].copy()

    def gau_blur(self, img, p):
        img = np.ascontiguousarray(img, np.float32)
        out = np.empty_like(img)
        self.r.ref_gau_blur(_p(img), _p(out), img.shape[1], img.shape[0], p)
        return out

    def first(self, img, ratio):
        img = np.ascontiguousarray(img, np.float32)
        h, w = img.shape
        padded = np.vstack([img, np.full((1, w), np.nan, np.float32)])
        nh = int(np.floor(np.float32(h) * np.float32(ratio)))
        nw = int(np.floor(np.float32(w) * np.float32(ratio)))
        out = np.empty((nh, nw), np.float32)
        self.r.ref_first(_p(padded), w, h, ratio, _p(out))
        return out

    def kdtree_match(self, q, r, th, max_chks=200):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 128)
        r = np.ascontiguousarray(r, np.float32).reshape(-1, 128)
        best = np.zeros(len(q), np.int32)
        self.r.ref_kdtree_match(_p(q), len(q), _p(r), len(r), th, max_chks, _p(best))
        return best
