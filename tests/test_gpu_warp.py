"""The stitcher's two warps (sift_hip_warp_perspective, sift_hip_warp_cylinder) through the
C ABI against the CPU restatement of stitch.cpp in oracle/stitch_oracle.cpp: byte-exact.
Parity with the reference itself is unpinned for this step (stitch.cpp needs a header this
environment lacks and the reference holds no vectors for it)."""
import ctypes as C

import numpy as np
import pytest

import sift_amd

pytestmark = pytest.mark.gpu


def rgb_image(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], axis=-1)
    noise = rng.integers(0, 64, (h, w, 3))
    img = np.clip(base + noise, 0, 255).astype(np.uint8)
    img[rng.random((h, w)) < 0.01] = 0  # black pixels: the blender keys on them
    return img


HS = {
    "identity": [1, 0, 0, 0, 1, 0, 0, 0, 1],
    "shift": [1, 0, 37.5, 0, 1, -12.25, 0, 0, 1],
    "persp": [0.98, 0.05, 40.0, -0.04, 1.02, 15.0, 1.5e-5, -1.0e-5, 1],
    "strong": [1.2, -0.2, -30.0, 0.15, 0.9, 25.0, 4e-4, 2e-4, 1],
    "zero": [0, 0, 0, 0, 0, 0, 0, 0, 0],
}


@pytest.mark.parametrize("name", list(HS))
@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 1024, 528), (333, 127, 531, 139), (64, 48, 61, 50), (5, 4, 9, 7)])
def test_perspective_bytes_equal(sift, oracle, name, sw, sh, dw, dh):
    src = rgb_image(sw, sh, 1)
    dst0 = rgb_image(dw, dh, 2)  # in/out: pixels that map outside the source keep these values
    H = np.array(HS[name], np.float32)
    want = oracle.warp_perspective(src, H, dst0)
    got = sift.warp_perspective(src, H, dst0.copy())
    assert np.array_equal(got, want)
    if name == "zero":
        assert np.array_equal(got, dst0)  # 0/0 everywhere: nothing is written
    elif name in ("identity", "shift", "persp") and sw >= 64:
        assert (got != dst0).any()


def test_perspective_identity_is_a_copy_of_the_interior(sift):
    src = rgb_image(200, 100, 3)
    got = sift.warp_perspective(src, HS["identity"], np.full((100, 200, 3), 7, np.uint8))
    assert np.array_equal(got[:99, :199], src[:99, :199])
    assert (got[99] == 0).all() and (got[:, 199] == 0).all()  # 2x2 neighbourhood leaves the image: black


@pytest.mark.parametrize("w,h,f", [(640, 480, 500.0), (640, 480, 3000.0), (333, 127, 200.0), (7, 5, 10.0), (4, 1, 3.0),
                                   (1, 3, 2.0)])
def test_cylinder_bytes_equal(sift, oracle, w, h, f):
    src = rgb_image(w, h, 4)
    want, up_w, down_w = oracle.warp_cylinder(src, f)
    got, up, down = sift.warp_cylinder(src, f)
    assert np.array_equal(got, want)
    assert np.array_equal(up.view(np.uint32), up_w.view(np.uint32))
    assert np.array_equal(down.view(np.uint32), down_w.view(np.uint32))


def test_full_size_4k(sift, oracle):
    src = rgb_image(3840, 2160, 5)
    dst0 = np.zeros((int(2160 * 1.1), int(3840 * 1.6), 3), np.uint8)  # the canvas blen2img makes (Blend.cpp:920)
    H = np.array([1.01, 0.02, 900.0, -0.015, 0.99, 40.0, 2e-6, -1e-6, 1], np.float32)
    got = sift.warp_perspective(src, H, dst0.copy())
    assert np.array_equal(got, oracle.warp_perspective(src, H, dst0))
    assert (got.reshape(-1, 3).any(axis=1)).mean() > 0.4
    got, up, down = sift.warp_cylinder(src, 4000.0)
    want, up_w, down_w = oracle.warp_cylinder(src, 4000.0)
    assert np.array_equal(got, want) and np.array_equal(up, up_w) and np.array_equal(down, down_w)


def test_arguments(sift):
    src = rgb_image(32, 16, 6)
    dst = np.zeros((16, 32, 3), np.uint8)
    H = np.array(HS["identity"], np.float32)
    lib, vp, E = sift.lib, C.c_void_p, sift_amd.capi
    hp = H.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.sift_hip_warp_perspective(sift.h, None, 32, 16, hp, vp(dst.ctypes.data), 32, 16) == E.ERR_ARG
    assert lib.sift_hip_warp_perspective(sift.h, vp(src.ctypes.data), 32, 16, None, vp(dst.ctypes.data), 32, 16) == E.ERR_ARG
    assert lib.sift_hip_warp_perspective(sift.h, vp(src.ctypes.data), 0, 16, hp, vp(dst.ctypes.data), 32, 16) == E.ERR_ARG
    assert lib.sift_hip_warp_perspective(sift.h, vp(src.ctypes.data), 32, 16, hp, vp(src.ctypes.data), 32, 16) == E.ERR_ARG
    assert lib.sift_hip_warp_cylinder(sift.h, vp(src.ctypes.data), 32, 16, 0.0, vp(dst.ctypes.data), None, None) == E.ERR_ARG
    assert lib.sift_hip_warp_cylinder(sift.h, vp(src.ctypes.data), 32, -1, 5.0, vp(dst.ctypes.data), None, None) == E.ERR_ARG
    # edges are optional
    assert lib.sift_hip_warp_cylinder(sift.h, vp(src.ctypes.data), 32, 16, 50.0, vp(dst.ctypes.data), None, None) == 0


def test_golden_fixture(sift):
    """The GPU against tests/golden/stitch_oracle.npz (oracle outputs; regression guard)."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stitch_oracle.npz"))
    assert np.array_equal(sift.warp_perspective(g["src"], g["Hw"], g["canvas"].copy()), g["warped"])
    cyl, up, down = sift.warp_cylinder(g["src"], float(g["focal"]))
    assert np.array_equal(cyl, g["cyl"]) and np.array_equal(up, g["cyl_up"]) and np.array_equal(down, g["cyl_down"])
    H, idx, _ = sift.ransac_homog_pairs(g["xyuv"], seed=int(g["ransac_seed"]))
    assert np.array_equal(idx, g["ransac_inliers"])
    np.testing.assert_allclose(H.reshape(-1), g["ransac_H"], rtol=2e-4, atol=1e-6)
