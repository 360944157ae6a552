"""CPU tests: the oracle against the committed golden vectors.

imglib_ref.npz was produced by the reference's own imglib.cpp object code, so
these comparisons pin the oracle's Gaus::gau_matrix / Gaus::GauBlur /
Scaling::first restatement to the reference bit for bit.  oracle_extract.npz was
produced by the oracle itself (Sift.cpp / Imgraw.cpp do not build here): those
checks guard against regressions only -- parity with the reference for
detection, orientation, descriptors and fastAtan2f is UNPINNED.
"""
import os

import numpy as np
import pytest

from tests.oracle_lib import REF_SO, Oracle, Reference

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def test_taps_match_reference_bits(orc):
    z = np.load(os.path.join(G, "imglib_ref.npz"))
    for i, (p, ml) in enumerate(zip(z["tap_p"], z["tap_len"])):
        want = z["taps"][z["tap_off"][i]: z["tap_off"][i + 1]]
        got = orc.gau_matrix(float(p), int(ml))
        assert got.shape == want.shape, (p, ml)
        assert np.array_equal(bits(got), bits(want)), (p, ml)


def test_tap_lengths_are_the_survey_table(orc):
    # kernel length = (int)round(6p+1) | 1, imglib.cpp:132
    lens = [len(orc.gau_matrix(float(np.float32(1.6) * np.float32(np.sqrt(np.float32(2.0))) ** i), 0)) for i in range(6)]
    assert lens == [11, 15, 21, 29, 39, 55]


def test_blur_matches_reference_bits(orc):
    z = np.load(os.path.join(G, "imglib_ref.npz"))
    for k in (0, 1):
        for j, p in enumerate(z["blur_p"]):
            got = orc.gau_blur(z[f"blur_in{k}"], float(p))
            assert np.array_equal(bits(got), bits(z[f"blur_out{k}"][j])), (k, p)


def test_first_matches_reference_bits(orc):
    z = np.load(os.path.join(G, "imglib_ref.npz"))
    for i, ratio in enumerate(z["first_ratio"]):
        got = orc.first(z["first_in"], float(ratio))
        assert np.array_equal(bits(got), bits(z[f"first_out{i}"])), ratio


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref only exists where the reference is mounted")
def test_oracle_equals_live_reference_objects(orc):
    ref = Reference()
    rng = np.random.default_rng(99)
    for shape, p in [((31, 47), 1.6), ((64, 33), 6.4), ((20, 90), 12.8)]:
        img = rng.random(shape, dtype=np.float32)
        assert np.array_equal(bits(orc.gau_blur(img, p)), bits(ref.gau_blur(img, p)))
    img = rng.random((53, 71), dtype=np.float32)
    for z in (2.0, 1.0, 0.5, 0.25):
        a, b = orc.first(img, z), ref.first(img, z)
        assert not np.isnan(b).any()  # the reference never touched the row past the image
        assert np.array_equal(bits(a), bits(b))


def test_octave_count(orc):
    # max(1, ceil(log2(max(w,h))) - 5), Sift.cpp:130-139
    assert [orc.pyramid_octaves(*s) for s in [(640, 480), (1920, 1080), (3840, 2160), (8192, 8192), (32, 32)]] == \
        [5, 6, 7, 8, 1]


def test_extract_regression(orc):
    z = np.load(os.path.join(G, "oracle_extract.npz"))
    img = (z["img_a"].astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    f, h = orc.extract(img)
    want = z["feat_a"]
    assert len(f) == len(want) == 102
    for k in ("size", "kai", "sigmaOCT", "x", "y", "sita"):
        assert np.array_equal(f[k], want[k])
    assert np.array_equal(bits(f["mm"]), bits(want["mm"]))
    assert np.array_equal(f["descr"], want["descr"])
    assert np.array_equal(bits(h), bits(z["hist_a"]))
    # quirks the restatement must keep: descriptor = 512 * min(hist, 0.2) -> at most 102,
    # sigmaOCT integer-truncated, orientation a multiple of 10
    assert f["descr"].max() == 102
    assert set(np.unique(f["sigmaOCT"])) <= {2.0, 3.0, 4.0, 6.0, 9.0, 12.0, 18.0}
    assert (f["sita"] % 10 == 0).all()
    # canonical order: octave (size descending), layer, row, column, bin
    key = np.stack([-f["size"], f["kai"], f["y"], f["x"], f["sita"]], axis=1)
    assert (np.lexsort(key.T[::-1]) == np.arange(len(f))).all()


def test_fast_atan2f_regression(orc):
    z = np.load(os.path.join(G, "oracle_extract.npz"))
    got = np.array([orc.fast_atan2f(float(y), float(x)) for y, x in z["atan_in"]], np.float32)
    assert np.array_equal(bits(got), bits(z["atan_out"]))
    # against libm: the polynomial is good to ~0.01 degree
    ref = np.degrees(np.arctan2(z["atan_in"][:, 0].astype(np.float64), z["atan_in"][:, 1].astype(np.float64))) % 360.0
    err = np.abs(((got - ref + 180.0) % 360.0) - 180.0)
    assert err.max() < 0.02


def test_match_regression_and_forms(orc):
    z = np.load(os.path.join(G, "oracle_extract.npz"))
    q, r = z["feat_b"]["descr"], z["feat_a"]["descr"]
    best, d1, d2 = orc.force_match(q.astype(np.float32), r.astype(np.float32), 0.5)
    assert np.array_equal(best, z["force_best"]) and np.array_equal(d1, z["force_d1"])
    bu, d1u, d2u = orc.match_u8(q, r, 0.5, 1)
    # integer distances equal forceMatchFeat's float sums exactly (all below 2^24)
    assert np.array_equal(d1u.astype(np.float32), d1) and np.array_equal(d2u.astype(np.float32), d2)
    assert np.array_equal(bu, best)
    b0, _, _ = orc.match_u8(q, r, 0.5, 0)
    assert np.array_equal(b0, z["u8_best"])


def test_match_edge_cases(orc):
    q = np.zeros((3, 128), np.uint8)
    r = np.zeros((1, 128), np.uint8)
    r[0, 0] = 5
    b0, d1, d2 = orc.match_u8(q, r, 0.5, 0)
    assert (b0 == -1).all() and (d1 == 25).all() and (d2 == np.iinfo(np.int32).max).all()  # fewer than 2 refs
    b1, _, _ = orc.match_u8(q, r, 0.5, 1)
    assert (b1 == 0).all()  # forceMatchFeat's form divides by FLT_MAX and accepts
    bq, _, _ = orc.match_u8(np.zeros((0, 128), np.uint8), r, 0.5, 0)
    assert len(bq) == 0


def test_kdtree_fixture_agrees_with_exact_nn(orc):
    z = np.load(os.path.join(G, "kdtree_ref.npz"))
    b, _, _ = orc.match_u8(z["q"], z["r"], float(z["th"]), 0)
    agree = (b == z["kd_best"]).mean()
    print("kd-tree (reference object code) vs exact NN agreement: %.4f" % agree)
    assert agree > 0.95


def test_subpixel_refinement_locates_an_off_grid_blob(orc):
    """interp_extremum (matrix.hpp:140-196) is dead code in the reference, so no
    vector of the reference can pin it (PARITY UNPINNED); what can be checked is
    that the restatement does what the method is for: it finds the centre of a
    Gaussian blob placed off the pixel grid.  Octave 0 is the image resampled with
    aligned corners (imglib.cpp:236-250), so octave pixel u lies at u*(W-1)/(2W-1)."""
    H = W = 160
    cx, cy = 80.3, 79.8
    yy, xx = np.mgrid[0:H, 0:W]
    img = (0.2 + 0.6 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 3.0 ** 2))).astype(np.float32)
    f, _, sub = orc.extract_ex(img, octaves=2, subpixel=1, contr_thr=0.03)
    assert len(f) >= 1 and (f["size"] == 2.0).all()
    assert (np.abs(sub) < 0.5).all()
    scale = (W - 1) / (2 * W - 1)
    px = (f["x"] + sub[:, 0]) * scale
    py = (f["y"] + sub[:, 1]) * scale
    assert np.abs(px - cx).max() < 0.02 and np.abs(py - cy).max() < 0.02
    # without the stage the same keypoint sits on the integer grid, up to half an octave pixel away
    f0, _ = orc.extract(img, octaves=2)
    assert len(f0) >= len(f)
    assert np.abs(f0["x"] * scale - cx).min() < 0.5 * scale + 0.02


def test_subpixel_only_drops_or_moves_keypoints(orc):
    img = __import__("sift_amd").blob_image(320, 240, seed=2)
    f0, _ = orc.extract(img, describe=False)
    f1, _, s1 = orc.extract_ex(img, describe=False, subpixel=1, contr_thr=0.0)
    f2, _, _ = orc.extract_ex(img, describe=False, subpixel=1, contr_thr=0.05)
    assert len(f2) < len(f1) <= len(f0)
    assert (np.abs(s1) < 0.5).all()


def test_stitch_warp_restatement_sanity(oracle):
    """oracle/stitch_oracle.cpp (parity unpinned): identity homography copies the interior and
    blacks the last row/column (stitch.cpp:39); the cylinder leaves the centre column in place."""
    rng = np.random.default_rng(0)
    src = rng.integers(1, 255, (40, 60, 3)).astype(np.uint8)
    got = oracle.warp_perspective(src, [1, 0, 0, 0, 1, 0, 0, 0, 1], np.full((40, 60, 3), 9, np.uint8))
    assert np.array_equal(got[:39, :59], src[:39, :59]) and (got[39] == 0).all() and (got[:, 59] == 0).all()
    shifted = oracle.warp_perspective(src, [1, 0, 5, 0, 1, 0, 0, 0, 1], np.zeros((40, 80, 3), np.uint8))
    assert np.array_equal(shifted[:39, 5:64], src[:39, :59]) and (shifted[:, :5] == 0).all()
    out, up, down = oracle.warp_cylinder(src, 100.0)
    assert np.array_equal(out[:39, 30], src[:39, 30])  # k = 1 on the axis
    assert up.shape == (60, 2) and up[30, 0] == 30.0 and down[30, 1] > up[30, 1]
    assert (out[:, 0] == 0).all()  # the outermost columns sample outside the image


def test_stitch_golden_regression_guard(oracle):
    """tests/golden/stitch_oracle.npz (made by oracle/make_golden_stitch.py from the oracles
    themselves: a regression guard, not a pin to the reference): the RANSAC restatement, the
    CPU ransac_xform of the product and the warp restatement still give these results."""
    import ctypes as C
    import sys

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ransac_oracle

    g = np.load(os.path.join(ROOT, "tests", "golden", "stitch_oracle.npz"))
    xyuv = g["xyuv"]
    H, fin, in_min = ransac_oracle.ransac_xform(xyuv[:, :2], xyuv[:, 2:], 4, 0.005, 3.0, int(g["ransac_seed"]))
    assert np.array_equal(fin, g["ransac_inliers"]) and in_min == int(g["ransac_in_min"])
    np.testing.assert_allclose(H, g["ransac_H"], rtol=1e-6)
    # the product's CPU implementation on the same pairs (integer coordinates, scale 1)
    host = C.CDLL(os.path.join(ROOT, "sift_amd", "lib", "libsift_host.so"))
    vp = C.c_void_p
    host.sift_host_ransac.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_float,
                                      C.c_uint, vp, vp]
    n = len(xyuv)
    x2, y2 = xyuv[:, 0].astype(np.int32), xyuv[:, 1].astype(np.int32)
    x1, y1 = xyuv[:, 2].astype(np.int32), xyuv[:, 3].astype(np.int32)
    one = np.ones(n, np.float32)
    match = np.arange(n, dtype=np.int32)
    Hh = np.zeros(9, np.float32)
    idx = np.full(n, -1, np.int32)
    p = lambda a: a.ctypes.data  # noqa: E731
    n_in = host.sift_host_ransac(p(x2), p(y2), p(one), n, p(x1), p(y1), p(one), n, p(match), 4, 0.005, 3.0,
                                 int(g["ransac_seed"]), p(Hh), p(idx))
    assert n_in == len(g["ransac_inliers"]) and np.array_equal(idx[:n_in], g["ransac_inliers"])
    np.testing.assert_allclose(Hh, g["ransac_H"], rtol=2e-4, atol=1e-6)
    assert np.array_equal(oracle.warp_perspective(g["src"], g["Hw"], g["canvas"]), g["warped"])
    cyl, up, down = oracle.warp_cylinder(g["src"], float(g["focal"]))
    assert np.array_equal(cyl, g["cyl"]) and np.array_equal(up, g["cyl_up"]) and np.array_equal(down, g["cyl_down"])


def test_blend_oracle_regression_fixture():
    """oracle/blend_oracle.cpp (multiBandBlend, alignMatch, getFocal restated; parity unpinned) still gives
    what it gave when tests/golden/stitch_oracle.npz was generated."""
    import os

    from tests.oracle_lib import Oracle

    orc = Oracle()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stitch_oracle.npz"))
    lo, ro = orc.multiband_blend(g["blend_in_l"], g["blend_in_r"], int(g["blend_dx"]), int(g["blend_dy"]))
    assert np.array_equal(lo, g["blend_l"]) and np.array_equal(ro, g["blend_r"])
    assert (lo != g["blend_in_l"]).sum() > 1000
    f = np.float32(orc.get_focal(g["blend_H"], 120 * 72 * 3, 120 * 72 * 3))
    assert f == g["blend_focal"]
    assert list(orc.align_match((120, 72), (120, 72), g["blend_a"], g["blend_b"], float(f))) == list(g["blend_align"])
    assert orc.get_focal(None, 10, 30) == 20.0  # no homography: the mean of the two sizes
