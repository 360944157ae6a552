"""GPU parity: every stage of the HIP path against the CPU oracle, through the C ABI.

Bars: bit-exact for the pyramid (blur_mode 0), keypoint records (position, layer,
orientation bin, mm) and match indices/distances; descriptors within L2 1e-3 of
the oracle's clamped histogram (hardware-rate atan2/exp/sqrt and fixed-point bin
sums instead of sequential float additions), bytes within +-1.
"""
import ctypes as C

import numpy as np
import pytest

import sift_amd
from sift_amd import SiftParams

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_library_is_hip_and_device_is_gfx950(sift):
    assert sift.h


@pytest.mark.parametrize("shape,p", [((61, 97), 1.6), ((48, 64), 3.2), ((150, 200), 4.525), ((17, 33), 9.05),
                                     ((70, 1100), 2.2627418), ((300, 40), 25.6), ((5, 3), 1.6)])
def test_gau_blur_bit_exact(sift, oracle, shape, p):
    rng = np.random.default_rng(7)
    img = rng.random(shape, dtype=np.float32)
    got = sift.gau_blur(img, p)
    want = oracle.gau_blur(img, p)
    assert np.array_equal(bits(got), bits(want))


def test_gau_blur_long_kernel_uses_global_column_pass(sift, oracle):
    # sigma 72.4 -> 435 taps: wider than the LDS-tiled column pass handles
    rng = np.random.default_rng(8)
    img = rng.random((96, 80), dtype=np.float32)
    assert np.array_equal(bits(sift.gau_blur(img, 72.4)), bits(oracle.gau_blur(img, 72.4)))


def test_gau_blur_very_long_kernel_uses_global_row_pass(sift, oracle):
    # sigma 2500 -> 15001 taps, as in octave 8+ of a 16384^2 image: too long for an LDS run
    rng = np.random.default_rng(12)
    img = rng.random((30, 40), dtype=np.float32)
    assert np.array_equal(bits(sift.gau_blur(img, 2500.0)), bits(oracle.gau_blur(img, 2500.0)))


def test_gau_blur_fast_mode_close(sift, oracle):
    rng = np.random.default_rng(9)
    img = rng.random((120, 160), dtype=np.float32)
    got = sift.gau_blur(img, 3.2, blur_mode=1)
    want = oracle.gau_blur(img, 3.2)
    assert np.max(np.abs(got - want)) < 2e-6


def test_gau_blur_same_buffer_is_an_error(sift):
    import ctypes as C

    a = np.zeros((8, 8), np.float32)
    rc = sift.lib.sift_hip_gau_blur(sift.h, C.c_void_p(a.ctypes.data), C.c_void_p(a.ctypes.data), 8, 8, 1.6, 0, 0)
    assert rc == sift_amd.capi.ERR_SAME


@pytest.mark.parametrize("shape", [(48, 64), (61, 97), (257, 333)])
@pytest.mark.parametrize("ratio", [2.0, 1.0, 0.5, 0.25, 0.125])
def test_first_bit_exact(sift, oracle, shape, ratio):
    rng = np.random.default_rng(3)
    img = rng.random(shape, dtype=np.float32)
    assert np.array_equal(bits(sift.first(img, ratio)), bits(oracle.first(img, ratio)))


def test_convert_gray_bit_exact(sift, oracle):
    rng = np.random.default_rng(4)
    rgb = rng.random((37, 53, 3), dtype=np.float32)
    assert np.array_equal(bits(sift.convert_gray(rgb)), bits(oracle.convert_gray(rgb)))


@pytest.mark.parametrize("octave", [0, 1, 2])
def test_octave_levels_bit_exact(sift, oracle, octave):
    img = sift_amd.blob_image(256, 192, seed=1)
    g, d = sift.octave_levels(img, octave)
    go, do = oracle.octave_levels(img, octave)
    assert np.array_equal(bits(g), bits(go))
    assert np.array_equal(bits(d), bits(do))


def check_features(f, h, fo, ho):
    assert len(f) == len(fo)
    for k in ("size", "kai", "sigmaOCT", "x", "y", "sita"):
        assert np.array_equal(f[k], fo[k]), k
    assert np.array_equal(bits(f["mm"]), bits(fo["mm"]))
    # descriptors: L2 of the clamped histogram within 1e-3, bytes within +-1
    l2 = np.sqrt(((h.astype(np.float64) - ho) ** 2).sum(axis=1))
    assert l2.max() < 1e-3, l2.max()
    diff = np.abs(f["descr"].astype(np.int16) - fo["descr"].astype(np.int16))
    assert diff.max() <= 1
    assert (diff != 0).mean() < 1e-3


@pytest.mark.parametrize("w,h,seed", [(256, 192, 1), (320, 200, 5), (640, 480, 1)])
def test_extract_matches_oracle(sift, oracle, w, h, seed):
    img = sift_amd.blob_image(w, h, seed=seed)
    f, hist = sift.extract(img, want_hist=True)
    fo, ho = oracle.extract(img)
    assert len(fo) > 50
    check_features(f, hist, fo, ho)


def test_extract_fixed_octaves_is_prefix_of_auto(sift, oracle):
    img = sift_amd.blob_image(320, 240, seed=2)
    f2, _ = sift.extract(img, SiftParams(octaves=2))
    fo, _ = oracle.extract(img, octaves=2)
    assert len(f2) == len(fo)
    assert np.array_equal(f2["x"], fo["x"]) and np.array_equal(f2["sita"], fo["sita"])


def test_extract_rgb_goes_through_convert_gray(sift, oracle):
    rng = np.random.default_rng(11)
    g = sift_amd.blob_image(256, 192, seed=3)
    rgb = np.stack([g, np.roll(g, 3, 1), np.roll(g, 5, 0)], axis=-1)
    rgb = (rgb * rng.uniform(0.9, 1.0)).astype(np.float32)
    f, h = sift.extract(rgb, want_hist=True)
    fo, ho = oracle.extract(oracle.convert_gray(rgb))
    check_features(f, h, fo, ho)


def test_extract_nospc_reports_needed_capacity(sift):
    import ctypes as C

    img = sift_amd.blob_image(256, 192, seed=1)
    out = np.zeros(10, sift_amd.FEAT_DTYPE)
    n = C.c_int(0)
    prm = SiftParams()._c()
    rc = sift.lib.sift_hip_extract(sift.h, C.c_void_p(img.ctypes.data), 256, 192, 1, C.byref(prm),
                                   C.c_void_p(out.ctypes.data), 10, C.byref(n), None)
    assert rc == sift_amd.capi.ERR_NOSPC and n.value > 10
    full, _ = sift.extract(img)
    assert n.value == len(full)
    assert np.array_equal(out["x"], full["x"][:10])


def test_extract_flat_image_has_no_features(sift):
    f, _ = sift.extract(np.full((96, 128), 0.5, np.float32))
    assert len(f) == 0


def test_extract_batch_equals_single_frames(sift):
    frames = np.stack([sift_amd.blob_image(256, 192, seed=s) for s in (1, 2, 3)])
    out, n = sift.extract_batch(frames, cap=4096)
    for i in range(3):
        f, _ = sift.extract(frames[i])
        assert n[i] == len(f)
        assert np.array_equal(out[i, : n[i]]["descr"], f["descr"])
        assert np.array_equal(out[i, : n[i]]["x"], f["x"])


@pytest.mark.parametrize("nq,nr", [(1, 1), (5, 2), (64, 64), (257, 129), (1000, 3000), (300, 1), (10, 0), (0, 10),
                                   (2049, 20001)])
@pytest.mark.parametrize("form", [0, 1])
def test_match_bit_exact(sift, oracle, nq, nr, form):
    q, r = sift_amd.planted_descriptors(nq, nr, vmax=255, plant=0.2)
    best, d1, d2 = sift.match_u8(q, r, 0.5, form)
    bo, d1o, d2o = oracle.match_u8(q, r, 0.5, form)
    assert np.array_equal(d1, d1o)
    assert np.array_equal(d2, d2o)
    assert np.array_equal(best, bo)


def test_match_ties_pick_lowest_index(sift, oracle):
    q, r = sift_amd.planted_descriptors(40, 600, vmax=102)
    r[300] = r[7]
    r[555] = r[7]
    q[0] = r[7]
    r[513] = r[299]
    q[1] = r[299]
    q[1, 5] ^= 1  # distance 1 to both copies: accepted at ratio 2, index must be the lower one
    best, d1, d2 = sift.match_u8(q, r, 0.9, 0)
    bo, d1o, d2o = oracle.match_u8(q, r, 0.9, 0)
    assert d1[0] == 0 and d2[0] == 0
    assert np.array_equal(best, bo) and np.array_equal(d1, d1o) and np.array_equal(d2, d2o)
    bf, d1f, d2f = sift.match_u8(q, r, 2.0, 0)  # accept everything: index of the first minimum
    bof, _, _ = oracle.match_u8(q, r, 2.0, 0)
    assert bf[1] == 299 and np.array_equal(bf, bof)


def test_match_on_extracted_descriptors_equals_forcematch(sift, oracle):
    a = sift_amd.blob_image(320, 240, seed=1)
    b = sift_amd.blob_image(320, 240, seed=1, shift_x=-12)
    fa, _ = sift.extract(a)
    fb, _ = sift.extract(b)
    best, d1, d2 = sift.match_u8(fb["descr"], fa["descr"], 0.5, 1)
    bo, d1o, d2o = oracle.force_match(fb["descr"].astype(np.float32), fa["descr"].astype(np.float32), 0.5)
    assert np.array_equal(best, bo)
    assert (best >= 0).sum() > 20
    # matched keypoints are the same scene points, 12 px apart at full resolution
    m = best >= 0
    dx = fb["x"][m] / fb["size"][m] - fa["x"][best[m]] / fa["size"][best[m]]
    assert np.median(np.abs(dx + 12)) <= 1.0


def test_profile_counts_launches(sift):
    img = sift_amd.blob_image(256, 192, seed=1)
    sift.profile(True)
    sift.profile_read(reset=True)
    sift.extract(img, SiftParams(octaves=1))
    prof = sift.profile_read(reset=True)
    sift.profile(False)
    # a 512x384 octave is below the large-octave threshold: the small-octave kernels run.
    # Five of the six Gaussian levels are built as images; the last one is evaluated
    # only around top-layer extrema (with the sub-pixel stage on, all six are built).
    assert prof["blur_row_small"][1] == 5 and prof["blur_col_small"][1] == 5 and prof["resample"][1] == 1
    assert prof["blur_row_small"][0] > 0 and prof["blur_row"][1] == 0
    sift.profile(True)
    sift.extract(img, SiftParams(octaves=1, subpixel=1))
    prof = sift.profile_read(reset=True)
    sift.profile(False)
    assert prof["blur_row_small"][1] == 6 and prof["blur_col_small"][1] == 6


def test_lazy_top_level_equals_full_level(sift):
    """The last Gaussian level evaluated only at top-layer extrema must give the
    features of the full-image evaluation (the library's A/B switch builds it).  The cases take the
    check through each of its row sweeps: the burst of 16-byte loads (kernels of up to 62 taps on rows
    with 64 pixels to read), the unrolled sweep (longer kernels: the deeper octaves, a larger sigma),
    and the clamped sweep (octaves narrower than the kernel, where every window hangs over the border)."""
    import os

    cases = [(sift_amd.blob_image(640, 480, seed=3), SiftParams()),
             (sift_amd.blob_image(200, 150, seed=6), SiftParams()),
             (sift_amd.textured_image(131, 97, seed=7), SiftParams()),
             (sift_amd.blob_image(320, 240, seed=8), SiftParams(sigma0=2.0)),
             (sift_amd.blob_image(320, 240, seed=9), SiftParams(sigma0=1.2, intvls=2))]
    got = [sift.extract(img, prm, want_hist=True) for img, prm in cases]
    os.environ["SIFT_HIP_FULL_TOP"] = "1"
    try:
        other = sift_amd.SiftHip(0)
        want = [other.extract(img, prm, want_hist=True) for img, prm in cases]
        other.close()
    finally:
        del os.environ["SIFT_HIP_FULL_TOP"]
    assert (got[0][0]["kai"] == 3).sum() > 20
    for (f, h), (f2, h2), (img, prm) in zip(got, want, cases):
        assert (f["kai"] == prm.intvls).sum() > 0, img.shape  # the top layer has features at all
        assert f.tobytes() == f2.tobytes() and np.array_equal(h, h2), img.shape


def test_octave_placement_on_the_streams_changes_nothing(sift, oracle):
    """Which side stream octave 3 runs on, and whether it leads or follows there, is a scheduling
    choice (sift_hip_set_tuning "o3_place"): every placement must return the same bytes, frames of
    exactly four octaves and frames with deeper ones alike, and those bytes are the oracle's."""
    img = sift_amd.blob_image(800, 600, seed=21)
    fo, _ = oracle.extract(img, octaves=4)
    try:
        for octaves in (4, 0):
            want = None
            for place in (3, 0, 1, 2, 4, 3):
                sift.set_tuning("o3_place", place)
                for _ in range(2):
                    f, _ = sift.extract(img, SiftParams(octaves=octaves))
                    want = f.tobytes() if want is None else want
                    assert f.tobytes() == want, (octaves, place)
            if octaves == 4:
                assert len(f) == len(fo) and all(np.array_equal(f[k], fo[k]) for k in ("x", "y", "kai", "sita", "size"))
    finally:
        sift.set_tuning("o3_place", 3)
    assert sift.lib.sift_hip_set_tuning(sift.h, b"o3_place", 5) == sift_amd.capi.ERR_ARG


# ---- against the committed golden vectors (reference object code outputs) ---------------
def _golden(name):
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def test_gpu_blur_and_first_equal_reference_golden(sift):
    z = _golden("imglib_ref.npz")
    for k in (0, 1):
        for j, p in enumerate(z["blur_p"]):
            assert np.array_equal(bits(sift.gau_blur(z[f"blur_in{k}"], float(p))), bits(z[f"blur_out{k}"][j])), (k, p)
    for i, ratio in enumerate(z["first_ratio"]):
        assert np.array_equal(bits(sift.first(z["first_in"], float(ratio))), bits(z[f"first_out{i}"])), ratio


def test_gpu_extract_equals_golden_features(sift):
    z = _golden("oracle_extract.npz")
    img = (z["img_a"].astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    f, h = sift.extract(img, want_hist=True)
    check_features(f, h, z["feat_a"], z["hist_a"])


def test_gpu_match_agreement_with_reference_kdtree(sift):
    z = _golden("kdtree_ref.npz")
    best, _, _ = sift.match_u8(z["q"], z["r"], float(z["th"]), 0)
    agree = float((best == z["kd_best"]).mean())
    print("exact GPU matcher vs the reference's kd-tree/BBF: agreement %.4f" % agree)
    assert agree > 0.95  # the kd-tree is approximate; the exact matcher is the authority


def test_gau_blur_explicit_length(sift, oracle):
    rng = np.random.default_rng(21)
    img = rng.random((30, 50), dtype=np.float32)
    got = sift.gau_blur(img, 2.0, mat_len=7)
    taps = oracle.gau_matrix(2.0, 7)
    # separable reference arithmetic with the explicit kernel, in numpy float64/float32
    def conv(a, axis):
        idx = np.clip(np.arange(a.shape[axis])[:, None] + np.arange(7)[None, :] - 3, 0, a.shape[axis] - 1)
        a = np.moveaxis(a, axis, 0)
        acc = np.zeros(a.shape, np.float64)
        for k in range(7):
            acc += (a[idx[:, k]] * taps[k]).astype(np.float32).astype(np.float64)
        return np.moveaxis(acc.astype(np.float32), 0, axis)
    want = conv(conv(img, 1), 0)
    assert np.array_equal(bits(got), bits(want))


# ---- optional sub-pixel stage (matrix.hpp:140-196, dead code in the reference) --------
@pytest.mark.parametrize("w,h,seed,thr", [(256, 192, 1, 0.05), (640, 480, 1, 0.05), (320, 200, 5, 0.0)])
def test_subpixel_extract_matches_oracle(sift, oracle, w, h, seed, thr):
    img = sift_amd.blob_image(w, h, seed=seed)
    prm = SiftParams(subpixel=1, contr_thr=thr)
    f, hist = sift.extract(img, prm, want_hist=True)
    sub = sift.last_offsets(len(f))
    fo, ho, so = oracle.extract_ex(img, subpixel=1, contr_thr=thr)
    assert len(fo) > 20
    check_features(f, hist, fo, ho)
    assert np.array_equal(bits(sub), bits(so))
    assert (np.abs(sub) < 0.5).all()
    # refinement only removes or moves keypoints: never more than without it
    f0, _ = sift.extract(img)
    assert len(f) <= len(f0)
    assert len(sift.last_offsets(len(f0))) == 0  # the last call ran without refinement


def test_subpixel_flag_is_validated(sift):
    import ctypes as C

    img = sift_amd.blob_image(64, 64, seed=1)
    out = np.zeros(16, sift_amd.FEAT_DTYPE)
    n = C.c_int(0)
    prm = SiftParams(subpixel=2)._c()
    rc = sift.lib.sift_hip_extract(sift.h, C.c_void_p(img.ctypes.data), 64, 64, 1, C.byref(prm),
                                   C.c_void_p(out.ctypes.data), 16, C.byref(n), None)
    assert rc == sift_amd.capi.ERR_ARG


# ---- other scale-space shapes and ragged sizes ------------------------------------------
@pytest.mark.parametrize("intvls", [1, 2, 4, 5])
def test_extract_other_interval_counts(sift, oracle, intvls):
    # Sift(ImgRaw, intvls) (Sift.hpp:99): pyWidth = intvls + 3 levels per octave
    img = sift_amd.blob_image(200, 168, seed=6)
    f, hist = sift.extract(img, SiftParams(intvls=intvls), want_hist=True)
    fo, ho = oracle.extract(img, intvls=intvls)
    assert len(fo) > 10
    check_features(f, hist, fo, ho)


@pytest.mark.parametrize("w,h", [(101, 99), (150, 77), (77, 150), (131, 67), (193, 65), (33, 200), (64, 64), (2, 2), (7, 5)])
def test_extract_ragged_and_tiny_sizes(sift, oracle, w, h):
    # widths that are not multiples of the 64-pixel tiles / 8-pixel thread runs, and
    # images smaller than every scan window (no features, no fault)
    img = sift_amd.blob_image(w, h, seed=10)
    f, hist = sift.extract(img, want_hist=True)
    fo, ho = oracle.extract(img)
    assert len(f) == len(fo)
    if min(w, h) > 64:
        assert len(fo) > 5
    if len(fo):
        check_features(f, hist, fo, ho)


def test_extract_subpixel_with_other_interval_count(sift, oracle):
    img = sift_amd.blob_image(240, 200, seed=7)
    prm = SiftParams(intvls=4, subpixel=1, contr_thr=0.0)
    f, hist = sift.extract(img, prm, want_hist=True)
    fo, ho, so = oracle.extract_ex(img, intvls=4, subpixel=1, contr_thr=0.0)
    assert len(fo) > 10
    check_features(f, hist, fo, ho)
    assert np.array_equal(bits(sift.last_offsets(len(f))), bits(so))


def test_extract_argument_errors(sift):
    import ctypes as C

    img = sift_amd.blob_image(64, 64, seed=1)
    out = np.zeros(16, sift_amd.FEAT_DTYPE)
    n = C.c_int(0)

    def call(prm, w=64, h=64, ch=1, img_p=img.ctypes.data, out_p=out.ctypes.data, cap=16, n_p=C.byref(n)):
        c = prm._c()
        return sift.lib.sift_hip_extract(sift.h, C.c_void_p(img_p), w, h, ch, C.byref(c), C.c_void_p(out_p), cap, n_p, None)

    E = sift_amd.capi
    assert call(SiftParams(intvls=0)) == E.ERR_ARG
    assert call(SiftParams(intvls=9)) == E.ERR_ARG
    assert call(SiftParams(octaves=25)) == E.ERR_ARG
    assert call(SiftParams(octaves=-1)) == E.ERR_ARG
    assert call(SiftParams(sigma0=0.0)) == E.ERR_ARG
    assert call(SiftParams(), ch=2) == E.ERR_ARG
    assert call(SiftParams(), w=1) == E.ERR_ARG
    assert call(SiftParams(), img_p=0) == E.ERR_ARG
    assert call(SiftParams(), out_p=0) == E.ERR_ARG
    assert call(SiftParams(), cap=-1) == E.ERR_ARG
    assert call(SiftParams(), n_p=None) == E.ERR_ARG
    # octave 0 of a 23171^2 image would not fit the 32-bit pixel index of a level; checked before any work
    assert call(SiftParams(), w=23171, h=23171) == E.ERR_LIMIT
    # the context is still usable afterwards
    f, _ = sift.extract(img)
    assert sift.lib.sift_hip_strerror(E.ERR_LIMIT).decode() != ""
    # cap = 0 with no output buffer: the count comes back with ERR_NOSPC (or OK when there is nothing)
    rc = call(SiftParams(), out_p=0, cap=0)
    assert rc in (E.OK, E.ERR_NOSPC) and n.value == len(f)


# ---- randomised shapes (fixed seeds): tails of every tiling at once ---------------------
def test_match_random_shapes(sift, oracle):
    rng = np.random.default_rng(2024)
    for case in range(24):
        nq = int(rng.integers(1, 700))
        nr = int(rng.integers(1, 2600))
        vmax = int(rng.choice([3, 102, 255]))
        q, r = sift_amd.planted_descriptors(nq, nr, seed_q=10 + case, seed_r=50 + case, vmax=vmax, plant=0.1)
        if case % 5 == 0 and nr > 4:  # exact duplicates among the references: ties on the winner
            r[nr // 2] = r[1]
            r[nr - 1] = r[1]
        form = case & 1
        th = float(rng.choice([0.3, 0.5, 0.8, 1.0]))
        best, d1, d2 = sift.match_u8(q, r, th, form)
        bo, d1o, d2o = oracle.match_u8(q, r, th, form)
        assert np.array_equal(d1, d1o) and np.array_equal(d2, d2o) and np.array_equal(best, bo), (case, nq, nr)


def test_extract_random_shapes(sift, oracle):
    rng = np.random.default_rng(77)
    total = 0
    for case in range(8):
        w = int(rng.integers(70, 330))
        h = int(rng.integers(70, 260))
        img = sift_amd.blob_image(w, h, seed=100 + case)
        prm = SiftParams(octaves=int(rng.integers(0, 4)))
        f, hist = sift.extract(img, prm, want_hist=True)
        fo, ho = oracle.extract(img, octaves=prm.octaves)
        assert len(f) == len(fo), (case, w, h)
        if len(fo):
            check_features(f, hist, fo, ho)
        total += len(fo)
    assert total > 300


def test_blur_and_resample_random_shapes(sift, oracle):
    rng = np.random.default_rng(5)
    for case in range(12):
        w = int(rng.integers(1, 700))
        h = int(rng.integers(1, 200))
        p = float(rng.uniform(0.6, 12.0))
        img = rng.random((h, w), dtype=np.float32)
        assert np.array_equal(bits(sift.gau_blur(img, p)), bits(oracle.gau_blur(img, p))), (case, w, h, p)
        if w >= 2 and h >= 2:
            ratio = float(rng.choice([2.0, 1.0, 0.5, 0.25]))
            if int(w * ratio) >= 1 and int(h * ratio) >= 1:
                assert np.array_equal(bits(sift.first(img, ratio)), bits(oracle.first(img, ratio))), (case, w, h, ratio)


def test_streaming_column_pass_bit_exact_on_ragged_shapes(sift, oracle):
    """The input-stationary column pass (used by itself only on images of 16 Mpixel and more)
    forced onto small and ragged images, with segment lengths that cut columns at awkward rows:
    every tap count it is instantiated for, widths that are not multiples of 64, images shorter
    than the kernel, one-row and one-column images."""
    sift.set_tuning("blur_stream", 1)
    rng = np.random.default_rng(31)
    shapes = [(200, 131), (64, 300), (1, 77), (130, 1), (65, 40), (333, 95)]
    for p, K in ((1.6, 11), (2.2627418, 15), (3.2, 21), (4.5254836, 29), (6.4, 39), (9.0509672, 55)):
        assert len(oracle.gau_matrix(p)) == K
        for i, (w, h) in enumerate(shapes):
            img = rng.random((h, w), dtype=np.float32)
            if i % 2:
                img[rng.random((h, w)) < 0.3] = 0.0  # exact zeros: +0.0 sums must stay +0.0
            want = bits(oracle.gau_blur(img, p))
            for seg in (None, 1, 7, 37, 1000):
                sift.set_tuning("blur_seg", seg or 0)
                assert np.array_equal(bits(sift.gau_blur(img, p)), want), (K, w, h, seg)
    sift.set_tuning("blur_seg", 0)
    # negative inputs and negative zeros (the public Gaus::GauBlur takes any float image)
    img = (rng.random((90, 150), dtype=np.float32) - 0.5).astype(np.float32)
    img[::7] = -0.0
    assert np.array_equal(bits(sift.gau_blur(img, 3.2)), bits(oracle.gau_blur(img, 3.2)))
    sift.set_tuning("blur_stream", -1)


def test_streaming_column_pass_on_a_very_wide_image(sift):
    """Rows of 2 MB: the segment the streaming column pass would choose (3198 rows) spans more than 2^31
    bytes, which its 32-bit row offsets cannot address, so it must cut shorter segments (975 rows here,
    two per column).  Checked bit for bit against the tiled column pass, which the other tests pin to
    the oracle."""
    w, h = 1 << 19, 1200
    rng = np.random.default_rng(41)
    img = np.tile(rng.random((h, 4096), dtype=np.float32), (1, w // 4096))
    img[:, ::4099] = 0.0
    sift.set_tuning("blur_stream", 0)
    want = sift.gau_blur(img, 6.4)
    sift.set_tuning("blur_stream", 1)
    got = sift.gau_blur(img, 6.4)
    sift.set_tuning("blur_stream", -1)
    assert np.array_equal(bits(got), bits(want))
    del got, want


def test_one_launch_blur_bit_exact_on_ragged_shapes(sift, oracle):
    """The one-launch form of the blur (row and column pass of a 64 x 64 tile through LDS; by
    itself only used on octaves of 3 Mpixel and more) forced onto small and ragged images: every
    tap count it is instantiated for, sizes that are not multiples of the tile, images smaller
    than the kernel, one-row and one-column images; exact mode against the oracle, float mode
    against the two-launch form."""
    rng = np.random.default_rng(37)
    shapes = [(200, 131), (64, 64), (1, 77), (130, 1), (65, 40), (333, 95), (129, 257)]
    for p, K in ((1.6, 11), (2.2627418, 15), (3.2, 21), (4.5254836, 29), (6.4, 39)):
        for i, (w, h) in enumerate(shapes):
            img = (rng.random((h, w), dtype=np.float32) - (0.5 if i == 3 else 0.0)).astype(np.float32)
            if i % 2:
                img[rng.random((h, w)) < 0.3] = 0.0
            sift.set_tuning("blur_fused", 99)
            got0 = bits(sift.gau_blur(img, p))
            got1 = bits(sift.gau_blur(img, p, blur_mode=1))
            sift.set_tuning("blur_fused", 0)
            assert np.array_equal(got0, bits(oracle.gau_blur(img, p))), (K, w, h)
            assert np.array_equal(got1, bits(sift.gau_blur(img, p, blur_mode=1))), (K, w, h)
    sift.set_tuning("blur_fused", -1)
    # the launch count shows the form in use: a 2048 x 1600 image takes it for its short kernels
    img = rng.random((1600, 2048), dtype=np.float32)
    sift.profile(True)
    sift.profile_read(reset=True)
    out = sift.gau_blur(img, 1.6)
    prof = sift.profile_read(reset=True)
    sift.profile(False)
    assert prof["blur_fused"][1] == 1 and prof["blur_row"][1] == 0 and prof["blur_col"][1] == 0
    sift.set_tuning("blur_fused", 0)
    try:
        assert np.array_equal(bits(out), bits(sift.gau_blur(img, 1.6)))
    finally:
        sift.set_tuning("blur_fused", -1)


def test_sweeping_one_launch_blur_equals_the_other_float_kernels(sift):
    """The sweeping one-launch form of the float blur (a workgroup walks down a strip of 256 columns, 8 input
    rows per step, the column pass from a register ring; by itself only used where the segments are long
    against the kernel) forced onto small and ragged images with segment lengths that cut the strips at
    awkward rows: every tap count it is instantiated for, widths that are not multiples of the strip, images
    smaller than the kernel, one-row and one-column images, negative inputs and exact zeros.  Its results must
    equal the two-launch float kernels bit for bit (same multiply-adds in the same order), which
    tests/test_gpu_fast_blur.py grades against the exact mode."""
    rng = np.random.default_rng(43)
    shapes = [(200, 131), (256, 64), (1, 77), (130, 1), (65, 40), (333, 95), (513, 130), (257, 9)]
    for p, K in ((3.2, 21), (4.5254836, 29), (6.4, 39)):
        assert len(sift_amd.gau_matrix(p)) == K
        for i, (w, h) in enumerate(shapes):
            img = (rng.random((h, w), dtype=np.float32) - (0.5 if i == 3 else 0.0)).astype(np.float32)
            if i % 2:
                img[rng.random((h, w)) < 0.3] = 0.0
            sift.set_tuning("blur_sweep", 0)
            sift.set_tuning("blur_fused", 0)
            want = bits(sift.gau_blur(img, p, blur_mode=1))
            for seg in (1, 8, 24, 56, 1000):
                sift.set_tuning("blur_sweep", seg)
                assert np.array_equal(bits(sift.gau_blur(img, p, blur_mode=1)), want), (K, w, h, seg)
    sift.set_tuning("blur_sweep", -1)
    sift.set_tuning("blur_fused", -1)
    # the launch count shows the form in use: an 8192 x 3072 image takes it for its 21-tap level, and the exact
    # mode never does
    img = rng.random((3072, 8192), dtype=np.float32)
    sift.profile(True)
    sift.profile_read(reset=True)
    out = sift.gau_blur(img, 3.2, blur_mode=1)
    prof = sift.profile_read(reset=True)
    assert prof["blur_fused"][1] == 1 and prof["blur_row"][1] == 0 and prof["blur_col"][1] == 0
    sift.set_tuning("blur_sweep", 0)
    try:
        assert np.array_equal(bits(out), bits(sift.gau_blur(img, 3.2, blur_mode=1)))
        sift.set_tuning("blur_sweep", 64)
        sift.profile_read(reset=True)
        sift.gau_blur(img, 3.2)
        prof = sift.profile_read(reset=True)
        assert prof["blur_fused"][1] == 0 and prof["blur_row"][1] == 1
    finally:
        sift.profile(False)
        sift.set_tuning("blur_sweep", -1)


def test_kernel_longer_than_the_image_bit_exact(sift, oracle):
    """Images shorter or narrower than the kernel go through the passes that stage the whole row /
    column strip once and clamp on the LDS read: every outputs-per-thread variant (chosen by pixel
    count), exact mode against the oracle, float mode within its tolerance."""
    rng = np.random.default_rng(41)
    for (h, w), p in (((20, 170000), 6.4), ((170000, 20), 6.4), ((30, 40000), 9.05), ((40000, 30), 9.05),
                      ((24, 3000), 12.8), ((3000, 24), 12.8), ((7, 5), 25.6),
                      # the last octaves of a 4K frame: one output per thread, clamped runs of taps
                      ((34, 60), 579.26), ((68, 120), 289.63), ((135, 240), 144.82), ((135, 240), 25.6), ((1, 200), 40.0),
                      ((200, 1), 40.0), ((100, 130), 3.2)):
        img = rng.random((h, w), dtype=np.float32)
        want = oracle.gau_blur(img, p)
        assert np.array_equal(bits(sift.gau_blur(img, p)), bits(want)), (h, w, p)
        assert np.max(np.abs(sift.gau_blur(img, p, blur_mode=1) - want)) < 2e-6, (h, w, p)


def test_two_contexts_extract_concurrently(sift):
    """Two contexts on one device driven from two threads: each keeps its two streams (the
    library joins them on the host, no stream waits on another's event) and both give the
    single-context result; then the same with the batch entry point (uploads on a third
    stream per context)."""
    import threading

    img = sift_amd.blob_image(640, 480, seed=8)
    want, _ = sift.extract(img)
    frames = np.stack([img, sift_amd.blob_image(640, 480, seed=9), img])
    want_b, want_n = sift.extract_batch(frames, cap=4096)
    other = sift_amd.SiftHip(0)
    got = {}

    def work(name, ctx):
        for _ in range(6):
            got[name], _ = ctx.extract(img)
        for _ in range(3):
            got[name + "_batch"] = ctx.extract_batch(frames, cap=4096)

    ths = [threading.Thread(target=work, args=("a", sift)), threading.Thread(target=work, args=("b", other))]
    [t.start() for t in ths]
    [t.join(timeout=100) for t in ths]
    assert not any(t.is_alive() for t in ths), "concurrent contexts hang"
    other.close()
    assert got["a"].tobytes() == want.tobytes() and got["b"].tobytes() == want.tobytes()
    for name in ("a_batch", "b_batch"):
        fb, nb = got[name]
        assert np.array_equal(nb, want_n)
        for f in range(len(frames)):
            assert fb[f, : nb[f]].tobytes() == want_b[f, : want_n[f]].tobytes()
    assert want_b[0, : want_n[0]].tobytes() == want.tobytes()


def test_page_locked_output_buffer(sift):
    """sift_hip_host_alloc: the same features arrive in a page-locked array as in a pageable one."""
    img = sift_amd.blob_image(320, 240, seed=11)
    ref, _ = sift.extract(img)
    out = sift.pinned_array(4096, sift_amd.FEAT_DTYPE)
    n = C.c_int(0)
    g = np.ascontiguousarray(img, np.float32)
    prm = sift_amd.SiftParams()._c()
    rc = sift.lib.sift_hip_extract(sift.h, C.c_void_p(g.ctypes.data), 320, 240, 1, C.byref(prm),
                                   C.c_void_p(out.ctypes.data), len(out), C.byref(n), None)
    assert rc == 0 and n.value == len(ref) and np.array_equal(out[: n.value], ref)
    p = C.c_void_p()
    assert sift.lib.sift_hip_host_alloc(sift.h, C.byref(p), 0) == sift_amd.capi.ERR_ARG
    assert sift.lib.sift_hip_host_alloc(sift.h, None, 16) == sift_amd.capi.ERR_ARG
    assert sift.lib.sift_hip_host_free(sift.h, None) == 0


def test_descriptors_kept_on_the_device_match_like_uploaded_ones(sift):
    """sift_hip_keep_descriptors / sift_hip_match_kept: the descriptor bytes of two extracts stay in device memory
    and are matched there; indices and both distances must equal sift_hip_match_u8 on the descr bytes of the two
    feature arrays that came back to the host, for both forms of the ratio test and for sets kept in either order;
    a kept set survives later extracts on the same context; with nothing to keep the call is an argument error."""
    fresh = sift_amd.SiftHip(0)
    h = C.c_void_p()
    assert fresh.lib.sift_hip_keep_descriptors(fresh.h, C.byref(h), None) == sift_amd.capi.ERR_ARG
    del fresh
    a = sift_amd.blob_image(640, 480, seed=5)
    b = np.roll(a, (7, -11), axis=(0, 1)).copy()
    prm = SiftParams(octaves=4)
    fa, _ = sift.extract(a, prm)
    ka, na = sift.keep_descriptors()
    fb, _ = sift.extract(b, prm)
    kb, nb = sift.keep_descriptors()
    try:
        assert na == len(fa) and nb == len(fb) and na > 100 and nb > 100
        sift.extract(sift_amd.blob_image(320, 240, seed=9), prm)  # the records of both extracts are gone now
        for form in (0, 1):
            for (kq, fq, kr, fr) in ((kb, fb, ka, fa), (ka, fa, kb, fb)):
                want = sift.match_u8(fq["descr"], fr["descr"], 0.5, form)
                got = sift.match_kept(kq, kr, 0.5, form)
                for w, g in zip(want, got):
                    assert np.array_equal(w, g)
                assert (got[0] >= 0).sum() > 50
        # after a batch call the kept set is the last frame's
        frames = np.stack([a, b, np.roll(a, (3, 5), axis=(0, 1))])
        outb, nbatch = sift.extract_batch(frames, prm)
        kl, nl = sift.keep_descriptors()
        try:
            last = outb[2][: nbatch[2]]
            assert nl == nbatch[2] > 100
            want = sift.match_u8(last["descr"], fa["descr"], 0.5, 0)
            got = sift.match_kept(kl, ka, 0.5, 0)
            for w, g in zip(want, got):
                assert np.array_equal(w, g)
        finally:
            sift.release_descriptors(kl)
    finally:
        sift.release_descriptors(ka)
        sift.release_descriptors(kb)
