"""GPU checks at the benchmark's full sizes (4K frame, 1080p batch, 8192^2 image,
1M x 1M match), where the CPU oracle cannot redo the whole job in seconds.

What is compared with the oracle: whole pyramid octaves that are small enough
(bit-exact), the features of octave 1 inside a crop of the 4K frame (octave 1
resamples 1:1, so a crop far enough from its border sees the same pixels), and a
slice of the 1M queries against all 1M references.  The rest are properties that
hold at any size: deterministic output, the reference's append order, bounds,
agreement between the large-image code paths and the small-image ones on the same
content, and planted matches being found.
"""
import numpy as np
import pytest

import sift_amd
from sift_amd import SiftParams
from sift_amd.synth import _uniform

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_reference_order(f):
    """(octave, layer, row, column, orientation) strictly increasing: the order in
    which Sift::pyramid appends (Sift.cpp:173-200, :285-293)."""
    if len(f) < 2:
        return
    key = np.stack([-f["size"].astype(np.float64), f["kai"], f["y"], f["x"], f["sita"]], axis=1)
    a, b = key[:-1], key[1:]
    neq = a != b
    first = neq.argmax(axis=1)
    assert neq.any(axis=1).all(), "duplicate feature record"
    rows = np.arange(len(a))
    assert (a[rows, first] < b[rows, first]).all()


def assert_sane(f, w, h, intvls=3):
    assert (f["descr"] <= 102).all()  # min(255, (int)(512 * min(v, 0.2)))
    assert ((f["kai"] >= 1) & (f["kai"] <= intvls)).all()
    assert (f["sita"] % 10 == 0).all() and (f["sita"] >= 0).all() and (f["sita"] < 360).all()
    ow = np.floor(w * f["size"]).astype(np.int64)
    oh = np.floor(h * f["size"]).astype(np.int64)
    assert ((f["x"] > 0) & (f["x"] < ow - 1) & (f["y"] > 0) & (f["y"] < oh - 1)).all()


def compare_records(f, h, fo, ho):
    assert len(f) == len(fo)
    for k in ("size", "kai", "sigmaOCT", "x", "y", "sita"):
        assert np.array_equal(f[k], fo[k]), k
    assert np.array_equal(bits(f["mm"]), bits(fo["mm"]))
    if h is not None:
        l2 = np.sqrt(((h.astype(np.float64) - ho) ** 2).sum(axis=1))
        assert l2.max() < 1e-3, l2.max()
    diff = np.abs(f["descr"].astype(np.int16) - fo["descr"].astype(np.int16))
    assert diff.max() <= 1


@pytest.fixture(scope="module")
def frame4k():
    return sift_amd.blob_image(3840, 2160, seed=1)


# ---- 4K frame, 4 octaves ---------------------------------------------------------------
def test_4k_extract_is_deterministic_ordered_and_in_bounds(sift, frame4k):
    prm = SiftParams(octaves=4)
    f1, h1 = sift.extract(frame4k, prm, cap=1 << 16, want_hist=True)
    f2, h2 = sift.extract(frame4k, prm, cap=1 << 16, want_hist=True)
    assert len(f1) > 10000
    assert f1.tobytes() == f2.tobytes()
    assert np.array_equal(h1, h2)
    assert_reference_order(f1)
    assert_sane(f1, 3840, 2160)
    assert set(np.unique(f1["size"])) <= {2.0, 1.0, 0.5, 0.25}


@pytest.mark.parametrize("octave", [0, 1, 2, 3])
def test_4k_pyramid_octaves_bit_exact(sift, oracle, frame4k, octave):
    # octave 0 = 7680 x 4320 (75 % of the frame's pixels; the only place the
    # 8-outputs-per-thread kernels run over 7680-pixel rows): about half a minute of oracle time
    g, d = sift.octave_levels(frame4k, octave)
    go, do = oracle.octave_levels(frame4k, octave)
    assert np.array_equal(bits(g), bits(go))
    del go
    assert np.array_equal(bits(d), bits(do))


def test_1080p_octave0_features_equal_oracle(sift, oracle):
    # One octave only: the doubled 1080p frame (3840 x 2160) is large enough for the kernels
    # that octave 0 of the 4K frame uses (8 outputs per thread, 16-row detect tiles), and the
    # oracle redoes the whole of it -- blur, detect, orientation, descriptors -- record for record.
    img = sift_amd.blob_image(1920, 1080, seed=21)
    f, h = sift.extract(img, SiftParams(octaves=1), cap=1 << 16, want_hist=True)
    fo, ho = oracle.extract(img, octaves=1)
    assert len(fo) > 3000 and (fo["size"] == 2.0).all()
    compare_records(f, h, fo, ho)


def test_7680_wide_octave0_features_equal_oracle(sift, oracle):
    # A 3840-pixel-wide strip, one octave: octave 0 is 7680 x 640, the row length of the 4K frame's
    # octave 0, so detection, orientation and descriptors run over rows of that width (tile
    # remapping, buffer offsets, 7680-pixel strides) and the oracle redoes every record.
    img = np.ascontiguousarray(sift_amd.blob_image(3840, 2160, seed=1)[900:1220])
    f, h = sift.extract(img, SiftParams(octaves=1), cap=1 << 16, want_hist=True)
    fo, ho = oracle.extract(img, octaves=1)
    assert len(fo) > 1500 and (fo["size"] == 2.0).all() and fo["x"].max() > 7000
    compare_records(f, h, fo, ho)


def test_720p_reference_octave_count_features_equal_oracle(sift, oracle):
    # octaves = 0 -> the reference's own count (Sift.cpp:131-139): the deep octaves are a few thousand
    # pixels under kernels longer than the image (whole-row / whole-column passes, their last level
    # built as an image because it has more than 1024 taps) and run on a stream of their own;
    # every record of every octave against the oracle.
    from scipy import ndimage

    # fine blobs plus the same kind of field magnified 8x and 40x, so that deep octaves find something
    img = 0.5 * sift_amd.blob_image(1280, 720, seed=23)
    img += 0.25 * ndimage.zoom(sift_amd.blob_image(160, 90, seed=24), 8, order=1)
    img += 0.25 * ndimage.zoom(sift_amd.blob_image(32, 18, seed=25), 40, order=1)
    img = np.ascontiguousarray(img.astype(np.float32))
    noct = sift_amd.pyramid_octaves(1280, 720)
    assert noct >= 6
    f, h = sift.extract(img, SiftParams(octaves=0), cap=1 << 16, want_hist=True)
    fo, ho = oracle.extract(img, octaves=noct)
    print("octaves %d, features per octave size:" % noct, {float(v): int((fo["size"] == v).sum()) for v in np.unique(fo["size"])})
    assert len(fo) > 1500 and len(np.unique(fo["size"])) >= 4
    compare_records(f, h, fo, ho)


def test_4k_octave1_features_equal_oracle_on_a_crop(sift, oracle, frame4k):
    # Octave 1 resamples 1:1 (exactly, for widths below 4097), every later stage is
    # local: blur radii of its six levels sum to 163 px, the descriptor window adds
    # up to 96.  Features further than that from the crop's border must agree.
    f, h = sift.extract(frame4k, SiftParams(octaves=2), cap=1 << 16, want_hist=True)
    ox, oy, cw, ch, margin = 1400, 700, 1100, 900, 270
    crop = np.ascontiguousarray(frame4k[oy:oy + ch, ox:ox + cw])
    fo, ho = oracle.extract(crop, octaves=2)
    sel = (f["size"] == 1.0) & (f["x"] >= ox + margin) & (f["x"] < ox + cw - margin) & \
          (f["y"] >= oy + margin) & (f["y"] < oy + ch - margin)
    selo = (fo["size"] == 1.0) & (fo["x"] >= margin) & (fo["x"] < cw - margin) & \
           (fo["y"] >= margin) & (fo["y"] < ch - margin)
    fo_s = fo[selo].copy()
    fo_s["x"] += ox
    fo_s["y"] += oy
    assert selo.sum() >= 20
    compare_records(f[sel], h[sel], fo_s, ho[selo])


# ---- batch of 1080p frames -------------------------------------------------------------
def test_1080p_batch_equals_single_frames_and_oracle(sift, oracle):
    frames = np.stack([sift_amd.blob_image(1920, 1080, seed=s) for s in (11, 12, 13, 14)])
    out, n = sift.extract_batch(frames, SiftParams(octaves=4), cap=1 << 15)
    for i in range(len(frames)):
        f, _ = sift.extract(frames[i], SiftParams(octaves=4))
        assert n[i] == len(f) and n[i] > 2000
        assert out[i, : n[i]].tobytes() == f.tobytes()
        assert_reference_order(f)
    # one frame of the batch against the oracle, record for record
    f, h = sift.extract(frames[2], SiftParams(octaves=4), want_hist=True)
    fo, ho = oracle.extract(frames[2], octaves=4)
    compare_records(f, h, fo, ho)


# ---- 8192 x 8192, 6 octaves ------------------------------------------------------------
@pytest.fixture(scope="module")
def image8k():
    """Fine blobs (tiles of the usual synthetic field) plus the same kind of field
    magnified 8x, 32x and 128x: every octave blurs with an absolute sigma that
    doubles while the image halves, so deep octaves only respond to large blobs."""
    from scipy import ndimage

    tiles = [sift_amd.blob_image(2048, 2048, seed=20 + k) for k in range(4)]
    rows = [np.concatenate([tiles[(r + c) % 4] for c in range(4)], axis=1) for r in range(4)]
    img = 0.35 * np.concatenate(rows, axis=0)
    for weight, mag, seed in ((0.25, 8, 30), (0.2, 32, 31), (0.2, 128, 32)):
        img += weight * ndimage.zoom(sift_amd.blob_image(8192 // mag, 8192 // mag, seed=seed), mag, order=1)
    return np.ascontiguousarray(img.astype(np.float32))


def test_8k_six_octaves_properties(sift, image8k):
    prm = SiftParams(octaves=6)
    f, _ = sift.extract(image8k, prm, cap=1 << 18)
    f2, _ = sift.extract(image8k, prm, cap=1 << 18)
    assert len(f) > 5000
    assert f.tobytes() == f2.tobytes()
    assert_reference_order(f)
    assert_sane(f, 8192, 8192)
    per_octave = {float(s): int((f["size"] == s).sum()) for s in np.unique(f["size"])}
    print("features per octave scale:", per_octave)
    assert per_octave.get(0.5, 0) > 50 and sum(v for k, v in per_octave.items() if k <= 0.25) > 0


def test_8k_octave2_equals_small_image_path_on_a_crop(sift, image8k):
    # Octaves below 1:1 sample the source at (i + 0.5) / ratio - 0.5, an absolute
    # position, so a crop whose origin is a multiple of the decimation sees the
    # same samples.  Octave 2 of the 8192^2 image (2-outputs-per-thread kernels
    # are not used there, long column passes are) against the same octave of a
    # 4096^2 crop.  Margin in octave pixels: detection reads all six levels (blur
    # radii sum to 324) +-1, the descriptor reads level <= 3 (138) over a window of
    # up to 191; twice that in source pixels.
    f, _ = sift.extract(image8k, SiftParams(octaves=3), cap=1 << 18)
    ox, oy, cs, margin = 2048, 3072, 4096, 700
    crop = np.ascontiguousarray(image8k[oy:oy + cs, ox:ox + cs])
    fc, _ = sift.extract(crop, SiftParams(octaves=3), cap=1 << 18)
    s = 0.5
    lo, hi = margin * s, (cs - margin) * s
    sel = (f["size"] == s) & (f["x"] >= ox * s + lo) & (f["x"] < ox * s + hi) & \
          (f["y"] >= oy * s + lo) & (f["y"] < oy * s + hi)
    selc = (fc["size"] == s) & (fc["x"] >= lo) & (fc["x"] < hi) & (fc["y"] >= lo) & (fc["y"] < hi)
    want = fc[selc].copy()
    want["x"] += int(ox * s)
    want["y"] += int(oy * s)
    print("octave-2 features compared:", int(selc.sum()))
    assert selc.sum() >= 3
    assert f[sel].tobytes() == want.tobytes()


def test_8k_deep_octave_levels_bit_exact(sift, oracle, image8k):
    # octave 5 = 512 x 512 with kernels of 307 ... 1739 taps: the column pass that
    # slides over global memory, at full length
    g, d = sift.octave_levels(image8k, 5, SiftParams(octaves=6))
    go, do = oracle.octave_levels(image8k, 5)
    assert np.array_equal(bits(g), bits(go))
    assert np.array_equal(bits(d), bits(do))


# ---- 1M x 1M match ---------------------------------------------------------------------
def test_1m_match_slice_equals_oracle_and_planted_pairs_are_found(sift, oracle):
    n = 1 << 20
    q, r = sift_amd.planted_descriptors(n, n)
    best, d1, d2 = sift.match_u8(q, r, 0.5, 0)
    # queries that planted_descriptors overwrote with a noisy copy of a reference
    npl = int(n * 0.01)
    qi = (_uniform(2, 7, npl) * n).astype(np.int64)
    ri = (_uniform(2, 8, npl) * n).astype(np.int64)
    last = {}
    for a, b in zip(qi, ri):
        last[int(a)] = int(b)
    qs = np.fromiter(last.keys(), np.int64)
    rs = np.fromiter(last.values(), np.int64)
    assert np.array_equal(best[qs], rs.astype(np.int32))
    assert (d1[qs] <= 128 * 4).all()
    # nothing else passes a 0.5 ratio among i.i.d. descriptors
    assert (best >= 0).sum() == len(qs)
    # exact nearest / second nearest of a slice of queries against all references
    rows = np.concatenate([np.arange(0, 64), qs[:32], np.arange(n - 64, n)])
    bo, d1o, d2o = oracle.match_u8(q[rows], r, 0.5, 0)
    assert np.array_equal(best[rows], bo)
    assert np.array_equal(d1[rows], d1o)
    assert np.array_equal(d2[rows], d2o)
