"""blur_mode 1 (fp32 FMA accumulation instead of the reference's float product + double
accumulation) against the oracle, in the terms the north star states for non-exact stages:
keypoints with a partner at the same scale and layer within 0.5 source pixels, and descriptor
distance of the partners.  The pyramid itself must stay within a few float ulps."""
import numpy as np
import pytest

import sift_amd
from sift_amd import SiftParams

pytestmark = pytest.mark.gpu

# The bars are the ones stated for non-exact stages: every keypoint has a partner within 0.5 source
# pixels at the same scale, layer and orientation (the few that do not are near-ties of the non-strict
# extremum test, counted and bounded below), and the UNQUANTISED descriptors (the clamped histograms
# that the bytes are cut from) of keypoints at identical positions are within 1e-3 in L2.
MAX_UNPARTNERED = 0.005      # fraction of either set without a partner (measured: 0 .. 0.4 %)
MAX_HIST_L2 = 1e-3           # clamped histograms of exact-position pairs, before quantisation
MAX_LEVEL_ABS_ERR = 2e-6     # Gaussian levels of a [0, 1] image (measured 4.8e-7 .. 6.6e-7)


def key(f):
    return np.stack([np.round(np.log2(2.0 / f["size"])).astype(np.int64), f["kai"].astype(np.int64),
                     f["y"].astype(np.int64), f["x"].astype(np.int64), f["sita"].astype(np.int64)], axis=1)


def partner_stats(a, b, ha=None, hb=None):
    """Fraction of a's keypoints with a partner in b: same octave, layer and orientation bin, position
    within 0.5 source pixels (octave pixels / size); the L2 distances of the unquantised descriptors
    (clamped histograms ha, hb) of the exact-position pairs; and how many records are identical
    (position, layer, orientation)."""
    found = 0
    identical = 0
    l2 = []
    index = {}
    for i, k in enumerate(map(tuple, key(b))):
        index.setdefault((k[0], k[1], k[4]), []).append(i)
    kb = key(b)
    for i, k in enumerate(map(tuple, key(a))):
        cand = index.get((k[0], k[1], k[4]), [])
        if not cand:
            continue
        c = np.asarray(cand)
        tol = 0.5 * float(a["size"][i])  # 0.5 source pixels in octave pixels
        d = np.maximum(np.abs(kb[c, 2] - k[2]), np.abs(kb[c, 3] - k[3]))
        if (d <= tol).any():
            found += 1
        exact = c[d == 0]
        if len(exact):
            identical += 1
            if ha is not None:
                da = ha[i].astype(np.float64)
                db = hb[exact[0]].astype(np.float64)
                l2.append(float(np.sqrt(((da - db) ** 2).sum())))
    return found / max(len(a), 1), np.asarray(l2), identical


@pytest.mark.parametrize("case", ["640x480", "1080p", "4k_crop"])
def test_fast_blur_keypoints_agree_with_oracle(sift, oracle, case):
    if case == "640x480":
        img, octaves = sift_amd.blob_image(640, 480, seed=1), 0
    elif case == "1080p":
        img, octaves = sift_amd.blob_image(1920, 1080, seed=12), 4
    else:
        img, octaves = np.ascontiguousarray(sift_amd.blob_image(3840, 2160, seed=1)[700:1600, 1400:2500]), 2
    f, h = sift.extract(img, SiftParams(octaves=octaves, blur_mode=1), want_hist=True)
    fo, ho = oracle.extract(img, octaves=octaves)
    fwd, l2, same = partner_stats(fo, f, ho, h)
    back, _, _ = partner_stats(f, fo)
    print("%s: %d oracle / %d fast keypoints, %d identical records, without a partner: %d of the oracle's, %d of the "
          "fast mode's; L2 of the unquantised descriptors of identical records: median %.3g max %.3g"
          % (case, len(fo), len(f), same, round((1 - fwd) * len(fo)), round((1 - back) * len(f)), np.median(l2), l2.max()))
    assert len(fo) > 500
    assert 1 - fwd <= MAX_UNPARTNERED and 1 - back <= MAX_UNPARTNERED
    assert same >= (1 - MAX_UNPARTNERED) * len(fo)
    assert l2.max() <= MAX_HIST_L2


def test_fast_blur_levels_within_a_few_ulps(sift, oracle):
    img = sift_amd.blob_image(800, 600, seed=4)
    for octave in (0, 1, 2):
        g, _ = sift.octave_levels(img, octave, SiftParams(blur_mode=1))
        go, _ = oracle.octave_levels(img, octave)
        err = np.abs(g.astype(np.float64) - go).max()
        print("octave %d: max |fast - exact| over the six levels = %.3g" % (octave, err))
        assert err <= MAX_LEVEL_ABS_ERR
