"""multiBandBlend on the GPU (sift_hip_multiband_blend) against the CPU restatement of
Sift/stitch/Blend.cpp:638-740, byte for byte.  Parity with the reference itself is unpinned
for this step (Blend.cpp cannot be built here; the reference holds no vectors for it)."""
import numpy as np
import pytest

import sift_amd
from sift_amd import capi

pytestmark = pytest.mark.gpu


def scene(h, w, seed, border=0):
    """Smooth colourful image (the blender works on photographs, not noise) with an optional black frame
    like the one the cylindrical projection leaves."""
    from scipy import ndimage

    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 3)).astype(np.float32)
    img = ndimage.gaussian_filter(img, (4, 4, 0))
    img = (img - img.min()) / (img.max() - img.min())
    img = (img * 235 + 15).astype(np.uint8)
    if border:
        yy, xx = np.mgrid[0:h, 0:w]
        curve = (border * (1 - ((xx - w / 2) / (w / 2)) ** 2)).astype(np.int32)  # arched top and bottom edge
        img[(yy < border - curve) | (yy >= h - border + curve)] = 0
    return np.ascontiguousarray(img)


CASES = [
    # lh, lw, rh, rw, dx, dy, border
    (240, 320, 240, 320, 101, 239, 12),
    (240, 320, 240, 320, 100, 240, 0),      # even offsets too (the reference's own caller makes dx odd)
    (200, 260, 200, 260, 57, -191, 8),      # right image above: dy < 0
    (300, 200, 280, 220, 33, 270, 10),      # images of different sizes, overlap taller than wide
    (120, 400, 120, 400, 333, 117, 6),      # overlap much wider than tall: the resampler's height comes from the width
    (90, 120, 90, 120, 9, 85, 0),           # narrower than 16 columns: the coarse levels are empty
    (64, 64, 64, 64, 1, 1, 0),
    (480, 640, 480, 640, 215, 471, 20),
]


@pytest.mark.parametrize("case", CASES)
def test_multiband_blend_equals_restatement(sift, oracle, case):
    lh, lw, rh, rw, dx, dy, border = case
    l = scene(lh, lw, seed=lh + dx, border=border)
    r = scene(rh, rw, seed=rw + dx + 1, border=border)
    if border:
        l[:, lw - 3:] = 0   # a few all-black columns and rows inside the overlap: the carried indices
        r[rh // 2] = 0
    lo, ro = oracle.multiband_blend(l, r, dx, dy)
    lg, rg = l.copy(), r.copy()
    sift.multiband_blend(lg, rg, dx, dy)
    changed = int((lo != l).sum() + (ro != r).sum())
    print("overlap %d x %d: %d bytes changed by the blend" % (dx, abs(dy), changed))
    if dx > 40:
        assert changed > 1000
    assert np.array_equal(lg, lo)
    assert np.array_equal(rg, ro)


def test_multiband_blend_rejects_bad_arguments(sift):
    a = np.zeros((32, 32, 3), np.uint8)
    b = np.zeros((32, 32, 3), np.uint8)
    call = lambda dx, dy: sift.lib.sift_hip_multiband_blend(sift.h, capi._ptr(a), 32, 32, capi._ptr(b), 32, 32, dx, dy)  # noqa: E731
    assert call(0, 5) == capi.ERR_ARG and call(5, 0) == capi.ERR_ARG and call(-3, 4) == capi.ERR_ARG
    assert call(20000, 4) == capi.ERR_ARG
    assert sift.lib.sift_hip_multiband_blend(sift.h, capi._ptr(a), 32, 32, capi._ptr(a), 32, 32, 3, 3) == capi.ERR_ARG
    assert call(40, 50) == capi.OK  # an overlap larger than the images: everything outside them counts as black
    assert not a.any() and not b.any()
