"""Deterministic synthetic inputs (the reference ships no images).

Everything derives from a 64-bit integer hash, not from a library RNG, so the
same seed gives the same bytes on every machine.
"""
from __future__ import annotations

import numpy as np

_M = (1 << 64) - 1


def _splitmix(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M)
    return z ^ (z >> np.uint64(31))


def _uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n doubles in [0,1) from (seed, stream, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        base = np.uint64((seed * 0x100000001B3 + stream * 0x9E3779B1) & _M)
        h = _splitmix(_splitmix(idx ^ base) + np.uint64(stream))
    return (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def blob_image(w: int, h: int, seed: int = 1, shift_x: int = 0, as_u8: bool = False) -> np.ndarray:
    """Sum of w*h/400 Gaussian blobs (sigma in [1.5,7.5], amplitude in [-.5,.5]),
    min-max normalised and quantised to 8 bits, returned as float32 k/255 (what
    ImgRaw makes of an 8-bit bitmap, Imgraw.cpp:144-150) or as the bytes.
    shift_x moves the blob field (a second view of the same scene)."""
    nb = max(1, (w * h) // 400)
    cx = _uniform(seed, 1, nb) * (w + 40) - 20 + shift_x
    cy = _uniform(seed, 2, nb) * (h + 40) - 20
    sg = 1.5 + 6.0 * _uniform(seed, 3, nb)
    am = _uniform(seed, 4, nb) - 0.5
    img = np.zeros((h, w), np.float64)
    for k in range(nb):
        r = int(np.ceil(4.0 * sg[k]))
        x0, x1 = max(0, int(cx[k]) - r), min(w, int(cx[k]) + r + 1)
        y0, y1 = max(0, int(cy[k]) - r), min(h, int(cy[k]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        gx = np.exp(-((np.arange(x0, x1) - cx[k]) ** 2) / (2 * sg[k] ** 2))
        gy = np.exp(-((np.arange(y0, y1) - cy[k]) ** 2) / (2 * sg[k] ** 2))
        img[y0:y1, x0:x1] += am[k] * gy[:, None] * gx[None, :]
    lo, hi = img.min(), img.max()
    u8 = np.clip(np.rint((img - lo) / max(hi - lo, 1e-30) * 255.0), 0, 255).astype(np.uint8)
    if as_u8:
        return u8
    return (u8.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)


def textured_image(w: int, h: int, seed: int = 3) -> np.ndarray:
    """Dense, anisotropic content: band-passed noise at four scales (sigma 1.5 ... 12 pixels) on top
    of the blob field, float32 in [0, 1].  Where blob_image leaves most of a frame smooth, nearly
    every pixel of this one passes the detector's contrast bar, and its structure has direction
    (isotropic blobs look the same in any descriptor frame)."""
    from scipy import ndimage

    tex = np.zeros((h, w), np.float64)
    for k, (sigma, weight) in enumerate(((1.5, 1.0), (3.0, 1.0), (6.0, 1.2), (12.0, 1.5))):
        u1 = np.maximum(_uniform(seed, 20 + 2 * k, w * h), 1e-300)
        u2 = _uniform(seed, 21 + 2 * k, w * h)
        noise = (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).reshape(h, w)
        n = ndimage.gaussian_filter(noise, sigma)
        tex += weight * n / n.std()
    tex = (tex - tex.min()) / (tex.max() - tex.min())
    return (0.6 * tex + 0.4 * blob_image(w, h, seed=seed)).astype(np.float32)


def planted_descriptors(nq: int, nr: int, seed_q: int = 2, seed_r: int = 3, vmax: int = 102, plant: float = 0.01):
    """Query / reference byte descriptors: i.i.d. values in [0, vmax], with a
    fraction `plant` of the queries replaced by a noisy copy of some reference so
    that the ratio test accepts them."""
    def block(seed, n):
        with np.errstate(over="ignore"):
            idx = np.arange(n * 16, dtype=np.uint64)
            hsh = _splitmix(_splitmix(idx ^ np.uint64((seed * 0x100000001B3) & _M)) + np.uint64(seed))
        b = hsh.view(np.uint8).reshape(n, 128)
        return (b.astype(np.uint16) * (vmax + 1) >> 8).astype(np.uint8)

    q, r = block(seed_q, nq), block(seed_r, nr)
    npl = int(nq * plant)
    if npl and nr:
        qi = (_uniform(seed_q, 7, npl) * nq).astype(np.int64)
        ri = (_uniform(seed_q, 8, npl) * nr).astype(np.int64)
        noise = (_uniform(seed_q, 9, npl * 128).reshape(npl, 128) * 5).astype(np.int16) - 2
        q[qi] = np.clip(r[ri].astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return q, r
