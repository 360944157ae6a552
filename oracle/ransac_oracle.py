"""CPU restatement of the reference's RANSAC homography (Sift/ransac/xform.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing that ships) to check
sift_amd/host/xform.cpp and the device hypothesis scorer.

PARITY UNPINNED: the reference solves its least-squares systems with OpenCV
(cvSolve(..., CV_SVD), xform.cpp:208) and multiplies with cvMatMul
(xform.cpp:352); OpenCV is not available here and the reference holds no golden
vectors for this step, so nothing below was checked against the reference's
outputs.  The solve is numpy's SVD-based lstsq in float64 on the float32 matrix
entries the reference builds, i.e. the solution its float32 SVD approximates.

Follows the reference literally where that is cheap: float32 log-factorial sums
(xform.cpp:246-281), libc rand() % n sampling with redraw (xform.cpp:284-306),
strict `>` consensus update (xform.cpp:77), the double-promoted pow for the
stopping rule (xform.cpp:61,94), the x-then-y exchange sort before the last fit
(xform.cpp:114-137).
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

_libc = ctypes.CDLL(None)
_libc.rand.restype = ctypes.c_int
_libc.srand.argtypes = [ctypes.c_uint]

f32 = np.float32


def _log_factorial(n):
    """xform.cpp:253-259: float32 running sum of log(i), i = 1..n."""
    f = f32(0)
    for i in range(1, n + 1):
        f = f32(f + f32(math.log(f32(i))))
    return f


def _log_range(a, b):
    """xform.cpp:246-252: float32 running sum of log(i), i = a..b (log(0) = -inf)."""
    f = f32(0)
    for i in range(a, b + 1):
        li = f32(-np.inf) if i == 0 else f32(math.log(f32(i)))
        f = f32(f + li)
    return f


def calc_min_inliers(n, m, p_badsupp, p_badxform):
    """xform.cpp:261-281, in float32 as written there."""
    p_badsupp = f32(p_badsupp)
    log_n = _log_factorial(n)
    with np.errstate(invalid="ignore", over="ignore"):
        log_nm = f32(log_n - _log_range(n - m, n))
        temp = np.zeros(n + 1, dtype=f32)
        total = f32(0)
        lb = f32(math.log(p_badsupp))
        lg = f32(math.log(f32(1) - p_badsupp))
        for i in range(m + 1, n + 1):
            pi = f32(f32(f32(f32(f32(i - m) * lb) + f32(f32(n - i + m) * lg)) + log_nm) - _log_factorial(i - m))
            pi = f32(pi - f32(log_n - _log_range(n - i, n)))
            temp[i] = f32(math.exp(pi)) if np.isfinite(pi) else f32(0)
            total = f32(total + temp[i])
    s = total
    j = m + 1
    while j <= n:
        s = f32(s - temp[j])
        if s < f32(p_badxform):
            break
        j += 1
    return j


def lsq_homog(pts, mpts):
    """xform.cpp:177-219; pts, mpts float32 (n, 2).  None when n < 4 or rank deficient."""
    n = len(pts)
    if n < 4:
        return None
    A = np.zeros((2 * n, 8), dtype=f32)
    B = np.zeros(2 * n, dtype=f32)
    x, y, u, v = pts[:, 0], pts[:, 1], mpts[:, 0], mpts[:, 1]
    A[:n, 0] = x
    A[n:, 3] = x
    A[:n, 1] = y
    A[n:, 4] = y
    A[:n, 2] = 1
    A[n:, 5] = 1
    A[:n, 6] = -x * u
    A[:n, 7] = -y * u
    A[n:, 6] = -x * v
    A[n:, 7] = -y * v
    B[:n] = u
    B[n:] = v
    sol, _, rank, _ = np.linalg.lstsq(A.astype(np.float64), B.astype(np.float64), rcond=1e-13)
    if rank < 8:
        return None
    H = np.ones(9, dtype=f32)
    H[:8] = sol.astype(f32)
    return H


def xfer_err(H, pts, mpts):
    """xform.cpp:334-363: float products accumulated wide, rounded to float32, float32 division and distance."""
    H64 = H.astype(np.float64)
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    with np.errstate(all="ignore"):
        u = (H64[0] * x + H64[1] * y + H64[2]).astype(f32)
        v = (H64[3] * x + H64[4] * y + H64[5]).astype(f32)
        w = (H64[6] * x + H64[7] * y + H64[8]).astype(f32)
        dx = (u / w - mpts[:, 0]).astype(f32)
        dy = (v / w - mpts[:, 1]).astype(f32)
        return np.sqrt((dx * dx + dy * dy).astype(f32)).astype(f32)


def find_consensus(H, pts, mpts, err_tol):
    """xform.cpp:365-395: indices with error <= err_tol (NaN is out)."""
    return np.nonzero(xfer_err(H, pts, mpts) <= f32(err_tol))[0]


def ransac_xform(pts, mpts, m, p_badxform, err_tol, seed, trace=None):
    """xform.cpp:37-168 over the matched pairs pts -> mpts (float32 (nm, 2)).

    Returns (H or None, inlier indices, in_min).  `trace`, when a list, receives
    (sample indices, consensus size or -1) per iteration.
    """
    pts = np.ascontiguousarray(pts, dtype=f32)
    mpts = np.ascontiguousarray(mpts, dtype=f32)
    nm = len(pts)
    _libc.srand(seed)
    if nm < m or m < 4:
        return None, np.zeros(0, dtype=np.int64), 0
    in_min = calc_min_inliers(nm, m, 0.10, p_badxform)
    in_frac = f32(0.25)
    k = 0

    def p_bad(frac, k):
        return f32((1.0 - float(frac) ** m) ** k)

    p = p_bad(in_frac, k)
    best = np.zeros(0, dtype=np.int64)
    while p > f32(p_badxform):
        sampled = np.zeros(nm, dtype=bool)
        sample = []
        for _ in range(m):
            while True:
                x = _libc.rand() % nm
                if not sampled[x]:
                    break
            sampled[x] = True
            sample.append(x)
        H = lsq_homog(pts[sample], mpts[sample])
        cnt = -1
        if H is not None:
            cons = find_consensus(H, pts, mpts, err_tol)
            cnt = len(cons)
            if cnt > len(best):
                best = cons
                in_frac = f32(f32(cnt) / f32(nm))
        if trace is not None:
            trace.append((tuple(sample), cnt))
        k += 1
        p = p_bad(in_frac, k)
    if len(best) < in_min:
        return None, np.zeros(0, dtype=np.int64), in_min
    H = lsq_homog(pts[best], mpts[best])
    if H is None:
        return None, np.zeros(0, dtype=np.int64), in_min
    fin = find_consensus(H, pts, mpts, err_tol)
    order = list(fin)
    for key in (0, 1):  # exchange sort by x, then by y (xform.cpp:114-137)
        for a in range(len(order) - 1):
            for b in range(a + 1, len(order)):
                if pts[order[a], key] > pts[order[b], key]:
                    order[a], order[b] = order[b], order[a]
    H = lsq_homog(pts[order], mpts[order])
    return H, fin, in_min
