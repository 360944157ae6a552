// Planar homography from point pairs by linear least squares, and the
// transfer error of a pair under a homography.  Shared by the host RANSAC
// (sift_amd/host/xform.cpp) and the device hypothesis kernels, so both sides
// run the same arithmetic: every operation below is an IEEE double or float
// add/mul/div/sqrt, with fused-multiply-add contraction switched off.
//
// Stands in for the reference's lsq_homog (Sift/ransac/xform.cpp:177-219,
// cvSolve(A, B, X, CV_SVD) on the 2n x 8 system with h8 = 1) and
// persp_xform_pt / homog_xfer_err (xform.cpp:340-363, cvMatMul of a 3x3 by a
// 3x1 float matrix, then the Euclidean distance).  OpenCV is not a dependency
// here: the solve is a Householder QR with column pivoting in double.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SIFT_HD __host__ __device__
#else
#define SIFT_HD
#endif
// clang contracts a*b+c into one fused operation by default in device code; gcc
// in ISO mode does not.  Keep both sides unfused.
#if defined(__clang__)
#define SIFT_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define SIFT_NO_CONTRACT
#endif

namespace sift_homog {

constexpr int NCOL = 8;         // unknowns h0..h7
constexpr int NAUG = NCOL + 1;  // [A | b] row length

// The two rows one pair (x,y) -> (u,v) contributes (xform.cpp:193-206).  The
// products are rounded to float first, as the reference stores them in a
// CV_32FC1 matrix.
SIFT_HD inline void dlt_rows(float x, float y, float u, float v, double* r0, double* r1) {
    SIFT_NO_CONTRACT
    const float xu = -x * u, yu = -y * u, xv = -x * v, yv = -y * v;
    r0[0] = x; r0[1] = y; r0[2] = 1.0; r0[3] = 0.0; r0[4] = 0.0; r0[5] = 0.0; r0[6] = xu; r0[7] = yu; r0[8] = u;
    r1[0] = 0.0; r1[1] = 0.0; r1[2] = 0.0; r1[3] = x; r1[4] = y; r1[5] = 1.0; r1[6] = xv; r1[7] = yv; r1[8] = v;
}

// Least-squares solution of A h = b for the row-major rows x 9 matrix [A | b]
// (overwritten).  Returns 1 and h[0..8) on success, 0 when A is rank deficient
// (|R_kk| <= 1e-12 |R_00|: repeated or collinear sample points).
SIFT_HD inline int solve_lsq(double* M, int rows, double* h) {
    SIFT_NO_CONTRACT
    int perm[NCOL];
    for (int c = 0; c < NCOL; c++) perm[c] = c;
    double r00 = 0.0;
    for (int k = 0; k < NCOL; k++) {
        // pivot: remaining column of largest norm (recomputed, rows are few or the call is rare)
        int best = k;
        double bn = -1.0;
        for (int c = k; c < NCOL; c++) {
            double s = 0.0;
            for (int r = k; r < rows; r++) s += M[r * NAUG + c] * M[r * NAUG + c];
            if (s > bn) { bn = s; best = c; }
        }
        if (best != k) {
            for (int r = 0; r < rows; r++) {
                const double t = M[r * NAUG + k];
                M[r * NAUG + k] = M[r * NAUG + best];
                M[r * NAUG + best] = t;
            }
            const int tp = perm[k]; perm[k] = perm[best]; perm[best] = tp;
        }
        const double norm = sqrt(bn);
        if (k == 0) r00 = norm;
        if (!(norm > 1e-12 * r00) || norm == 0.0) return 0;
        // Householder vector v = x + sign(x0) |x| e0, stored in place below the diagonal
        const double x0 = M[k * NAUG + k];
        const double alpha = x0 >= 0.0 ? -norm : norm;
        const double v0 = x0 - alpha;
        const double vtv = v0 * v0 + (bn - x0 * x0);
        if (vtv > 0.0) {
            for (int c = k + 1; c < NAUG; c++) {
                double dot = v0 * M[k * NAUG + c];
                for (int r = k + 1; r < rows; r++) dot += M[r * NAUG + k] * M[r * NAUG + c];
                const double f = 2.0 * dot / vtv;
                M[k * NAUG + c] -= f * v0;
                for (int r = k + 1; r < rows; r++) M[r * NAUG + c] -= f * M[r * NAUG + k];
            }
        }
        M[k * NAUG + k] = alpha;
    }
    double z[NCOL];
    for (int k = NCOL - 1; k >= 0; k--) {
        double s = M[k * NAUG + NCOL];
        for (int c = k + 1; c < NCOL; c++) s -= M[k * NAUG + c] * z[c];
        z[k] = s / M[k * NAUG + k];
    }
    for (int k = 0; k < NCOL; k++) h[perm[k]] = z[k];
    return 1;
}

// (x, y) through the homography T (row-major 3x3): float products summed in
// double and rounded to float once, then the perspective division in float.
SIFT_HD inline void persp_xform_pt(const float* T, float x, float y, float* ox, float* oy) {
    SIFT_NO_CONTRACT
    const float u = (float)((double)T[0] * x + (double)T[1] * y + (double)T[2]);
    const float v = (float)((double)T[3] * x + (double)T[4] * y + (double)T[5]);
    const float w = (float)((double)T[6] * x + (double)T[7] * y + (double)T[8]);
    *ox = u / w;
    *oy = v / w;
}

// Distance between T(x, y) and (u, v) in float (xform.cpp:334-339, 359-363).
SIFT_HD inline float homog_xfer_err(const float* T, float x, float y, float u, float v) {
    SIFT_NO_CONTRACT
    float px, py;
    persp_xform_pt(T, x, y, &px, &py);
    const float dx = px - u, dy = py - v;
    return sqrtf(dx * dx + dy * dy);
}

}  // namespace sift_homog
