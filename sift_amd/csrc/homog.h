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

#include <algorithm>
#include <vector>

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
    // Every sweep walks the matrix row by row (it is row-major) over all nine columns
    // with one running sum per column, so each sum adds its terms in row order and the
    // inner loops have a fixed length (columns already reduced carry values nobody reads).
    int perm[NCOL];
    for (int c = 0; c < NCOL; c++) perm[c] = c;
    double cn[NAUG];  // squared column norms below the current row
    for (int c = 0; c < NAUG; c++) cn[c] = 0.0;
    for (int r = 0; r < rows; r++) {
        const double* row = M + r * NAUG;
        for (int c = 0; c < NAUG; c++) cn[c] += row[c] * row[c];
    }
    double r00 = 0.0;
    for (int k = 0; k < NCOL; k++) {
        // pivot: the remaining column of largest norm below row k (first of equals)
        int best = k;
        for (int c = k + 1; c < NCOL; c++)
            if (cn[c] > cn[best]) best = c;
        const double bn = cn[best];
        if (best != k) {
            for (int r = 0; r < rows; r++) {
                double* row = M + r * NAUG;
                const double t = row[k];
                row[k] = row[best];
                row[best] = t;
            }
            const int tp = perm[k]; perm[k] = perm[best]; perm[best] = tp;
        }
        const double norm = sqrt(bn);
        if (k == 0) r00 = norm;
        if (!(norm > 1e-12 * r00) || norm == 0.0) return 0;
        // Householder vector v = x - alpha e0 with alpha = -sign(x0) |x|; v's tail stays in column k
        double* top = M + k * NAUG;
        const double x0 = top[k];
        const double alpha = x0 >= 0.0 ? -norm : norm;
        const double v0 = x0 - alpha;
        const double vtv = v0 * v0 + (bn - x0 * x0);
        double dot[NAUG];
        for (int c = 0; c < NAUG; c++) dot[c] = v0 * top[c];
        for (int r = k + 1; r < rows; r++) {
            const double* row = M + r * NAUG;
            const double vk = row[k];
            for (int c = 0; c < NAUG; c++) dot[c] += vk * row[c];
        }
        for (int c = 0; c < NAUG; c++) dot[c] = vtv > 0.0 ? 2.0 * dot[c] / vtv : 0.0;
        for (int c = k + 1; c < NAUG; c++) top[c] -= dot[c] * v0;
        // apply the reflection to the rows below and gather the next step's column norms in the same sweep
        for (int c = 0; c < NAUG; c++) cn[c] = 0.0;
        for (int r = k + 1; r < rows; r++) {
            double* row = M + r * NAUG;
            const double vk = row[k];
            for (int c = 0; c < NAUG; c++) {
                const double t = row[c] - dot[c] * vk;
                row[c] = t;
                cn[c] += t * t;
            }
        }
        top[k] = alpha;
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

// ---- host-side pieces of the RANSAC loop, shared by the CPU and GPU drivers ----------
// Smallest consensus size for which the chance that a wrong model gathers it is
// below p_badxform (xform.cpp:246-281).  The reference's helper sums log(i) over
// [n-m, n] and over [n-i, n], one term more than a binomial coefficient needs, so
// the factorials are of n-m-1 and n-i-1, and its i = n term takes log(0) and
// comes out as exp(-inf) = 0.
// Up to kMinInliersFloatN matches the sums are accumulated exactly as the reference
// accumulates them: float additions of logf(i) in ascending i, each range summed
// from zero (about n^2/2 additions, which is why there is a limit), every operation
// of the expression in float and in its order.  Beyond the limit the same
// quantities are evaluated in double through lgamma; the result can then differ
// from the reference's by one at most (tested against the float form).
constexpr int kMinInliersFloatN = 8192;

inline int ransac_min_inliers_float(int n, int m, float p_badsupp, float p_badxform) {
    SIFT_NO_CONTRACT
    std::vector<float> lg((size_t)n + 1), fact((size_t)n + 1, 0.f), term((size_t)n + 1, 0.f);
    for (int i = 0; i <= n; i++) lg[(size_t)i] = logf((float)i);  // lg[0] = -inf, used by the i = n term only
    for (int i = 1; i <= n; i++) fact[(size_t)i] = fact[(size_t)i - 1] + lg[(size_t)i];  // log_factorial(i)
    auto log_range = [&](int a, int b) {  // _log_factorial(a, b)
        float f = 0.f;
        for (int i = a; i <= b; i++) f += lg[(size_t)i];
        return f;
    };
    const float log_n = fact[(size_t)n];
    const float log_nm = log_n - log_range(n - m, n);
    const float lb = logf(p_badsupp), lgood = logf(1.f - p_badsupp);
    float total = 0.f;
    for (int i = m + 1; i <= n; i++) {
        float pi = (float)(i - m) * lb + (float)(n - i + m) * lgood;
        pi = pi + log_nm;
        pi = pi - fact[(size_t)(i - m)];
        pi = pi - (log_n - log_range(n - i, n));
        term[(size_t)i] = expf(pi);
        total += term[(size_t)i];
    }
    float sum = total;
    int j = m + 1;
    for (; j <= n; j++) {
        sum -= term[(size_t)j];
        if (sum < p_badxform) break;
    }
    return j;
}

inline int ransac_min_inliers(int n, int m, float p_badsupp, float p_badxform) {
    if (n <= kMinInliersFloatN) return ransac_min_inliers_float(n, m, p_badsupp, p_badxform);
    const double lb = log((double)p_badsupp), lg = log(1.0 - (double)p_badsupp);
    const double log_nm = lgamma((double)(n - m));  // log((n-m-1)!)
    std::vector<double> term((size_t)(n > 0 ? n : 0) + 1, 0.0);
    double sum = 0.0;
    for (int i = m + 1; i < n; i++) {
        term[(size_t)i] = exp((double)(i - m) * lb + (double)(n - i + m) * lg + log_nm - lgamma((double)(i - m + 1)) -
                              lgamma((double)(n - i)));
        sum += term[(size_t)i];
        // far beyond the mode (about p_badsupp * n) the terms only shrink: once one has
        // underflowed to zero the rest are zero too and add nothing
        if (term[(size_t)i] == 0.0 && (double)(i - m) > 2.0 * (double)p_badsupp * (double)n + 16.0) break;
    }
    int j = m + 1;
    for (; j <= n; j++) {
        sum -= term[(size_t)j];
        if (sum < (double)p_badxform) break;
    }
    return j;
}

// Chance that k samples of m all held an outlier when a fraction `frac` of the
// matches are inliers; pow(float, int) promotes to double in the reference's
// expression (xform.cpp:61,94) and the result is stored as float.
inline float ransac_p_bad(float frac, int m, int k) {
    return (float)pow(1.0 - pow((double)frac, (double)m), (double)k);
}

// lsq_homog (xform.cpp:177-219) over the pairs idx[0..n) of xyuv (4 floats per
// pair: x, y -> u, v).  False when n < 4 or the system is rank deficient.
inline bool fit_pairs(const float* xyuv, const int* idx, int n, float* H9) {
    if (n < 4) return false;
    static thread_local std::vector<double> M;  // reused: the final fits take every consensus pair
    M.resize((size_t)2 * n * NAUG);
    for (int i = 0; i < n; i++) {
        const float* p = xyuv + (size_t)idx[i] * 4;
        dlt_rows(p[0], p[1], p[2], p[3], &M[(size_t)(2 * i) * NAUG], &M[(size_t)(2 * i + 1) * NAUG]);
    }
    double h[NCOL];
    if (!solve_lsq(M.data(), 2 * n, h)) return false;
    for (int k = 0; k < NCOL; k++) H9[k] = (float)h[k];
    H9[8] = 1.f;
    return true;
}

// Row order of the last fit.  The reference orders the consensus points by x and
// then by y with an exchange sort (xform.cpp:114-137); row order only moves the
// solution's rounding, so a stable sort on the same keys stands in for it.
inline std::vector<int> last_fit_order(const float* xyuv, const std::vector<int>& consensus) {
    std::vector<int> order = consensus;  // ascending, so the index breaks ties the way a stable sort would
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        const float ya = xyuv[(size_t)a * 4 + 1], yb = xyuv[(size_t)b * 4 + 1];
        if (ya != yb) return ya < yb;
        const float xa = xyuv[(size_t)a * 4], xb = xyuv[(size_t)b * 4];
        if (xa != xb) return xa < xb;
        return a < b;
    });
    return order;
}

}  // namespace sift_homog
