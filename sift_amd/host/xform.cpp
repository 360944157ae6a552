// RANSAC homography between matched features: the consumer right after the
// matcher in Stitching::Check (reference: Sift/ransac/xform.cpp).  Host code;
// the least-squares solve and the transfer error come from csrc/homog.h so a
// device scorer can run the very same arithmetic.
#include "xform.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <limits>

#include "../csrc/homog.h"
#include "imagedata.hpp"

namespace {

constexpr double kInlierFracEst = 0.25;  // RANSAC_INLIER_FRAC_EST, xform.cpp:22
constexpr float kProbBadSupp = 0.10f;    // RANSAC_PROB_BAD_SUPP, xform.cpp:24

// One correspondence: the feature's source-image position and its match's
// (xform.cpp:326-327 reads both through rX()/rY()).
struct Pair {
    float x, y, u, v;
    Feature* feat;
};

// lsq_homog (xform.cpp:177-219) over pairs picked by `idx`; empty on failure.
std::vector<float> fit(const std::vector<Pair>& pairs, const std::vector<int>& idx) {
    const int n = (int)idx.size();
    if (n < 4) return {};
    std::vector<double> M((size_t)2 * n * sift_homog::NAUG);
    for (int i = 0; i < n; i++) {
        const Pair& p = pairs[(size_t)idx[(size_t)i]];
        sift_homog::dlt_rows(p.x, p.y, p.u, p.v, &M[(size_t)(2 * i) * sift_homog::NAUG],
                             &M[(size_t)(2 * i + 1) * sift_homog::NAUG]);
    }
    double h[8];
    if (!sift_homog::solve_lsq(M.data(), 2 * n, h)) return {};
    std::vector<float> H(9);
    for (int k = 0; k < 8; k++) H[(size_t)k] = (float)h[k];
    H[8] = 1.f;
    return H;
}

// find_consensus (xform.cpp:365-395): indices of the pairs H moves to within err_tol.
std::vector<int> consensus_of(const std::vector<Pair>& pairs, const std::vector<float>& H, float err_tol) {
    std::vector<int> in;
    for (int i = 0; i < (int)pairs.size(); i++) {
        const Pair& p = pairs[(size_t)i];
        // a NaN error (point on the line H sends to infinity) is not an inlier, as `err <= err_tol` is false
        if (sift_homog::homog_xfer_err(H.data(), p.x, p.y, p.u, p.v) <= err_tol) in.push_back(i);
    }
    return in;
}

}  // namespace

unsigned& sift_amd::ransac_seed() {
    static unsigned seed = 0;
    return seed;
}

int sift_amd::ransac_min_inliers(int n, int m, float p_badsupp, float p_badxform) {
    // xform.cpp:246-281.  Its helper sums log(i) over [n-m, n] and over [n-i, n], one
    // term more than a binomial coefficient needs, so the factorials below are of
    // n-m-1 and n-i-1; the i = n term takes log(0) and comes out as exp(-inf) = 0.
    const double lb = std::log((double)p_badsupp), lg = std::log(1.0 - (double)p_badsupp);
    const double log_nm = std::lgamma((double)(n - m));  // log((n-m-1)!)
    std::vector<double> term((size_t)std::max(n, 0) + 1, 0.0);
    double sum = 0.0;
    for (int i = m + 1; i <= n; i++) {
        if (i < n) {
            const double pi = (double)(i - m) * lb + (double)(n - i + m) * lg + log_nm - std::lgamma((double)(i - m + 1)) -
                              std::lgamma((double)(n - i));
            term[(size_t)i] = std::exp(pi);
        }
        sum += term[(size_t)i];
    }
    int j = m + 1;
    for (; j <= n; j++) {
        sum -= term[(size_t)j];
        if (sum < (double)p_badxform) break;
    }
    return j;
}

std::vector<float> ransac_xform(Feature* features, int n, int m, float p_badxform, float err_tol, Feature*** inliers,
                                int& n_in) {
    const unsigned seed = sift_amd::ransac_seed();
    srand(seed ? seed : (unsigned)time(nullptr));  // xform.cpp:42

    // get_matched_features (xform.cpp:222-244).  The reference parks a `sampled`
    // flag in feature_data and restores it on return; a side array does the same
    // without touching the features.
    std::vector<Pair> pairs;
    for (int i = 0; i < n; i++) {
        Feature* f = features + i;
        if (const Feature* mt = f->fwd_match) pairs.push_back({f->rX(), f->rY(), mt->rX(), mt->rY(), f});
    }
    const int nm = (int)pairs.size();
    if (nm < m || m < 4) return {};  // "not enough matches to compute xform" (xform.cpp:53-56)

    const int in_min = sift_amd::ransac_min_inliers(nm, m, kProbBadSupp, p_badxform);
    // pow(float, int) promotes to double in the reference's expressions (xform.cpp:61,94)
    auto p_bad = [m](float frac, int k) {
        return (float)std::pow(1.0 - std::pow((double)frac, (double)m), (double)k);
    };
    float in_frac = (float)kInlierFracEst;
    int k = 0;
    float p = p_bad(in_frac, k);

    std::vector<char> sampled((size_t)nm);
    std::vector<int> sample((size_t)m), best;
    while (p > p_badxform) {
        // draw_ransac_sample (xform.cpp:284-306): m distinct pairs, redrawing on repeats
        std::fill(sampled.begin(), sampled.end(), 0);
        for (int i = 0; i < m; i++) {
            int x;
            do x = rand() % nm; while (sampled[(size_t)x]);
            sample[(size_t)i] = x;
            sampled[(size_t)x] = 1;
        }
        const std::vector<float> M = fit(pairs, sample);
        if (!M.empty()) {
            std::vector<int> in = consensus_of(pairs, M, err_tol);
            if (in.size() > best.size()) {  // strictly larger: the first of equals stays (xform.cpp:77)
                best.swap(in);
                in_frac = (float)best.size() / (float)nm;
            }
        }
        p = p_bad(in_frac, ++k);
    }

    if ((int)best.size() < in_min) {  // xform.cpp:150-156
        if (inliers) *inliers = nullptr;
        return {};
    }
    // xform.cpp:101-146: refit on the best consensus, gather that model's consensus, fit again
    std::vector<float> M = fit(pairs, best);
    if (M.empty()) {
        if (inliers) *inliers = nullptr;
        return {};
    }
    const std::vector<int> fin = consensus_of(pairs, M, err_tol);
    // The reference orders the points by x and then by y with an exchange sort before
    // the last fit (xform.cpp:114-137); row order only moves the solution's rounding.
    std::vector<int> order = fin;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pairs[(size_t)a].x < pairs[(size_t)b].x; });
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pairs[(size_t)a].y < pairs[(size_t)b].y; });
    M = fit(pairs, order);
    if (inliers) {
        Feature** out = new Feature*[(size_t)std::max(nm, 1)];
        for (size_t i = 0; i < fin.size(); i++) out[i] = pairs[(size_t)fin[i]].feat;
        *inliers = out;
    }
    if (n_in == 0) n_in = (int)fin.size();
    return M;
}

// Plain-array entry for bindings and tests: features of image 2 (queries) and
// image 1 given as octave coordinates and octave scales, `match[i]` the index in
// image 1 matched to feature i of image 2 or -1.  Returns the consensus size or
// -1 when no model was found; H9 and inlier_idx (capacity n2, indices into image
// 2) may be null.
extern "C" int sift_host_ransac(const int32_t* x2, const int32_t* y2, const float* size2, int n2, const int32_t* x1,
                                const int32_t* y1, const float* size1, int n1, const int32_t* match, int m,
                                float p_badxform, float err_tol, unsigned seed, float* H9, int32_t* inlier_idx) {
    std::vector<Feature> f1((size_t)std::max(n1, 1)), f2((size_t)std::max(n2, 1));
    for (int i = 0; i < n1; i++) {
        f1[(size_t)i].x = x1[i];
        f1[(size_t)i].y = y1[i];
        f1[(size_t)i].size = size1[i];
    }
    for (int i = 0; i < n2; i++) {
        f2[(size_t)i].x = x2[i];
        f2[(size_t)i].y = y2[i];
        f2[(size_t)i].size = size2[i];
        f2[(size_t)i].fwd_match = (match[i] >= 0 && match[i] < n1) ? &f1[(size_t)match[i]] : nullptr;
    }
    const unsigned keep = sift_amd::ransac_seed();
    sift_amd::ransac_seed() = seed;
    Feature** in = nullptr;
    int n_in = 0;
    const std::vector<float> H = ransac_xform(f2.data(), n2, m, p_badxform, err_tol, &in, n_in);
    sift_amd::ransac_seed() = keep;
    if (H.empty()) {
        delete[] in;
        return -1;
    }
    if (H9) std::copy(H.begin(), H.end(), H9);
    if (inlier_idx)
        for (int i = 0; i < n_in; i++) inlier_idx[i] = (int32_t)(in[i] - f2.data());
    delete[] in;
    return n_in;
}

extern "C" int sift_host_ransac_min_inliers(int n, int m, float p_badsupp, float p_badxform) {
    return sift_amd::ransac_min_inliers(n, m, p_badsupp, p_badxform);
}
