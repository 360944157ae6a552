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

// find_consensus (xform.cpp:365-395): indices of the pairs H moves to within err_tol.
std::vector<int> consensus_of(const std::vector<float>& xyuv, const float* H, float err_tol) {
    std::vector<int> in;
    for (int i = 0; i < (int)(xyuv.size() / 4); i++) {
        const float* p = &xyuv[(size_t)i * 4];
        // a NaN error (point on the line H sends to infinity) is not an inlier, as `err <= err_tol` is false
        if (sift_homog::homog_xfer_err(H, p[0], p[1], p[2], p[3]) <= err_tol) in.push_back(i);
    }
    return in;
}

}  // namespace

unsigned& sift_amd::ransac_seed() {
    static unsigned seed = 0;
    return seed;
}

int sift_amd::ransac_min_inliers(int n, int m, float p_badsupp, float p_badxform) {
    return sift_homog::ransac_min_inliers(n, m, p_badsupp, p_badxform);
}

std::vector<float> ransac_xform(Feature* features, int n, int m, float p_badxform, float err_tol, Feature*** inliers,
                                int& n_in) {
    const unsigned seed = sift_amd::ransac_seed();
    srand(seed ? seed : (unsigned)time(nullptr));  // xform.cpp:42

    // get_matched_features (xform.cpp:222-244).  The reference parks a `sampled`
    // flag in feature_data and restores it on return; a side array does the same
    // without touching the features.  Positions are read through rX()/rY()
    // (xform.cpp:326-327): 4 floats per pair, x, y -> u, v.
    std::vector<float> xyuv;
    std::vector<Feature*> feat;
    for (int i = 0; i < n; i++) {
        Feature* f = features + i;
        if (const Feature* mt = f->fwd_match) {
            xyuv.insert(xyuv.end(), {f->rX(), f->rY(), mt->rX(), mt->rY()});
            feat.push_back(f);
        }
    }
    const int nm = (int)feat.size();
    if (nm < m || m < 4) return {};  // "not enough matches to compute xform" (xform.cpp:53-56)

    const int in_min = sift_amd::ransac_min_inliers(nm, m, kProbBadSupp, p_badxform);
    float in_frac = (float)kInlierFracEst;
    int k = 0;
    float p = sift_homog::ransac_p_bad(in_frac, m, k);

    std::vector<char> sampled((size_t)nm);
    std::vector<int> sample((size_t)m), best;
    std::vector<float> M(9);
    while (p > p_badxform) {
        // draw_ransac_sample (xform.cpp:284-306): m distinct pairs, redrawing on repeats
        std::fill(sampled.begin(), sampled.end(), 0);
        for (int i = 0; i < m; i++) {
            int x;
            do x = rand() % nm; while (sampled[(size_t)x]);
            sample[(size_t)i] = x;
            sampled[(size_t)x] = 1;
        }
        if (sift_homog::fit_pairs(xyuv.data(), sample.data(), m, M.data())) {
            std::vector<int> in = consensus_of(xyuv, M.data(), err_tol);
            if (in.size() > best.size()) {  // strictly larger: the first of equals stays (xform.cpp:77)
                best.swap(in);
                in_frac = (float)best.size() / (float)nm;
            }
        }
        p = sift_homog::ransac_p_bad(in_frac, m, ++k);
    }

    if (inliers) *inliers = nullptr;
    if ((int)best.size() < in_min) return {};  // xform.cpp:150-156
    // xform.cpp:101-146: refit on the best consensus, gather that model's consensus, fit again
    if (!sift_homog::fit_pairs(xyuv.data(), best.data(), (int)best.size(), M.data())) return {};
    const std::vector<int> fin = consensus_of(xyuv, M.data(), err_tol);
    const std::vector<int> order = sift_homog::last_fit_order(xyuv.data(), fin);
    if (!sift_homog::fit_pairs(xyuv.data(), order.data(), (int)order.size(), M.data())) M.clear();
    if (inliers) {
        Feature** out = new Feature*[(size_t)std::max(nm, 1)];
        for (size_t i = 0; i < fin.size(); i++) out[i] = feat[(size_t)fin[i]];
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
