// TEST INFRASTRUCTURE ONLY -- C entry points over the reference's OWN object
// code, used to pin the oracle and to generate golden vectors.
//
// oracle/Makefile compiles /root/reference/Sift/imglib/imglib.cpp,
// kdtree/kdtree.cpp and kdtree/minpq.cpp where they lie (no copies) and links
// them with this file into oracle/_ref/libsift_ref.so.  Only those three
// translation units build in this environment; Sift.cpp and Imgraw.cpp need
// headers that are absent here and are therefore not part of this library.
// This file only declares/uses the reference's public interfaces.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "imglib/imglib.hpp"  // Gaus, Scaling (reference header, via -I)
#include "imagedata.hpp"      // Feature
#include "kdtree.hpp"         // kdtree_build / kdtree_bbf_knn / kdtree_release

extern float descr_dist_sq(Feature* f1, Feature* f2);  // Sift/kdtree/kdtree.cpp:42

extern "C" {

int ref_gau_matrix(float p, int mat_len, float* out, int cap) {
    std::vector<float> g = Gaus::gau_matrix(p, (size_t)mat_len);
    if ((int)g.size() > cap) return -(int)g.size();
    std::memcpy(out, g.data(), g.size() * sizeof(float));
    return (int)g.size();
}

void ref_gau_blur(const float* in, float* out, int w, int h, float p) {
    std::vector<float> src(in, in + (size_t)w * h), dst;
    Gaus::GauBlur(dst, src, (size_t)w, (size_t)h, p);
    std::memcpy(out, dst.data(), dst.size() * sizeof(float));
}

// Scaling::first.  `in` must carry one padding row after the image: the
// reference's bilinear() may touch index ceil(y)*width + ceil(x) one row past
// the end with weight 0 (a caller-side guard, not a change to its arithmetic).
void ref_first(const float* in, int w, int h, float ratio, float* out) {
    std::vector<float> src(in, in + (size_t)w * (h + 1)), dst;
    Scaling::first(dst, src, (size_t)w, (size_t)h, ratio);
    std::memcpy(out, dst.data(), dst.size() * sizeof(float));
}

// The loop of kdtreeMatchFeat (Sift/Sift.cpp:568-584) driven over the
// reference's kd-tree: queries q (nq x 128), tree built over r (nr x 128).
// best[i] = index into r (original order) or -1.  The tree reorders its
// feature array in place, so the original index rides in Feature::x.
void ref_kdtree_match(const float* q, int nq, const float* r, int nr, float th, int max_chks,
                      int32_t* best) {
    std::vector<Feature> fq(nq), fr(nr);
    for (int i = 0; i < nq; i++) {
        std::memcpy(fq[i].descr, q + (size_t)i * 128, 128 * sizeof(float));
        fq[i].d = 128;
        fq[i].x = i;
    }
    for (int i = 0; i < nr; i++) {
        std::memcpy(fr[i].descr, r + (size_t)i * 128, 128 * sizeof(float));
        fr[i].d = 128;
        fr[i].x = i;
    }
    kd_node* root = kdtree_build(fr.data(), nr);
    for (int i = 0; i < nq; i++) {
        Feature** nbrs = nullptr;
        best[i] = -1;
        int k = kdtree_bbf_knn(root, &fq[i], 2, &nbrs, max_chks);
        if (k == 2) {
            float d0 = descr_dist_sq(&fq[i], nbrs[0]);
            float d1 = descr_dist_sq(&fq[i], nbrs[1]);
            if (d0 < d1 * th) best[i] = nbrs[0]->x;
        }
        delete[] nbrs;
    }
    kdtree_release(root);
}

}  // extern "C"
