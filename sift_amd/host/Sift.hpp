// Drop-in for the reference's Sift/Sift.hpp: same macros, same `Sift` and
// `Stitching` classes with the same public members, so StitchImg
// (Sift_main.cpp:31-55), ransac_xform (ransac/xform.cpp:37) and blen2img
// (stitch/Blend.cpp:898) keep compiling against it.  Sift::pyramid() and the
// matcher inside Stitching::Check() run on the GPU through include/sift_hip.h.
#pragma once
#include <chrono>
#include <string>
#include <vector>

#include "Imgraw.hpp"
#include "imagedata.hpp"
#include "imglib.hpp"
#include "xform.hpp"

#define SIFT_INTVLS 3
#define SIFT_INTVLS_DIFF 3
#define SIFT_GauSigma 1.6f
#define SIFT_Dx 0.03f
#define SIFT_DESCR_SCL_FCTR 3.f
#define SIFT_DESCR_HIST_RADIUS 4
#define SIFT_DESCR_HIST_BINS 8
#define SIFT_DESCR_MAG_THR 0.2
#define SIFT_INT_DESCR_FCTR 512.0
#define SIFT_CURV_THR 10
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_IMG_BORDER 5
#define SIFT_CONTR_THR 0.05
#define KDTREE_BBF_MAX_NN_CHKS 200
#define NN_SQ_DIST_RATIO_THR 0.4

struct sift_hip_ctx;

namespace sift_amd {
// Process-wide GPU context used by the classes below (device = $SIFT_HIP_DEVICE or 0).
// Throws std::runtime_error when the library cannot get a gfx950 device.
sift_hip_ctx* default_context();
// The classes below share that one context, which like every sift_hip_ctx serves one caller at
// a time: each of their calls into the library holds this lock for its duration, so objects may
// be used from several threads (their GPU work is then serialised).
class ContextUse {
public:
    ContextUse();
    ~ContextUse();
    ContextUse(const ContextUse&) = delete;
    ContextUse& operator=(const ContextUse&) = delete;
    sift_hip_ctx* ctx;
};
// Overrides for parameters the reference fixes at compile time.
struct Options {
    int octaves = 0;    // 0 = reference's automatic count (Sift.cpp:130-139)
    int subpixel = 0;   // 1 = Lowe refinement (matrix.hpp:140-196), off in the reference
    int blur_mode = 0;  // 0 = exact reference arithmetic
    int textbook = 0;   // 1 = quality mode of sift_hip_params (not the reference's output)
    int ratio_form = 0; // 0 = d0 < d1*th (Sift.cpp:576), 1 = (d0/d1) < th (Sift.cpp:563)
    int ransac_on_gpu = 1; // Stitching::Check: 1 = sift_hip_ransac_homog, 0 = ransac_xform on the CPU (same result)
    int keep_descriptors = 1; // Sift::pyramid keeps the descriptor bytes on the device and Stitching::Match matches those
                              // (0 = descriptors packed from Feature[] and uploaded again; same result)
};
Options& options();
// Wall time of the steps of a host call, printed to stderr when SIFT_HOST_TIMING=1.
class StepTimer {
public:
    explicit StepTimer(const char* what);
    void step(const char* name);

private:
    const char* what_;
    std::chrono::steady_clock::time_point t0_;
    bool on_;
};
}  // namespace sift_amd

class Size_error : public std::runtime_error {
public:
    explicit Size_error(const std::string& str) : std::runtime_error(str) {}
};
class Out_of_imgRange : public std::runtime_error {
public:
    explicit Out_of_imgRange(const std::string& str) : std::runtime_error(str) {}
};

class Sift {
public:
    Sift(ImgRaw img, size_t intvls = SIFT_INTVLS);
    ~Sift();
    Sift(const Sift&) = delete;
    Sift& operator=(const Sift&) = delete;
    // Scale space, detection, orientation and descriptors of raw_img; fills
    // FeatStart[0..feaNum) in the reference's order.  Throws std::runtime_error
    // if the GPU path fails.
    void pyramid();

public:
    ImgRaw raw_img;
    size_t pyWidth = 3 + 2 + 1;
    std::vector<std::vector<ImgRaw>> pyrs;      // never filled (as in the reference)
    std::vector<std::vector<ImgRaw>> pyrs_dog;  // never filled
    Feature* FeatStart;  // contiguous Feature[feaNum] after pyramid(); owned
    Feature* FeatEnd;
    size_t feaNum = 0;
    // The descriptor bytes of FeatStart[0..feaNum) as pyramid() left them in device memory (opaque handle of the
    // C library, nullptr if none) and the array they describe.  Stitching::Match matches them directly while
    // FeatStart / feaNum still are what pyramid() produced; a Feature[] the caller built or whose descriptors the
    // caller edited must set dev_descr_of = nullptr (or options().keep_descriptors = 0) to be packed and uploaded.
    void* dev_descr = nullptr;
    const Feature* dev_descr_of = nullptr;
    size_t dev_descr_n = 0;
};

class Stitching {
public:
    Stitching(const Sift& desc1, const Sift& desc2);
    ~Stitching();
    Stitching(const Stitching&) = delete;
    Stitching& operator=(const Stitching&) = delete;
    // Sift.cpp:669-702.  Matching (the part on the GPU path): queries = image-2
    // features, references = image-1 features, result in feat2[i].fwd_match.
    // Then ransac_xform (xform.hpp; on the GPU unless options().ransac_on_gpu is 0)
    // over those matches; its results stay in the three members below, which the
    // reference keeps as locals and prints.
    // Then, for 24-bit images with a model, blen2img (Blend.hpp; projection and multi-band
    // blend on the GPU); the panorama stays in the member below (the reference writes it to
    // "dataResult\\N.bmp", Sift.cpp:699-701: file output is the caller's business here).
    void Check(float matchTh = NN_SQ_DIST_RATIO_THR);
    // The matching step of Check() alone; returns the number of accepted matches.
    size_t Match(float matchTh = NN_SQ_DIST_RATIO_THR);

    std::vector<float> HomogMat;     // image 2 -> image 1, row-major 3x3; empty if RANSAC found no model
    Feature** RANSAC_feat = nullptr; // consensus features (of image 2); owned
    int RANSAC_num = 0;
    ImgRaw Panorama;                 // what blen2img produced; empty for gray images or without a model

private:
    Feature *feat1, *feat2;
    const Sift *sift1, *sift2;  // where the features came from (for the descriptors kept on the device)
    size_t link_matches(const std::vector<int32_t>& best);  // best[i] -> feat2[i].fwd_match; returns the accepted count
    size_t Width, Height;
    ImgRaw stackImg;
    ImgRaw img1, img2;
    size_t feat1_Count, feat2_Count;
};
