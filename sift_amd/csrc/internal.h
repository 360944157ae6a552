// Internal declarations shared by the HIP translation units of libsift_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sift_hip.h"

namespace sifthip {

// ---- device-side records ------------------------------------------------------
struct Cand {  // an extremum that passed the contrast and edge tests
    int32_t x, y, layer;
};

// Slots of an octave's device counter block.
enum Counter {
    CNT_NCAND = 0,        // candidates stored for this octave (<= capacity)
    CNT_NCAND_TOTAL = 1,  // candidates found for this octave
    CNT_OCT_BEGIN = 2,    // index of this octave's first feature in the output array
    CNT_OCT_END = 3,      // one past its last feature (may exceed the capacity)
    CNT_SLOTS = 8
};
// Slots of the per-context (all octaves) counter block.
// Every slot an octave owns is written unconditionally by that octave's kernels on
// every frame, so the block is never cleared: no stream has to wait for another
// stream's reset before it may start.
enum GlobalCounter {
    G_MAX_OCT = 32,
    G_OCT_FEATS = 0,             // + o: features found in octave o (may exceed the capacity)
    G_OCT_CAND = G_MAX_OCT,      // + o: candidates found in octave o (may exceed the candidate table)
    G_OCT_CAND_TOP = 2 * G_MAX_OCT,  // + o: tentative top-layer candidates of octave o (same table size)
    G_SLOTS = 3 * G_MAX_OCT
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// Octaves are independent (Sift.cpp:145-149: every octave restarts from the
// original image), so each one has its own workspace and feature buffer and runs
// on one of a few HIP streams; the octave-major output order of the reference is
// restored by the order of the device-to-host copies.
struct OctWs {
    hipStream_t stream = nullptr;  // where this octave's launches go during the current enqueue (not owned)
    bool lazy_top = false;         // the current frame evaluates the last Gaussian level only at top-layer extrema
    DevBuf base, tmp, gau_all, masks, row_cnt, row_off, cand, cand_mask, cand_mm, cand_off, counters, feat, hist;
    DevBuf cand_sub, feat_sub;  // sub-pixel offsets per candidate (float4) / per feature (3 floats)
    DevBuf row_off2, cand2, counters2;  // tentative top-layer candidates (the last level is built only there)
    DevBuf feat_alt, feat_sub_alt;  // second output set: a batch fetches frame f-1 while frame f runs
};

struct RansacWs {  // device workspace of the RANSAC entry points
    DevBuf blk, pairs, src, flags, samples, models, counts, qf, rf, best;
};

struct OctavePlan {        // host-side constants of one octave (Sift.cpp:145-183)
    float curr_size;       // 2, 1, 0.5 ...
    float coef;            // 1, 2, 4 ...
    int W, H;
    std::vector<float> sigma;      // per Gaussian level
    std::vector<int> tap_off;      // offset of each level's taps in the tap table
    std::vector<int> tap_len;
    std::vector<int> radius;       // per DoG layer 1..intvls: window radius r (Sift.cpp:177)
    std::vector<float> curr_sigma; // per layer: currSigma (Sift.cpp:183)
    std::vector<int> wts_off;      // per layer: offset of the orientation weights
};

// Launch-shape switches for A/B measurements and tests.  They are read from the environment
// once, when the context is created (SIFT_BLUR_SEG, SIFT_BLUR_STREAM, SIFT_BLUR_FUSED, SIFT_BLUR_SWEEP,
// SIFT_MATCH_NB, SIFT_HIP_INTERLEAVE, SIFT_O3_PLACE), and changed on a live context only through
// sift_hip_set_tuning: no launch path looks at the environment.
struct Tuning {
    int blur_seg = 0;      // streaming column pass: rows per segment (0 = chosen per image)
    int blur_stream = -1;  // streaming column pass: -1 = where it pays, 0 = never, 1 = wherever an instance exists
    int blur_fused = -1;   // one-launch blur: -1 = where it pays, else the longest kernel that takes it at any image size
    int blur_sweep = -1;   // sweeping one-launch blur (float arithmetic): -1 = where it pays, 0 = never, n > 0 = wherever an instance exists, n rows per segment
    int match_nb = 0;      // matcher: 32-query blocks per wave, 0 = by query count, else 2 or 4
    int interleave = 1;    // single frames: pyramids of the streams' first octaves enqueued level by level in turn
    int o3_place = 3;      // single frames of four octaves and more: where octave 3 runs (see enqueue_extract)
};

struct Plan {
    int w = 0, h = 0;
    sift_hip_params prm{};
    std::vector<OctavePlan> oct;
    std::vector<float> table;  // taps and orientation weights, concatenated
    bool valid = false;
};

}  // namespace sifthip

struct sift_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // octave 1 of a frame (urgent priority); never waits on another stream
    hipStream_t side_stream2 = nullptr;  // octaves 2 and 3: many short kernels that fill the gaps of the other two
    hipStream_t side_stream3 = nullptr;  // octaves >= 4 (frames processed with more than four octaves): a chain of
                                         // long-kernel, few-pixel launches that would otherwise end the frame alone
    int n_side = 2;                      // side streams in use (SIFT_HIP_NSIDE = 1: octaves >= 1 all on side_stream)
    hipStream_t copy_stream = nullptr;
    bool side_busy = false;             // the last enqueue put work on side_stream
    bool pipelined = false;             // inside the batch pipeline: the device never idles between frames
    std::string last_error;

    sifthip::Plan plan;
    // device workspace (grown on demand, never shrunk)
    sifthip::DevBuf in_img, in_rgb, tmp, table, layer_consts, gcounters, qi8, ri8, qnorm, rconst, best,
        d1, d2, io_a, io_b, frame2, dog_all;
    std::vector<sifthip::OctWs> lanes;  // lanes[0] runs on `stream`
    hipEvent_t ev_start = nullptr;
    int cand_cap = 0;
    // pinned host staging
    int32_t* h_counters = nullptr;      // pinned; the current output set's counters
    int32_t* h_counters_alt = nullptr;  // ... and the other set's
    sifthip::DevBuf gcounters_alt;
    sifthip::DevBuf gathered;  // all ranks' reference descriptors (sharded match)
    sifthip::DevBuf gather_in, gather_pad;  // ragged shards: this rank's padded contribution, all ranks' padded shards
    unsigned char* match_scratch = nullptr;  // this device's copy of the matcher's static work area (not owned)
    sifthip::RansacWs ransac;
    sifthip::DevBuf warp_src, warp_dst, warp_edges;
    sifthip::DevBuf pano_src, pano_canvas;       // sift_hip_panorama_pair: the two uploaded images; the canvas
    sifthip::DevBuf pano_f32;                    // sift_hip_panorama_pair_f32: float samples of one image on their way to 8 bits
    sifthip::DevBuf blend_ws, blend_l, blend_r;  // multiBandBlend: pyramids and scratch; device copies of host images
    hipEvent_t ev_frame[2] = {nullptr, nullptr};  // end of a batch frame's work on `stream` ...
    hipEvent_t ev_side[2] = {nullptr, nullptr};   // ... on `side_stream`
    hipEvent_t ev_side2[2] = {nullptr, nullptr};  // ... on `side_stream2`
    hipEvent_t ev_side3[2] = {nullptr, nullptr};  // ... and on `side_stream3`
    std::vector<float> h_sub;  // offsets of the last extract with subpixel = 1 (3 per feature)
    // where the records of the last extract that delivered features still sit in device memory: per octave the
    // array and the number of records that went to the caller (sift_hip_keep_descriptors); empty = nothing to keep
    std::vector<const sift_hip_feat*> last_feat_ptr;
    std::vector<int> last_feat_cnt;
    sifthip::DevBuf h_feat, h_hist;

    sifthip::Tuning tune;

    // profiling
    bool profiling = false;
    bool force_serial = false;
    bool timeline = false;  // SIFT_HIP_TIMELINE=1: events around every launch group with the streams left concurrent
    bool full_top_level = false;  // SIFT_HIP_FULL_TOP=1: build the last Gaussian level as an image (A/B measurements)
    int side_priority = 1;          // 1: the small-octave stream is created with the urgent priority
    struct Ev { int cls; hipEvent_t a, b; hipStream_t st; };
    std::vector<Ev> events;
    std::vector<hipEvent_t> free_events;
    float prof_ms[SIFT_HIP_K_COUNT] = {0};
    int prof_n[SIFT_HIP_K_COUNT] = {0};
};

namespace sifthip {

// Returns false and records the message in ctx->last_error on failure.
bool hip_ok(sift_hip_ctx* ctx, hipError_t e, const char* what);
int ensure(sift_hip_ctx* ctx, DevBuf& b, size_t bytes);
void join_all(sift_hip_ctx* ctx);  // host wait for every compute stream of the context with work pending
int ensure_pinned(sift_hip_ctx* ctx, DevBuf& b, size_t bytes);
void prof_begin(sift_hip_ctx* ctx, int cls, hipStream_t st);
void prof_end(sift_hip_ctx* ctx, hipStream_t st);

struct ProfScope {
    sift_hip_ctx* c;
    hipStream_t st;
    ProfScope(sift_hip_ctx* ctx, int cls, hipStream_t s) : c(ctx), st(s) { if (c->profiling || c->timeline) prof_begin(c, cls, st); }
    ~ProfScope() { if (c->profiling || c->timeline) prof_end(c, st); }
};

// ---- launchers (defined next to their kernels) --------------------------------
// pyramid.hip
void launch_gray(sift_hip_ctx* ctx, hipStream_t st, const float* rgb, float* gray, size_t n);
void launch_first(sift_hip_ctx* ctx, hipStream_t st, const float* src, int w, int h, float ratio, float* dst, int nw,
                  int nh);
// One separable blur: src -> (tmp) -> dst.
void launch_blur(sift_hip_ctx* ctx, hipStream_t st, const float* src, float* tmp, float* dst, int W, int H,
                 const float* d_taps, int K, int blur_mode);
// dog[k] = gau[k] - gau[k+1] for k < nlev-1 (only the octave_levels entry point stores DoG images)
void launch_dog(sift_hip_ctx* ctx, hipStream_t st, const float* gau, float* dog, size_t PX, int nlev);
// detect.hip
struct LayerConsts {  // per DoG layer 1..intvls of one octave
    int32_t radius;       // histogram window radius r (Sift.cpp:177)
    int32_t wts_off;      // offset of the 2r+1 orientation weights in the table (floats)
    float curr_sigma;     // currSigma (Sift.cpp:183)
    float sigma_oct;      // (float)(int)curr_sigma, what FeatAppend stores
};
// ws.gau_all: level-major Gaussian images of the octave, PX pixels per level
// (DoG values are recomputed from adjacent levels, never stored).
void launch_detect(sift_hip_ctx* ctx, OctWs& ws, int octave, size_t PX, int W, int H, int intvls,
                   const LayerConsts* d_lc, float thr, int cand_cap, const float* d_top_taps, int top_K);
void launch_refine(sift_hip_ctx* ctx, OctWs& ws, size_t PX, int W, int H, int intvls, const LayerConsts* d_lc,
                   float contr_thr);
void launch_orient(sift_hip_ctx* ctx, OctWs& ws, size_t PX, int W, const LayerConsts* d_lc, const float* d_table);
void launch_emit(sift_hip_ctx* ctx, OctWs& ws, int octave, float curr_size, const LayerConsts* d_lc, int feat_cap,
                 bool with_sub, bool textbook);
// descriptor.hip; max_window: sample count (2R+1)^2 of the octave's largest descriptor window
void launch_descr(sift_hip_ctx* ctx, OctWs& ws, size_t PX, int W, int H, const float* d_cs, int feat_cap,
                  float* d_hist, int max_window, bool textbook);
// match.hip
int run_match(sift_hip_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, float ratio,
              int ratio_form, int32_t* d_best, int32_t* d_d1, int32_t* d_d2);
// The same launches with the work areas in the library's static device array (no allocation);
// match_scratch_init looks that array's address up for ctx->device (which must be current).
int match_scratch_init(sift_hip_ctx* ctx);
int run_match_noalloc(sift_hip_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, float ratio,
                      int ratio_form, int32_t* d_best, int32_t* d_d1, int32_t* d_d2);

// warp.hip
void launch_warp_perspective(sift_hip_ctx* ctx, hipStream_t st, const uint8_t* d_src, int s_col, int s_row,
                             const float* H9, uint8_t* d_dst, int d_col, int d_row);
void launch_warp_cylinder(sift_hip_ctx* ctx, hipStream_t st, const uint8_t* d_src, int w, int h, float focal,
                          uint8_t* d_out, float* d_upedge, float* d_downedge);
// blend.hip: multiBandBlend on two device-resident 8-bit RGB images, updated in place
int run_multiband_blend(sift_hip_ctx* ctx, uint8_t* d_l, int lw, int lh, uint8_t* d_r, int rw, int rh, int dx, int dy);
void launch_compose_pair(sift_hip_ctx* ctx, hipStream_t st, const uint8_t* d_a, int ca, int ra, const uint8_t* d_b, int cb,
                         int rb, int top0, int left1, int top1, uint8_t* d_canvas, int cols, int rows);
// descriptor bytes of n feature records, packed 128 per feature (match.hip)
void launch_gather_descr(sift_hip_ctx* ctx, hipStream_t st, const sift_hip_feat* d_feat, int n, uint8_t* d_out);
// n float samples to 8 bits the way ImgRaw's 8-bit copy does (Imgraw.hpp:59-82); d_dst 4-byte aligned
void launch_float_to_u8(sift_hip_ctx* ctx, hipStream_t st, const float* d_src, size_t n, bool normalised, uint8_t* d_dst);
// ransac.hip; features and best in device memory, outputs in host memory
int run_ransac(sift_hip_ctx* ctx, const sift_hip_feat* d_q, int nq, const sift_hip_feat* d_r, int nr,
               const int32_t* d_best, int m, float p_badxform, float err_tol, unsigned seed, float* H9,
               int32_t* inlier_idx, int32_t* n_in, int32_t* iterations);
int run_ransac_pairs(sift_hip_ctx* ctx, const float* xyuv, int n, int m, float p_badxform, float err_tol,
                     unsigned seed, float* H9, int32_t* inlier_idx, int32_t* n_in, int32_t* iterations);

}  // namespace sifthip
