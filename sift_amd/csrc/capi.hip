// C ABI of libsift_hip.so (declared in include/sift_hip.h): context, workspace,
// the host-side constants of the pyramid (sigmas, taps, radii) and the launch
// sequence of the extract and match paths.  No CPU fallback exists: every entry
// point that does image or descriptor arithmetic runs the HIP kernels.
#include <algorithm>
#include <cctype>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <dlfcn.h>

#include "internal.h"

using namespace sifthip;

namespace sifthip {

bool hip_ok(sift_hip_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    if (ctx) ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

// Waits on the host for everything the context has enqueued on its compute streams.
void join_all(sift_hip_ctx* ctx) {
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->side_busy) {
        if (ctx->side_stream) hipStreamSynchronize(ctx->side_stream);
        if (ctx->side_stream2) hipStreamSynchronize(ctx->side_stream2);
        if (ctx->side_stream3) hipStreamSynchronize(ctx->side_stream3);
    }
}

int ensure(sift_hip_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return SIFT_HIP_OK;
    if (b.p) {
        join_all(ctx);
        hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        ctx->last_error = "hipMalloc failed for " + std::to_string(want) + " bytes";
        return SIFT_HIP_ERR_NOMEM;
    }
    b.cap = want;
    return SIFT_HIP_OK;
}

int ensure_pinned(sift_hip_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return SIFT_HIP_OK;
    if (b.p) {
        hipStreamSynchronize(ctx->stream);
        hipHostFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        ctx->last_error = "hipHostMalloc failed";
        return SIFT_HIP_ERR_NOMEM;
    }
    b.cap = want;
    return SIFT_HIP_OK;
}

void prof_begin(sift_hip_ctx* ctx, int cls, hipStream_t st) {
    sift_hip_ctx::Ev ev;
    ev.cls = cls;
    for (hipEvent_t* e : {&ev.a, &ev.b}) {
        if (!ctx->free_events.empty()) {
            *e = ctx->free_events.back();
            ctx->free_events.pop_back();
        } else {
            hipEventCreate(e);
        }
    }
    ev.st = st;
    hipEventRecord(ev.a, st);
    ctx->events.push_back(ev);
}

void prof_end(sift_hip_ctx* ctx, hipStream_t st) { hipEventRecord(ctx->events.back().b, st); }

// ---- host constants ------------------------------------------------------------
// Gaus::gau_matrix / Gaus::gau_meth, imglib/imglib.cpp:120-155.  The taps decide
// which pixels become keypoints, so every operation keeps the operand types of the
// reference (they fix the roundings): the exponent's numerator and the division in
// double, the 2*sigma^2 term in float, the result of exp stored as float, the
// 1/(sqrt(2*pi)*sigma) factor applied as one double division rounded to float.
static float gaussian_tap(int dist, float sigma) {
    const float two = 2.0f;
    const double dist_sq = std::pow((double)(size_t)dist, (double)two);
    const float two_sigma_sq = two * std::pow(sigma, two);  // float overload of pow
    float tap = (float)std::exp(-dist_sq / two_sigma_sq);
    tap = (float)(tap / (std::sqrt(two * 3.14159265358979323846) * sigma));
    return tap;
}

// Tap count: round(6*sigma + 1) forced odd unless the caller fixes it.  The kernel
// is built from its centre outwards; the normalising sum is a float that receives
// the centre tap, then twice each further tap, in order of distance.
static std::vector<float> gau_matrix(float sigma, size_t mat_len) {
    if (mat_len == 0) mat_len = (int)(std::round((sigma * 6 + 1))) | 1;
    if (mat_len % 2 == 0) ++mat_len;
    const int half = (int)(mat_len / 2);
    std::vector<float> from_centre((size_t)half + 1);
    float norm = 0;
    for (int dist = 0; dist <= half; ++dist) {
        const float tap = gaussian_tap(dist, sigma);
        from_centre[dist] = tap;
        norm += dist ? tap + tap : tap;
    }
    std::vector<float> g(mat_len);
    for (int k = 0; k < (int)mat_len; ++k) g[k] = from_centre[std::abs(k - half)] / norm;
    return g;
}

// getPyramidOctv, Sift.cpp:130-139
static int pyramid_octaves(size_t width, size_t height) {
    int n = 1;
    for (float Limit = 1.0; Limit < width || Limit < height; ++n) Limit = powf(2.0, n);
    if (n < 7) n = 1; else n -= 6;
    return n;
}

static bool same_params(const sift_hip_params& a, const sift_hip_params& b) {
    return a.octaves == b.octaves && a.intvls == b.intvls && a.sigma0 == b.sigma0 && a.dx_thr == b.dx_thr &&
           a.subpixel == b.subpixel && a.blur_mode == b.blur_mode && a.contr_thr == b.contr_thr && a.textbook == b.textbook;
}

// Everything Sift::pyramid derives from (w, h, params) before touching pixels.
static int build_plan(sift_hip_ctx* ctx, int w, int h, const sift_hip_params& prm) {
    Plan& pl = ctx->plan;
    if (pl.valid && pl.w == w && pl.h == h && same_params(pl.prm, prm)) return SIFT_HIP_OK;
    pl = Plan();
    pl.w = w;
    pl.h = h;
    pl.prm = prm;
    const int pyWidth = prm.intvls + 3;
    const int octvs = prm.octaves > 0 ? prm.octaves : pyramid_octaves((size_t)w, (size_t)h);
    // cos/sin of the 36 possible orientations, taken in DEGREES as radians (Sift.cpp:351-352)
    // (quality mode: the orientation in radians)
    const bool tb = prm.textbook != 0;
    auto frame_angle = [&](int k) -> float { return tb ? (float)(k * 10) * (float)(3.14159265358979323846 / 180.0) : (float)(k * 10); };
    for (int k = 0; k < 36; k++) pl.table.push_back(std::cos(frame_angle(k)));
    for (int k = 0; k < 36; k++) pl.table.push_back(std::sin(frame_angle(k)));
    // Sift.cpp:369-372 wraps (gradient angle - ori) into [0, 2*pi) by repeated float
    // additions of 2*pi; with ori up to 350 that is up to 56 roundings on a coarse
    // grid.  The net shift those additions give an angle of 0 is tabulated per
    // orientation as a two-float constant; the kernel adds it in one step.
    {
        const float PI2 = (float)(2.f * 3.14159265358979323846);
        double shift[36];
        for (int k = 0; k < 36; k++) {
            const float x0 = 0.f - frame_angle(k);
            float x = x0;
            while (x < 0.f) x += PI2;
            while (x >= PI2) x -= PI2;
            shift[k] = (double)x - (double)x0;
        }
        for (int k = 0; k < 36; k++) pl.table.push_back((float)shift[k]);
        for (int k = 0; k < 36; k++) pl.table.push_back((float)(shift[k] - (double)(float)shift[k]));
    }
    float curr_size = 2.f, coef = 1.f;
    for (int ph = 0; ph < octvs; ph++, curr_size /= 2.f, coef *= 2.f) {
        OctavePlan oc;
        oc.curr_size = curr_size;
        oc.coef = coef;
        oc.W = (int)(size_t)((uint32_t)w * curr_size);
        oc.H = (int)(size_t)((uint32_t)h * curr_size);
        if (oc.W < 2 || oc.H < 2) break;
        for (int i = 0; i < pyWidth; i++) {
            // Sift.cpp:152 (level 0) and :155 (levels >= 1)
            float s = (i == 0) ? prm.sigma0 * coef : prm.sigma0 * powf(std::sqrt(2.0), (size_t)i) * coef;
            float s_blur = s;  // the reference blurs level i-1 with the absolute sigma of level i (Sift.cpp:157)
            if (tb) {
                // quality mode: the same sigma ladder in octave pixels at every octave, reached by increments;
                // the resampled image counts as blurred by 0.5 source pixels
                s = prm.sigma0 * powf(std::sqrt(2.0), (size_t)i);
                const float below = i == 0 ? std::max(0.5f * curr_size, 0.1f) : oc.sigma.back();
                s_blur = std::sqrt(std::max(s * s - below * below, 0.01f));
            }
            std::vector<float> g = gau_matrix(s_blur, 0);
            oc.sigma.push_back(s);
            oc.tap_off.push_back((int)pl.table.size());
            oc.tap_len.push_back((int)g.size());
            pl.table.insert(pl.table.end(), g.begin(), g.end());
            while (pl.table.size() & 7) pl.table.push_back(0.f);
        }
        for (int li = 1; li <= prm.intvls; li++) {
            size_t r = curr_size * 3. * 1.5 * (li + 1);                                 // Sift.cpp:177
            float cs = prm.sigma0 * powf(std::sqrt(2.0), (size_t)li) * coef;            // Sift.cpp:183
            float ws_ = cs;                                                             // Sift.cpp:267
            if (tb) {  // quality mode: radius 3 x 1.5 sigma, weights of 1.5 sigma, sigma of the layer in octave pixels
                cs = prm.sigma0 * powf(std::sqrt(2.0), (size_t)li);
                ws_ = 1.5f * cs;
                r = (size_t)std::lround(3.0 * ws_);
            }
            std::vector<float> g = gau_matrix(ws_, r * 2 + 1);
            oc.radius.push_back((int)r);
            oc.curr_sigma.push_back(cs);
            oc.wts_off.push_back((int)pl.table.size());
            pl.table.insert(pl.table.end(), g.begin(), g.end());
            for (int z = 0; z < 4; z++) pl.table.push_back(0.f);  // the orientation kernel reads the weights four at a time
            while (pl.table.size() & 7) pl.table.push_back(0.f);
        }
        pl.oct.push_back(oc);
    }
    int rc = ensure(ctx, ctx->table, pl.table.size() * sizeof(float));
    if (rc) return rc;
    if (!hip_ok(ctx, hipMemcpyAsync(ctx->table.p, pl.table.data(), pl.table.size() * sizeof(float),
                                    hipMemcpyHostToDevice, ctx->stream), "upload tap table"))
        return SIFT_HIP_ERR_HIP;
    std::vector<LayerConsts> lc;
    for (const OctavePlan& oc : pl.oct)
        for (int li = 0; li < prm.intvls; li++)
            lc.push_back(LayerConsts{oc.radius[li], oc.wts_off[li], oc.curr_sigma[li],
                                     tb ? oc.curr_sigma[li] : (float)(int)oc.curr_sigma[li]});
    rc = ensure(ctx, ctx->layer_consts, std::max<size_t>(1, lc.size()) * sizeof(LayerConsts));
    if (rc) return rc;
    if (!lc.empty() &&
        !hip_ok(ctx, hipMemcpyAsync(ctx->layer_consts.p, lc.data(), lc.size() * sizeof(LayerConsts),
                                    hipMemcpyHostToDevice, ctx->stream), "upload layer constants"))
        return SIFT_HIP_ERR_HIP;
    // the uploads read pageable host memory: finish them before the vectors move
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "plan upload")) return SIFT_HIP_ERR_HIP;
    pl.valid = true;
    return SIFT_HIP_OK;
}

// Workspace of octave `oi` (created on first use; the streams belong to the context).
static int get_lane(sift_hip_ctx* ctx, int oi, OctWs** out) {
    while ((int)ctx->lanes.size() <= oi) {
        OctWs ws;
        ws.stream = ctx->stream;
        ctx->lanes.push_back(ws);
    }
    *out = &ctx->lanes[oi];
    return SIFT_HIP_OK;
}

// Gaussian levels of octave `oi` from the gray image on the device, on stream st.
// One octave's pyramid in steps: step -1 = workspace + resample from the frame, step i >= 0 =
// Gaussian level i (blurred from level i-1, Sift.cpp:145-160).  The levels of one octave must be
// enqueued in order on one stream; steps of different octaves may be interleaved by the host.
static int octave_pyramid_step(sift_hip_ctx* ctx, OctWs& ws, hipStream_t st, const float* d_gray, int oi, int step) {
    const Plan& pl = ctx->plan;
    const OctavePlan& oc = pl.oct[oi];
    const size_t PX = (size_t)oc.W * oc.H;
    const int pyWidth = pl.prm.intvls + 3;
    if (step < 0) {
        int rc;
        if ((rc = ensure(ctx, ws.base, PX * 4))) return rc;
        if ((rc = ensure(ctx, ws.tmp, PX * 4))) return rc;
        if ((rc = ensure(ctx, ws.gau_all, PX * 4 * pyWidth + 64))) return rc;  // (+64: the descriptor's 16-byte loads may run a few floats past the last row)
        launch_first(ctx, st, d_gray, pl.w, pl.h, oc.curr_size, (float*)ws.base.p, oc.W, oc.H);
        return SIFT_HIP_OK;
    }
    float* gau = (float*)ws.gau_all.p;
    const float* src = (step == 0) ? (const float*)ws.base.p : gau + (size_t)(step - 1) * PX;
    launch_blur(ctx, st, src, (float*)ws.tmp.p, gau + (size_t)step * PX, oc.W, oc.H,
                (const float*)ctx->table.p + oc.tap_off[step], oc.tap_len[step], pl.prm.blur_mode);
    return SIFT_HIP_OK;
}

static int run_octave_pyramid(sift_hip_ctx* ctx, OctWs& ws, hipStream_t st, const float* d_gray, int oi,
                              int skip_levels = 0) {
    const int pyWidth = ctx->plan.prm.intvls + 3;
    for (int step = -1; step < pyWidth - skip_levels; step++) {
        const int rc = octave_pyramid_step(ctx, ws, st, d_gray, oi, step);
        if (rc) return rc;
    }
    return SIFT_HIP_OK;
}

static int check_async(sift_hip_ctx* ctx, const char* what) {
    return hip_ok(ctx, hipGetLastError(), what) ? SIFT_HIP_OK : SIFT_HIP_ERR_HIP;
}

// The whole of Sift::pyramid for one device-resident gray frame; results stay on
// the device (per-octave feature buffers, ctx->gcounters).  No device-side
// synchronisation between streams anywhere: octave 0 runs on ctx->stream, the
// other octaves on the side streams, every kernel depends only on its
// predecessors in its own stream, and the host joins them (fetch_features).  A
// stream that waits on another stream's event parks a barrier packet in its
// hardware queue; streams share the few hardware queues of a process, so such a
// packet can sit in front of the very work that has to signal it -- the library
// therefore never enqueues one, whatever else the process runs (other contexts,
// torch, RCCL).
// frame_state: 0 = the frame is complete in device memory, 1 = it is produced by
// work already enqueued on ctx->stream (an upload, the gray conversion): the host
// then waits for that work before it feeds the side streams.
static int enqueue_extract(sift_hip_ctx* ctx, const float* d_gray, int w, int h, const sift_hip_params& prm,
                           int feat_cap, int cand_cap, bool want_hist, int frame_state) {
    ctx->last_feat_ptr.clear();  // the octaves' record arrays are about to be rewritten
    ctx->last_feat_cnt.clear();
    int rc = build_plan(ctx, w, h, prm);
    if (rc) return rc;
    const Plan& pl = ctx->plan;
    const int noct = (int)pl.oct.size();
    if (noct > G_MAX_OCT) return SIFT_HIP_ERR_LIMIT;
    // (page-locked host memory that the kernels write through the bus: once a stream has been joined its
    // counters are simply there, without a copy and a second wait)
    if ((rc = ensure_pinned(ctx, ctx->gcounters, G_SLOTS * sizeof(int)))) return rc;
    // all allocation first: ensure() may synchronise, which must not happen between launches
    for (int oi = 0; oi < noct; oi++) {
        OctWs* ws;
        if ((rc = get_lane(ctx, oi, &ws))) return rc;
        const OctavePlan& oc = pl.oct[oi];
        const size_t PX = (size_t)oc.W * oc.H;
        const int WW = (oc.W + 63) / 64;
        const int nrows = prm.intvls * oc.H;
        if ((rc = ensure(ctx, ws->base, PX * 4))) return rc;
        if ((rc = ensure(ctx, ws->tmp, PX * 4))) return rc;
        if ((rc = ensure(ctx, ws->gau_all, PX * 4 * (prm.intvls + 3) + 64))) return rc;  // (+64: the descriptor's 16-byte loads may run a few floats past the last row)
        if ((rc = ensure(ctx, ws->masks, (size_t)nrows * WW * 8))) return rc;
        if ((rc = ensure(ctx, ws->row_cnt, (size_t)nrows * 4))) return rc;
        if ((rc = ensure(ctx, ws->row_off, (size_t)nrows * 4))) return rc;
        if ((rc = ensure(ctx, ws->cand, (size_t)cand_cap * sizeof(Cand)))) return rc;
        if ((rc = ensure(ctx, ws->cand_mask, (size_t)cand_cap * sizeof(unsigned long long)))) return rc;
        if ((rc = ensure(ctx, ws->cand_mm, (size_t)cand_cap * 36 * sizeof(float)))) return rc;
        if ((rc = ensure(ctx, ws->cand_off, (size_t)cand_cap * sizeof(int)))) return rc;
        if ((rc = ensure(ctx, ws->counters, CNT_SLOTS * sizeof(int)))) return rc;
        if ((rc = ensure(ctx, ws->counters2, CNT_SLOTS * sizeof(int)))) return rc;
        if ((rc = ensure(ctx, ws->row_off2, (size_t)oc.H * 4))) return rc;
        if ((rc = ensure(ctx, ws->cand2, (size_t)cand_cap * sizeof(Cand)))) return rc;
        if ((rc = ensure(ctx, ws->feat, (size_t)std::max(1, feat_cap) * sizeof(sift_hip_feat)))) return rc;
        if (want_hist && (rc = ensure(ctx, ws->hist, (size_t)std::max(1, feat_cap) * 128 * sizeof(float)))) return rc;
        if (prm.subpixel) {
            if ((rc = ensure(ctx, ws->cand_sub, (size_t)cand_cap * sizeof(float4)))) return rc;
            if ((rc = ensure(ctx, ws->feat_sub, (size_t)std::max(1, feat_cap) * 3 * sizeof(float)))) return rc;
        }
    }
    const float thr = (float)((double)prm.dx_thr / 2.0);
    // With profiling on (or SIFT_HIP_SERIAL=1) everything runs on the main stream, so that
    // kernel durations are not inflated by another octave's kernels sharing the GPU.
    const bool serial = ctx->profiling || ctx->force_serial || noct < 2;
    // Where octave 3 goes (tune.o3_place): 0 = behind octave 2 on the second side stream, 1 = in front of
    // octave 1 on the first, 2 = behind octave 1, 3 = in front of octave 2, 4 = on the deep-octave stream.
    const int o3_place = (!serial && ctx->n_side >= 2 && noct > 3 && ctx->tune.interleave != 0 && !ctx->pipelined)
                             ? ctx->tune.o3_place : 0;
    auto stream_of = [&](int oi) -> hipStream_t {
        if (serial || oi == 0) return ctx->stream;
        if (oi == 1 || ctx->n_side < 2) return ctx->side_stream;
        if (oi == 3 && (o3_place == 1 || o3_place == 2)) return ctx->side_stream;
        if (oi == 3 && o3_place == 4) return ctx->side_stream3;
        return oi < 4 ? ctx->side_stream2 : ctx->side_stream3;
    };
    if (frame_state == 1 && !serial) hipEventRecord(ctx->ev_start, ctx->stream);
    ctx->side_busy = !serial;
    // An octave's launches stay in order on its stream; the host's order between octaves is chosen
    // below (level by level in turn for a single frame, octave by octave inside the batch pipeline).
    auto enqueue_part = [&](int oi, int part) -> int {  // part 0 = pyramid, 1 = detect ... describe
        OctWs& ws = ctx->lanes[oi];
        const hipStream_t st = stream_of(oi);
        ws.stream = st;
        const OctavePlan& oc = pl.oct[oi];
        const size_t PX = (size_t)oc.W * oc.H;
        // The last Gaussian level is only read next to top-layer extrema: unless the
        // sub-pixel stage (which reads it around moved points) is on or its kernel is very
        // long, it is not built as an image.
        const int top_K = oc.tap_len[prm.intvls + 2];
        // (a workgroup sweeps K+2 rows of K+2 inputs per candidate: beyond ~1000 taps, i.e. in
        // octaves of a few thousand pixels, building the level is cheaper)
        const bool lazy = !prm.subpixel && !ctx->full_top_level && top_K <= 1024;
        ws.lazy_top = lazy;
        int r = SIFT_HIP_OK;
        if (part == 0) {
            r = run_octave_pyramid(ctx, ws, st, d_gray, oi, lazy ? 1 : 0);
        } else {
            const LayerConsts* d_lc = (const LayerConsts*)ctx->layer_consts.p + (size_t)oi * prm.intvls;
            launch_detect(ctx, ws, oi, PX, oc.W, oc.H, prm.intvls, d_lc, thr, cand_cap,
                          lazy ? (const float*)ctx->table.p + oc.tap_off[prm.intvls + 2] : nullptr, top_K);
            // (quality mode: the contrast bar per interval, the rule the reference's header keeps commented out, matrix.hpp:181)
            if (prm.subpixel)
                launch_refine(ctx, ws, PX, oc.W, oc.H, prm.intvls, d_lc,
                              prm.textbook ? prm.contr_thr / (float)prm.intvls : prm.contr_thr);
            launch_orient(ctx, ws, PX, oc.W, d_lc, (const float*)ctx->table.p);
            launch_emit(ctx, ws, oi, oc.curr_size, d_lc, feat_cap, prm.subpixel != 0, prm.textbook != 0);
            int max_window = 1;
            for (int li = 0; li < prm.intvls; li++) {
                // descriptor radius of Sift.cpp:357 for this layer's integer-truncated sigma
                const float hw = 3.f * (prm.textbook ? oc.curr_sigma[li] : (float)(int)oc.curr_sigma[li]);
                const int R = (int)(hw * std::sqrt(2.f) * (4 + 1.f) * 0.5f + 0.5f);
                max_window = std::max(max_window, (2 * R + 1) * (2 * R + 1));
            }
            launch_descr(ctx, ws, PX, oc.W, oc.H, (const float*)ctx->table.p, feat_cap,
                         want_hist ? (float*)ws.hist.p : nullptr, max_window, prm.textbook != 0);
        }
        return r;
    };
    const bool interleave = ctx->tune.interleave != 0;
    // (not in the batch pipeline, where the previous frame's tail keeps the device busy meanwhile and
    // launching octave by octave measured 5 % faster)
    if (interleave && !ctx->pipelined && !serial && ctx->n_side >= 2 && noct > 2) {
        // Three streams: the pyramids of octaves 0-2 are enqueued level by level in turn, so that
        // all three streams have work within the first few launches instead of after octave 0's
        // whole pyramid (the host needs ~5 us per launch; the small octaves' long, few-pixel kernels
        // are what ends the frame, so they should start first, not 0.2 ms in).
        // the first octave of each stream: 0, 1, 2 and (frames with more than four octaves) 4
        int heads[4] = {0, 1, 2, 4};
        const int nhead = (noct > 4 || o3_place == 4) ? 4 : 3;
        if (o3_place == 1) heads[1] = 3;  // octave 3 leads the first side stream, octave 1 follows it
        if (o3_place == 3) heads[2] = 3;  // octave 3 leads the second side stream, octave 2 follows it
        if (o3_place == 4) heads[3] = 3;  // octave 3 leads the deep stream, octaves 4, 5, ... follow it
        int nlev[4];
        for (int k = 0; k < nhead; k++) {
            const int oi = heads[k];
            OctWs& ws = ctx->lanes[oi];
            ws.stream = stream_of(oi);
            const int top_K = pl.oct[oi].tap_len[prm.intvls + 2];
            ws.lazy_top = !prm.subpixel && !ctx->full_top_level && top_K <= 1024;
            nlev[k] = prm.intvls + 3 - (ws.lazy_top ? 1 : 0);
        }
        if ((rc = octave_pyramid_step(ctx, ctx->lanes[0], stream_of(0), d_gray, 0, -1))) return rc;
        if ((rc = octave_pyramid_step(ctx, ctx->lanes[0], stream_of(0), d_gray, 0, 0))) return rc;
        if (frame_state == 1 && !hip_ok(ctx, hipEventSynchronize(ctx->ev_start), "frame upload")) return SIFT_HIP_ERR_HIP;
        for (int k = nhead - 1; k >= 1; k--)
            if ((rc = octave_pyramid_step(ctx, ctx->lanes[heads[k]], stream_of(heads[k]), d_gray, heads[k], -1))) return rc;
        for (int lvl = 0; lvl < prm.intvls + 3; lvl++)
            for (int k = nhead - 1; k >= 0; k--)
                if (lvl < nlev[k] && !(k == 0 && lvl == 0) &&
                    (rc = octave_pyramid_step(ctx, ctx->lanes[heads[k]], stream_of(heads[k]), d_gray, heads[k], lvl)))
                    return rc;
        for (int k = nhead - 1; k >= 0; k--)
            if ((rc = enqueue_part(heads[k], 1))) return rc;
        // the followers (the octave that shares a stream with a leader; octaves 5, 6, ... on the deep stream) in turn
        int a = o3_place == 1 ? 1 : o3_place == 3 ? 2 : o3_place == 4 ? 4 : 3, b = o3_place == 4 ? 4 : 5;
        while (a < 4 || b < noct) {
            if (a < 4 && a < noct) {
                if ((rc = enqueue_part(a, 0))) return rc;
                if ((rc = enqueue_part(a, 1))) return rc;
            }
            a = 4;
            if (b < noct) {
                if ((rc = enqueue_part(b, 0))) return rc;
                if ((rc = enqueue_part(b, 1))) return rc;
                b++;
            }
        }
        return check_async(ctx, "extract launch");
    }
    if ((rc = enqueue_part(0, 0))) return rc;
    // the main stream has its first long kernels; now the frame must be there for the side streams
    if (frame_state == 1 && !serial && !hip_ok(ctx, hipEventSynchronize(ctx->ev_start), "frame upload"))
        return SIFT_HIP_ERR_HIP;
    if (noct > 1) {
        if ((rc = enqueue_part(1, 0))) return rc;
        if ((rc = enqueue_part(1, 1))) return rc;
    }
    // With a stream of their own the first of the small octaves goes in before the rest of octave
    // 0, which the main stream does not reach before its pyramid kernels (about a millisecond at
    // 4K) are done; the others follow it, so that the main stream never waits for the host.
    const bool small_first = !serial && ctx->n_side >= 2 && noct > 2;
    if (small_first) {
        if ((rc = enqueue_part(2, 0))) return rc;
        if ((rc = enqueue_part(2, 1))) return rc;
    }
    if ((rc = enqueue_part(0, 1))) return rc;
    for (int oi = small_first ? 3 : 2; oi < noct; oi++) {
        if ((rc = enqueue_part(oi, 0))) return rc;
        if ((rc = enqueue_part(oi, 1))) return rc;
    }
    return check_async(ctx, "extract launch");
}

// Waits (on the host) for the frame's work on the side streams, copies the counters
// back, then each octave's features (and histograms) to their place in the caller's
// array: octave-major, as the reference appends them.  *cand_overflow receives the
// largest candidate count that did not fit the table (0 if all did).  A caller that
// has already waited for both streams passes join_side = false and the stream the
// copies should use.
static int fetch_features(sift_hip_ctx* ctx, sift_hip_feat* out, int cap, int* n, float* hist, int* cand_overflow,
                          hipStream_t st = nullptr, bool join_side = true) {
    if (!st) st = ctx->stream;
    int early = 0;   // records of octave 0 whose copy is already under way
    int early1 = 0;  // ... and of octave 1 (only if all of octave 0's are)
    bool plain = false;
    if (join_side && ctx->side_busy) {
        // Octave 0 holds three quarters of the features and its stream is usually done while the
        // small octaves still drain: read its count and start its copy (on the copy stream) now.
        // (worth its extra counter read, about 10 us, only for frames with a long drain: 4 Mpixel and more)
        plain = !hist && ctx->plan.prm.subpixel == 0 && st == ctx->stream &&
                (size_t)ctx->plan.oct[0].W * ctx->plan.oct[0].H >= ((size_t)16 << 20);
        if (plain) {
            if (!hip_ok(ctx, hipStreamSynchronize(st), "extract (octave 0)")) return SIFT_HIP_ERR_HIP;
            std::memcpy(ctx->h_counters, ctx->gcounters.p, G_SLOTS * sizeof(int));
            int found0 = ctx->h_counters[G_OCT_CAND];
            if (ctx->lanes[0].lazy_top) found0 = std::max(found0, ctx->h_counters[G_OCT_CAND_TOP]);
            if (found0 <= ctx->cand_cap) {
                early = std::max(0, std::min(ctx->h_counters[G_OCT_FEATS], cap));
                if (early > 0 && !hip_ok(ctx, hipMemcpyAsync(out, ctx->lanes[0].feat.p, (size_t)early * sizeof(sift_hip_feat),
                                                             hipMemcpyDeviceToHost, ctx->copy_stream), "read features"))
                    return SIFT_HIP_ERR_HIP;
            }
        }
        if (!hip_ok(ctx, hipStreamSynchronize(ctx->side_stream), "extract (side stream)")) return SIFT_HIP_ERR_HIP;
        // Octave 1 likewise (its stream is joined now, the deeper octaves' stream usually not yet): its records go
        // behind octave 0's, whose count is final.
        if (plain && early > 0 && early == ctx->h_counters[G_OCT_FEATS] && ctx->plan.oct.size() > 2) {
            const volatile int* gc = (const volatile int*)ctx->gcounters.p;
            int found1 = gc[G_OCT_CAND + 1];
            if (ctx->lanes[1].lazy_top) found1 = std::max(found1, (int)gc[G_OCT_CAND_TOP + 1]);
            const int n1 = gc[G_OCT_FEATS + 1];
            if (found1 <= ctx->cand_cap && n1 > 0 && n1 <= cap - early) {
                if (!hip_ok(ctx, hipMemcpyAsync(out + early, ctx->lanes[1].feat.p, (size_t)n1 * sizeof(sift_hip_feat),
                                                hipMemcpyDeviceToHost, ctx->copy_stream), "read features"))
                    return SIFT_HIP_ERR_HIP;
                early1 = n1;
            }
        }
        if (!hip_ok(ctx, hipStreamSynchronize(ctx->side_stream2), "extract (second side stream)") ||
            !hip_ok(ctx, hipStreamSynchronize(ctx->side_stream3), "extract (third side stream)"))
            return SIFT_HIP_ERR_HIP;
    }
    if (!hip_ok(ctx, hipStreamSynchronize(st), "extract")) return SIFT_HIP_ERR_HIP;
    std::memcpy(ctx->h_counters, ctx->gcounters.p, G_SLOTS * sizeof(int));
    *cand_overflow = 0;
    for (int oi = 0; oi < (int)ctx->plan.oct.size(); oi++) {
        int found = ctx->h_counters[G_OCT_CAND + oi];
        if (ctx->lanes[oi].lazy_top) found = std::max(found, ctx->h_counters[G_OCT_CAND_TOP + oi]);
        if (found > ctx->cand_cap) *cand_overflow = std::max(*cand_overflow, found);
    }
    if (*cand_overflow) {
        if (early > 0 || early1 > 0) hipStreamSynchronize(ctx->copy_stream);  // the caller's array is rewritten by the next attempt
        return SIFT_HIP_OK;
    }
    const int noct = (int)ctx->plan.oct.size();
    const bool with_sub = ctx->plan.prm.subpixel != 0;
    ctx->h_sub.clear();
    if (with_sub) {
        long long want = 0;
        for (int oi = 0; oi < noct; oi++) want += ctx->h_counters[G_OCT_FEATS + oi];
        ctx->h_sub.resize((size_t)std::min<long long>(want, cap) * 3);
    }
    long long total = 0;
    int copied = 0;
    ctx->last_feat_ptr.assign((size_t)noct, nullptr);
    ctx->last_feat_cnt.assign((size_t)noct, 0);
    for (int oi = 0; oi < noct; oi++) {
        const int cnt = ctx->h_counters[G_OCT_FEATS + oi];
        const int m = std::max(0, std::min(cnt, cap - copied));  // the octave's buffer holds at most cap
        ctx->last_feat_ptr[(size_t)oi] = (const sift_hip_feat*)ctx->lanes[oi].feat.p;
        ctx->last_feat_cnt[(size_t)oi] = m;
        if (m > 0) {
            if (!(oi == 0 && early == m) && !(oi == 1 && early1 == m && early1 > 0) &&
                !hip_ok(ctx, hipMemcpyAsync(out + copied, ctx->lanes[oi].feat.p, (size_t)m * sizeof(sift_hip_feat),
                                            hipMemcpyDeviceToHost, st), "read features"))
                return SIFT_HIP_ERR_HIP;
            if (hist && !hip_ok(ctx, hipMemcpyAsync(hist + (size_t)copied * 128, ctx->lanes[oi].hist.p,
                                                    (size_t)m * 128 * sizeof(float), hipMemcpyDeviceToHost,
                                                    st), "read histograms"))
                return SIFT_HIP_ERR_HIP;
            if (with_sub && !hip_ok(ctx, hipMemcpyAsync(ctx->h_sub.data() + (size_t)copied * 3, ctx->lanes[oi].feat_sub.p,
                                                        (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost,
                                                        st), "read offsets"))
                return SIFT_HIP_ERR_HIP;
            copied += m;
        }
        total += cnt;
    }
    *n = (int)std::min<long long>(total, INT32_MAX);
    if (copied > 0 && !hip_ok(ctx, hipStreamSynchronize(st), "read features")) return SIFT_HIP_ERR_HIP;
    if ((early > 0 || early1 > 0) && !hip_ok(ctx, hipStreamSynchronize(ctx->copy_stream), "read features (octaves 0, 1)")) return SIFT_HIP_ERR_HIP;
    return total > cap ? SIFT_HIP_ERR_NOSPC : SIFT_HIP_OK;
}

static int checked_params(const sift_hip_params* params, int w, int h, sift_hip_params* prm) {
    if (params) *prm = *params; else sift_hip_default_params(prm);
    if (prm->intvls < 1 || prm->intvls > 8 || prm->octaves < 0 || prm->octaves > 24 || !(prm->sigma0 > 0.f))
        return SIFT_HIP_ERR_ARG;
    if (prm->subpixel != 0 && prm->subpixel != 1) return SIFT_HIP_ERR_ARG;
    if (prm->textbook != 0 && prm->textbook != 1) return SIFT_HIP_ERR_ARG;
    if (prm->textbook) prm->subpixel = 1;  // the quality mode refines every keypoint
    // octave 0 is the image doubled in both directions; the kernels index one level with 32 bits
    if ((long long)w * h * 4 >= (1ll << 31)) return SIFT_HIP_ERR_LIMIT;
    return SIFT_HIP_OK;
}

static int extract_dev_impl(sift_hip_ctx* ctx, const float* d_gray, int w, int h, const sift_hip_params* params,
                            sift_hip_feat* out, int cap, int* n, float* hist, int frame_state = 0) {
    sift_hip_params prm;
    int rc = checked_params(params, w, h, &prm);
    if (rc) return rc;
    int cand_cap = std::max(ctx->cand_cap, std::max(cap, 1 << 16));
    for (int attempt = 0; attempt < 4; attempt++) {
        ctx->cand_cap = cand_cap;
        rc = enqueue_extract(ctx, d_gray, w, h, prm, cap, cand_cap, hist != nullptr, frame_state);
        if (rc) return rc;
        int overflow = 0;
        rc = fetch_features(ctx, out, cap, n, hist, &overflow);
        if (!overflow) return rc;
        cand_cap = overflow + overflow / 4;  // the candidate table was too small: grow and redo
    }
    return SIFT_HIP_ERR_LIMIT;
}

// The two output sets (feature buffers + counters) trade places.
static void swap_output_sets(sift_hip_ctx* ctx) {
    std::swap(ctx->gcounters, ctx->gcounters_alt);
    std::swap(ctx->h_counters, ctx->h_counters_alt);
    for (OctWs& ws : ctx->lanes) {
        std::swap(ws.feat, ws.feat_alt);
        std::swap(ws.feat_sub, ws.feat_sub_alt);
    }
}

// A batch of equal-sized frames, software-pipelined over the two output sets:
// frame f's kernels are enqueued before frame f-1's counters and features are
// copied back (on the copy stream), so the device never waits for the host
// between frames.  frame_ptr(f) returns the device address of frame f once the
// frame is complete there (it waits on the host if it has to); fetched(f) is called
// when frame f's kernels have finished and its features are on the host, i.e. when
// whatever held frame f may be reused.  If a frame overflows the candidate table
// the rest of the batch falls back to the one-frame-at-a-time path, which grows it;
// reload(f) returns frame f's device address for that path (the pipelined uploads
// may have reused its slot by then).
template <typename FramePtr, typename Fetched, typename Reload>
static int extract_batch_pipelined(sift_hip_ctx* ctx, FramePtr frame_ptr, Fetched fetched, Reload reload, int w, int h,
                                   int nframes, const sift_hip_params* params, sift_hip_feat* out, int cap, int* n) {
    sift_hip_params prm;
    int rc = checked_params(params, w, h, &prm);
    if (rc) return rc;
    int worst = SIFT_HIP_OK;
    auto plain_from = [&](int f0) -> int {
        for (int f = f0; f < nframes; f++) {
            n[f] = 0;
            const float* d = reload(f);
            if (!d) return SIFT_HIP_ERR_HIP;
            int r = extract_dev_impl(ctx, d, w, h, params, out + (size_t)f * cap, cap, &n[f], nullptr);
            if (r == SIFT_HIP_ERR_NOSPC) worst = r;
            else if (r) return r;
        }
        return worst;
    };
    if (ctx->profiling || nframes < 2) return plain_from(0);
    const int cand_cap = std::max(ctx->cand_cap, std::max(cap, 1 << 16));
    ctx->cand_cap = cand_cap;
    for (int f = 0; f < nframes; f++) {
        n[f] = 0;
        const float* d = frame_ptr(f);
        swap_output_sets(ctx);  // frame f writes the set that frame f-2 has released
        ctx->pipelined = true;
        rc = enqueue_extract(ctx, d, w, h, prm, cap, cand_cap, false, 0);
        ctx->pipelined = false;
        if (rc) {
            join_all(ctx);
            return rc;
        }
        hipEventRecord(ctx->ev_frame[f & 1], ctx->stream);
        if (ctx->side_busy) {
            hipEventRecord(ctx->ev_side[f & 1], ctx->side_stream);
            hipEventRecord(ctx->ev_side2[f & 1], ctx->side_stream2);
            hipEventRecord(ctx->ev_side3[f & 1], ctx->side_stream3);
        }
        if (f >= 1) {
            swap_output_sets(ctx);  // look at frame f-1's set
            // The host waits for frame f-1 on both streams (frame f keeps the device busy
            // meanwhile), then copies on the copy stream: no stream ever waits for another.
            if (!hip_ok(ctx, hipEventSynchronize(ctx->ev_frame[(f - 1) & 1]), "batch frame") ||
                (ctx->side_busy && (!hip_ok(ctx, hipEventSynchronize(ctx->ev_side[(f - 1) & 1]), "batch frame (side)") ||
                                    !hip_ok(ctx, hipEventSynchronize(ctx->ev_side2[(f - 1) & 1]), "batch frame (side 2)") ||
                                    !hip_ok(ctx, hipEventSynchronize(ctx->ev_side3[(f - 1) & 1]), "batch frame (side 3)"))))
                return SIFT_HIP_ERR_HIP;
            int overflow = 0;
            rc = fetch_features(ctx, out + (size_t)(f - 1) * cap, cap, &n[f - 1], nullptr, &overflow, ctx->copy_stream,
                                false);
            swap_output_sets(ctx);
            fetched(f - 1);
            if (overflow) {
                join_all(ctx);
                return plain_from(f - 1);
            }
            if (rc == SIFT_HIP_ERR_NOSPC) worst = rc;
            else if (rc) {
                join_all(ctx);
                return rc;
            }
        }
    }
    int overflow = 0;
    rc = fetch_features(ctx, out + (size_t)(nframes - 1) * cap, cap, &n[nframes - 1], nullptr, &overflow);
    if (overflow) return plain_from(nframes - 1);
    if (rc == SIFT_HIP_ERR_NOSPC) worst = rc;
    else if (rc) return rc;
    return worst;
}

static bool bad_ctx(sift_hip_ctx* ctx) {
    if (!ctx) return true;
    return hipSetDevice(ctx->device) != hipSuccess;
}

}  // namespace sifthip

extern "C" {

int sift_hip_create(sift_hip_ctx** out, int device) {
    if (!out) return SIFT_HIP_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        (void)hipGetLastError();
        return SIFT_HIP_ERR_NODEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SIFT_HIP_ERR_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SIFT_HIP_ERR_NODEVICE;  // kernels are gfx950-only
    if (hipSetDevice(device) != hipSuccess) return SIFT_HIP_ERR_NODEVICE;
    sift_hip_ctx* ctx = new (std::nothrow) sift_hip_ctx();
    if (!ctx) return SIFT_HIP_ERR_NOMEM;
    ctx->device = device;
    // SIFT_HIP_SERIAL=1: all octaves on one stream (clean per-kernel durations under a tracer)
    if (const char* e = std::getenv("SIFT_HIP_SERIAL")) ctx->force_serial = (e[0] == '1');
    // SIFT_HIP_TIMELINE=1: sift_hip_profile_read prints when each launch group began and ended on
    // which stream (events only, the streams stay concurrent: what a tracer's host overhead hides)
    if (const char* e = std::getenv("SIFT_HIP_TIMELINE")) ctx->timeline = (e[0] == '1');
    if (const char* e = std::getenv("SIFT_HIP_FULL_TOP")) ctx->full_top_level = (e[0] == '1');
    if (const char* e = std::getenv("SIFT_HIP_SIDE_PRIO")) ctx->side_priority = (e[0] == '0') ? 0 : 1;
    if (const char* e = std::getenv("SIFT_HIP_NSIDE")) ctx->n_side = (e[0] == '1') ? 1 : 2;
    // launch-shape switches: looked up here once, never on a launch path
    for (const char* key : {"blur_seg", "blur_stream", "blur_fused", "blur_sweep", "match_nb", "interleave", "o3_place"}) {
        std::string name = key == std::string("interleave") ? "SIFT_HIP_INTERLEAVE"
                         : key == std::string("match_nb")   ? "SIFT_MATCH_NB"
                                                            : std::string("SIFT_") + key;
        for (char& c : name) c = (char)std::toupper((unsigned char)c);
        if (const char* e = std::getenv(name.c_str())) sift_hip_set_tuning(ctx, key, std::atoi(e));
    }
    // the small octaves' stream gets the more urgent priority: their kernels are short and
    // latency-bound and would otherwise queue behind octave 0's large grids
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // prio_hi = most urgent
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, ctx->side_priority == 0 ? 0 : prio_hi) !=
            hipSuccess ||
        hipStreamCreateWithPriority(&ctx->side_stream2, hipStreamNonBlocking, ctx->side_priority == 0 ? 0 : prio_hi) !=
            hipSuccess ||
        hipStreamCreateWithPriority(&ctx->side_stream3, hipStreamNonBlocking, ctx->side_priority == 0 ? 0 : prio_hi) !=
            hipSuccess ||
        hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_start, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_frame[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_frame[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_side[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_side[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_side2[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_side2[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_side3[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_side3[1], hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_counters, G_SLOTS * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_counters_alt, G_SLOTS * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        sift_hip_destroy(ctx);
        return SIFT_HIP_ERR_HIP;
    }
    if (match_scratch_init(ctx) != SIFT_HIP_OK) {  // the device is current: once per device
        sift_hip_destroy(ctx);
        return SIFT_HIP_ERR_HIP;
    }
    *out = ctx;
    return SIFT_HIP_OK;
}

int sift_hip_destroy(sift_hip_ctx* ctx) {
    if (!ctx) return SIFT_HIP_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->side_stream) hipStreamSynchronize(ctx->side_stream);
    if (ctx->side_stream2) hipStreamSynchronize(ctx->side_stream2);
    if (ctx->side_stream3) hipStreamSynchronize(ctx->side_stream3);
    if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
    for (OctWs& ws : ctx->lanes) {
        DevBuf* wb[] = {&ws.base, &ws.tmp, &ws.gau_all, &ws.masks, &ws.row_cnt, &ws.row_off, &ws.cand, &ws.cand_mask,
                        &ws.cand_mm, &ws.cand_off, &ws.counters, &ws.feat, &ws.hist, &ws.cand_sub, &ws.feat_sub, &ws.feat_alt, &ws.feat_sub_alt, &ws.row_off2, &ws.cand2, &ws.counters2};
        for (DevBuf* b : wb)
            if (b->p) hipFree(b->p);
    }
    DevBuf* bufs[] = {&ctx->in_img, &ctx->in_rgb, &ctx->tmp, &ctx->table, &ctx->layer_consts,
                      &ctx->gathered, &ctx->gather_in, &ctx->gather_pad, &ctx->qi8, &ctx->ri8, &ctx->qnorm, &ctx->rconst, &ctx->best, &ctx->d1, &ctx->d2,
                      &ctx->io_a, &ctx->io_b, &ctx->frame2, &ctx->dog_all,
                      &ctx->ransac.blk, &ctx->ransac.pairs, &ctx->ransac.src, &ctx->ransac.flags, &ctx->ransac.samples,
                      &ctx->ransac.models, &ctx->ransac.counts, &ctx->ransac.qf, &ctx->ransac.rf, &ctx->ransac.best,
                      &ctx->warp_src, &ctx->warp_dst, &ctx->warp_edges, &ctx->blend_ws, &ctx->blend_l, &ctx->blend_r,
                      &ctx->pano_src, &ctx->pano_canvas, &ctx->pano_f32};
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    if (ctx->ev_start) hipEventDestroy(ctx->ev_start);
    if (ctx->h_feat.p) hipHostFree(ctx->h_feat.p);
    if (ctx->h_hist.p) hipHostFree(ctx->h_hist.p);
    if (ctx->gcounters.p) hipHostFree(ctx->gcounters.p);
    if (ctx->gcounters_alt.p) hipHostFree(ctx->gcounters_alt.p);
    if (ctx->h_counters) hipHostFree(ctx->h_counters);
    if (ctx->h_counters_alt) hipHostFree(ctx->h_counters_alt);
    for (hipEvent_t e : ctx->ev_frame) if (e) hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_side) if (e) hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_side2) if (e) hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_side3) if (e) hipEventDestroy(e);
    for (auto& e : ctx->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto& e : ctx->free_events) hipEventDestroy(e);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    if (ctx->side_stream) hipStreamDestroy(ctx->side_stream);
    if (ctx->side_stream2) hipStreamDestroy(ctx->side_stream2);
    if (ctx->side_stream3) hipStreamDestroy(ctx->side_stream3);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    delete ctx;
    return SIFT_HIP_OK;
}

void sift_hip_default_params(sift_hip_params* p) {
    if (!p) return;
    p->octaves = 0;
    p->intvls = 3;       // SIFT_INTVLS
    p->sigma0 = 1.6f;    // SIFT_GauSigma
    p->dx_thr = 0.03f;   // SIFT_Dx
    p->subpixel = 0;
    p->blur_mode = 0;
    p->contr_thr = 0.05f;  // SIFT_CONTR_THR
    p->textbook = 0;
}

const char* sift_hip_strerror(int s) {
    switch (s) {
        case SIFT_HIP_OK: return "ok";
        case SIFT_HIP_ERR_ARG: return "invalid argument";
        case SIFT_HIP_ERR_NOSPC: return "output capacity too small";
        case SIFT_HIP_ERR_HIP: return "HIP runtime error";
        case SIFT_HIP_ERR_NOMEM: return "out of device memory";
        case SIFT_HIP_ERR_SAME: return "input and output are the same buffer";
        case SIFT_HIP_ERR_NODEVICE: return "no usable gfx950 device";
        case SIFT_HIP_ERR_LIMIT: return "internal limit exceeded";
        default: return "unknown status";
    }
}

int sift_hip_set_tuning(sift_hip_ctx* ctx, const char* key, int value) {
    if (!ctx || !key) return SIFT_HIP_ERR_ARG;
    const std::string k(key);
    Tuning& t = ctx->tune;
    if (k == "blur_seg") t.blur_seg = std::max(0, value);
    else if (k == "blur_stream") t.blur_stream = value < 0 ? -1 : (value != 0);
    else if (k == "blur_fused") t.blur_fused = value < 0 ? -1 : value;
    else if (k == "blur_sweep") t.blur_sweep = value < 0 ? -1 : value;
    else if (k == "match_nb") { if (value != 0 && value != 2 && value != 4) return SIFT_HIP_ERR_ARG; t.match_nb = value; }
    else if (k == "interleave") t.interleave = value != 0;
    else if (k == "o3_place") { if (value < 0 || value > 4) return SIFT_HIP_ERR_ARG; t.o3_place = value; }
    else return SIFT_HIP_ERR_ARG;
    return SIFT_HIP_OK;
}

const char* sift_hip_last_error(const sift_hip_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
void* sift_hip_stream(sift_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int sift_hip_synchronize(sift_hip_ctx* ctx) {
    if (bad_ctx(ctx)) return SIFT_HIP_ERR_ARG;
    return hip_ok(ctx, hipStreamSynchronize(ctx->stream), "synchronize") ? SIFT_HIP_OK : SIFT_HIP_ERR_HIP;
}

int sift_hip_host_alloc(sift_hip_ctx* ctx, void** ptr, size_t bytes) {
    if (bad_ctx(ctx) || !ptr || bytes == 0) return SIFT_HIP_ERR_ARG;
    *ptr = nullptr;
    if (!hip_ok(ctx, hipSetDevice(ctx->device), "set device")) return SIFT_HIP_ERR_HIP;
    if (hipHostMalloc(ptr, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return SIFT_HIP_ERR_NOMEM;
    }
    return SIFT_HIP_OK;
}

int sift_hip_host_free(sift_hip_ctx* ctx, void* ptr) {
    (void)ctx;  // the allocation does not belong to a context
    if (!ptr) return SIFT_HIP_OK;
    return hipHostFree(ptr) == hipSuccess ? SIFT_HIP_OK : SIFT_HIP_ERR_HIP;
}

int sift_hip_pyramid_octaves(int w, int h) {
    if (w <= 0 || h <= 0) return SIFT_HIP_ERR_ARG;
    return pyramid_octaves((size_t)w, (size_t)h);
}

int sift_hip_gau_matrix(float p, int mat_len, float* taps, int cap) {
    if (!taps || mat_len < 0 || !(p > 0.f)) return SIFT_HIP_ERR_ARG;
    std::vector<float> g = gau_matrix(p, (size_t)mat_len);
    if ((int)g.size() > cap) return -(int)g.size();
    std::memcpy(taps, g.data(), g.size() * sizeof(float));
    return (int)g.size();
}

int sift_hip_extract_dev(sift_hip_ctx* ctx, const float* d_gray, int w, int h, const sift_hip_params* params,
                         sift_hip_feat* out, int cap, int* n, float* hist) {
    if (bad_ctx(ctx) || !d_gray || w < 2 || h < 2 || !n || cap < 0 || (cap > 0 && !out)) return SIFT_HIP_ERR_ARG;
    *n = 0;
    // the frame may come from work the caller enqueued on the context's stream
    const int frame_state = hipStreamQuery(ctx->stream) == hipSuccess ? 0 : 1;
    (void)hipGetLastError();  // hipErrorNotReady is not a failure
    return extract_dev_impl(ctx, d_gray, w, h, params, out, cap, n, hist, frame_state);
}

int sift_hip_extract(sift_hip_ctx* ctx, const float* img, int w, int h, int channels, const sift_hip_params* params,
                     sift_hip_feat* out, int cap, int* n, float* hist) {
    if (bad_ctx(ctx) || !img || w < 2 || h < 2 || !n || cap < 0 || (cap > 0 && !out)) return SIFT_HIP_ERR_ARG;
    if (channels != 1 && channels != 3) return SIFT_HIP_ERR_ARG;
    *n = 0;
    const size_t px = (size_t)w * h;
    int rc;
    {
        sift_hip_params checked;  // reject bad parameters and sizes before touching the image
        if ((rc = checked_params(params, w, h, &checked))) return rc;
    }
    if ((rc = ensure(ctx, ctx->in_img, px * 4))) return rc;
    if (channels == 3) {
        if ((rc = ensure(ctx, ctx->in_rgb, px * 12))) return rc;
        if (!hip_ok(ctx, hipMemcpyAsync(ctx->in_rgb.p, img, px * 12, hipMemcpyHostToDevice, ctx->stream), "upload"))
            return SIFT_HIP_ERR_HIP;
        launch_gray(ctx, ctx->stream, (const float*)ctx->in_rgb.p, (float*)ctx->in_img.p, px);
    } else {
        if (!hip_ok(ctx, hipMemcpyAsync(ctx->in_img.p, img, px * 4, hipMemcpyHostToDevice, ctx->stream), "upload"))
            return SIFT_HIP_ERR_HIP;
    }
    return extract_dev_impl(ctx, (const float*)ctx->in_img.p, w, h, params, out, cap, n, hist, /*frame_state=*/1);
}

int sift_hip_extract_batch_dev(sift_hip_ctx* ctx, const float* d_frames, int w, int h, int nframes,
                               const sift_hip_params* params, sift_hip_feat* out, int cap, int* n) {
    if (bad_ctx(ctx) || !d_frames || w < 2 || h < 2 || nframes < 0 || !n || cap < 0 || (cap > 0 && !out))
        return SIFT_HIP_ERR_ARG;
    const size_t px = (size_t)w * h;
    // frames produced by work the caller enqueued on the context's stream: wait for it once
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "batch frames")) return SIFT_HIP_ERR_HIP;
    return extract_batch_pipelined(
        ctx, [&](int f) { return d_frames + (size_t)f * px; }, [](int) {},
        [&](int f) { return d_frames + (size_t)f * px; }, w, h, nframes, params, out, cap, n);
}

int sift_hip_extract_batch(sift_hip_ctx* ctx, const float* frames, int w, int h, int nframes,
                           const sift_hip_params* params, sift_hip_feat* out, int cap, int* n) {
    if (bad_ctx(ctx) || !frames || w < 2 || h < 2 || nframes < 0 || !n || cap < 0 || (cap > 0 && !out))
        return SIFT_HIP_ERR_ARG;
    const size_t px = (size_t)w * h;
    int rc;
    {
        sift_hip_params checked;
        if ((rc = checked_params(params, w, h, &checked))) return rc;
    }
    if ((rc = ensure(ctx, ctx->in_img, px * 4))) return rc;
    if ((rc = ensure(ctx, ctx->frame2, px * 4))) return rc;
    float* slot[2] = {(float*)ctx->in_img.p, (float*)ctx->frame2.p};
    hipEvent_t up[2];
    for (int i = 0; i < 2; i++) hipEventCreateWithFlags(&up[i], hipEventDisableTiming);
    // Frame f lives in slot f & 1.  Uploads run on the copy stream while earlier frames'
    // kernels run; all ordering between the streams is done by the host: it waits for
    // frame f's upload before enqueueing frame f's kernels, and starts the upload of
    // frame f+2 into frame f's slot only when frame f's kernels have finished.
    auto upload = [&](int f) {
        hipMemcpyAsync(slot[f & 1], frames + (size_t)f * px, px * 4, hipMemcpyHostToDevice, ctx->copy_stream);
        hipEventRecord(up[f & 1], ctx->copy_stream);
    };
    for (int f = 0; f < std::min(nframes, 2); f++) upload(f);
    auto frame_ptr = [&](int f) -> const float* {
        hipEventSynchronize(up[f & 1]);
        return slot[f & 1];
    };
    auto fetched = [&](int f) {
        if (f + 2 < nframes) upload(f + 2);
    };
    auto reload = [&](int f) -> const float* {
        hipStreamSynchronize(ctx->copy_stream);
        hipStreamSynchronize(ctx->stream);
        hipStreamSynchronize(ctx->side_stream);
        hipStreamSynchronize(ctx->side_stream2);
        hipStreamSynchronize(ctx->side_stream3);
        if (!hip_ok(ctx, hipMemcpy(slot[0], frames + (size_t)f * px, px * 4, hipMemcpyHostToDevice), "upload frame"))
            return nullptr;
        return slot[0];
    };
    const int worst = extract_batch_pipelined(ctx, frame_ptr, fetched, reload, w, h, nframes, params, out, cap, n);
    hipStreamSynchronize(ctx->copy_stream);
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->side_stream);
    hipStreamSynchronize(ctx->side_stream2);
    hipStreamSynchronize(ctx->side_stream3);
    for (int i = 0; i < 2; i++) hipEventDestroy(up[i]);
    return worst;
}

int sift_hip_gau_blur(sift_hip_ctx* ctx, const float* in, float* out, int w, int h, float p, int mat_len,
                      int blur_mode) {
    if (bad_ctx(ctx) || !in || !out || w < 1 || h < 1 || !(p > 0.f) || mat_len < 0) return SIFT_HIP_ERR_ARG;
    if (in == out) return SIFT_HIP_ERR_SAME;
    const size_t px = (size_t)w * h;
    std::vector<float> g = gau_matrix(p, (size_t)mat_len);
    int rc;
    if ((rc = ensure(ctx, ctx->io_a, px * 4))) return rc;
    if ((rc = ensure(ctx, ctx->io_b, px * 4))) return rc;
    if ((rc = ensure(ctx, ctx->tmp, px * 4))) return rc;
    if ((rc = ensure(ctx, ctx->best, g.size() * 4 + 64))) return rc;  // scratch for this call's taps
    hipMemcpyAsync(ctx->best.p, g.data(), g.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    hipMemcpyAsync(ctx->io_a.p, in, px * 4, hipMemcpyHostToDevice, ctx->stream);
    launch_blur(ctx, ctx->stream, (const float*)ctx->io_a.p, (float*)ctx->tmp.p, (float*)ctx->io_b.p, w, h,
                (const float*)ctx->best.p, (int)g.size(), blur_mode);
    hipMemcpyAsync(out, ctx->io_b.p, px * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "gau_blur")) return SIFT_HIP_ERR_HIP;
    return check_async(ctx, "gau_blur");
}

int sift_hip_first(sift_hip_ctx* ctx, const float* in, int w, int h, float ratio, float* out) {
    if (bad_ctx(ctx) || !in || !out || w < 1 || h < 1 || !(ratio > 0.f)) return SIFT_HIP_ERR_ARG;
    const int nh = static_cast<int>(std::floor((size_t)h * ratio));
    const int nw = static_cast<int>(std::floor((size_t)w * ratio));
    if (nw < 1 || nh < 1) return SIFT_HIP_ERR_ARG;
    int rc;
    if ((rc = ensure(ctx, ctx->io_a, (size_t)w * h * 4))) return rc;
    if ((rc = ensure(ctx, ctx->io_b, (size_t)nw * nh * 4))) return rc;
    hipMemcpyAsync(ctx->io_a.p, in, (size_t)w * h * 4, hipMemcpyHostToDevice, ctx->stream);
    launch_first(ctx, ctx->stream, (const float*)ctx->io_a.p, w, h, ratio, (float*)ctx->io_b.p, nw, nh);
    hipMemcpyAsync(out, ctx->io_b.p, (size_t)nw * nh * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "first")) return SIFT_HIP_ERR_HIP;
    return check_async(ctx, "first");
}

int sift_hip_convert_gray(sift_hip_ctx* ctx, const float* rgb, int n, float* gray) {
    if (bad_ctx(ctx) || !rgb || !gray || n < 1) return SIFT_HIP_ERR_ARG;
    int rc;
    if ((rc = ensure(ctx, ctx->io_a, (size_t)n * 12))) return rc;
    if ((rc = ensure(ctx, ctx->io_b, (size_t)n * 4))) return rc;
    hipMemcpyAsync(ctx->io_a.p, rgb, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream);
    launch_gray(ctx, ctx->stream, (const float*)ctx->io_a.p, (float*)ctx->io_b.p, (size_t)n);
    hipMemcpyAsync(gray, ctx->io_b.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "convert_gray")) return SIFT_HIP_ERR_HIP;
    return check_async(ctx, "convert_gray");
}

int sift_hip_octave_levels(sift_hip_ctx* ctx, const float* gray, int w, int h, const sift_hip_params* params,
                           int octave, float* gau, float* dog, int* ow, int* oh) {
    if (bad_ctx(ctx) || !gray || w < 2 || h < 2 || octave < 0 || !ow || !oh) return SIFT_HIP_ERR_ARG;
    sift_hip_params prm;
    {
        const int vrc = checked_params(params, w, h, &prm);
        if (vrc) return vrc;
    }
    if (prm.octaves > 0 && prm.octaves <= octave) prm.octaves = octave + 1;
    int rc = build_plan(ctx, w, h, prm);
    if (rc) return rc;
    if (octave >= (int)ctx->plan.oct.size()) return SIFT_HIP_ERR_ARG;
    const OctavePlan& oc = ctx->plan.oct[octave];
    const size_t PX = (size_t)oc.W * oc.H;
    if ((rc = ensure(ctx, ctx->in_img, (size_t)w * h * 4))) return rc;
    hipMemcpyAsync(ctx->in_img.p, gray, (size_t)w * h * 4, hipMemcpyHostToDevice, ctx->stream);
    OctWs* ws;
    if ((rc = get_lane(ctx, 0, &ws))) return rc;
    if ((rc = run_octave_pyramid(ctx, *ws, ctx->stream, (const float*)ctx->in_img.p, octave))) return rc;
    const int pyWidth = prm.intvls + 3;
    if (gau) hipMemcpyAsync(gau, ws->gau_all.p, PX * 4 * pyWidth, hipMemcpyDeviceToHost, ctx->stream);
    if (dog) {
        if ((rc = ensure(ctx, ctx->dog_all, PX * 4 * (pyWidth - 1)))) return rc;
        launch_dog(ctx, ctx->stream, (const float*)ws->gau_all.p, (float*)ctx->dog_all.p, PX, pyWidth);
        hipMemcpyAsync(dog, ctx->dog_all.p, PX * 4 * (pyWidth - 1), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "octave_levels")) return SIFT_HIP_ERR_HIP;
    *ow = oc.W;
    *oh = oc.H;
    return check_async(ctx, "octave_levels");
}

int sift_hip_match_u8_dev(sift_hip_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, float ratio,
                          int ratio_form, int32_t* d_best, int32_t* d_d1, int32_t* d_d2) {
    if (bad_ctx(ctx) || nq < 0 || nr < 0 || (nq > 0 && (!d_q || !d_best)) || (nr > 0 && !d_r)) return SIFT_HIP_ERR_ARG;
    if (ratio_form != 0 && ratio_form != 1) return SIFT_HIP_ERR_ARG;
    if (nq == 0) return SIFT_HIP_OK;
    int rc = run_match(ctx, d_q, nq, d_r, nr, ratio, ratio_form, d_best, d_d1, d_d2);
    if (rc) return rc;
    return check_async(ctx, "match launch");
}

int sift_hip_match_u8_noalloc(sift_hip_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, float ratio,
                              int ratio_form, int32_t* d_best, int32_t* d_d1, int32_t* d_d2) {
    if (bad_ctx(ctx) || nq < 0 || nr < 0 || (nq > 0 && (!d_q || !d_best)) || (nr > 0 && !d_r)) return SIFT_HIP_ERR_ARG;
    if (ratio_form != 0 && ratio_form != 1) return SIFT_HIP_ERR_ARG;
    if (nq == 0) return SIFT_HIP_OK;
    int rc = run_match_noalloc(ctx, d_q, nq, d_r, nr, ratio, ratio_form, d_best, d_d1, d_d2);
    if (rc) return rc;
    return check_async(ctx, "match launch");
}

// RCCL is reached through the copy the calling process already uses (the
// communicator belongs to it): the symbol is looked up at call time, so the
// library itself carries no link dependency on a particular librccl.
typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static allgather_fn find_allgather() {
    static allgather_fn fn = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        fn = (allgather_fn)dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!fn) {
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                    fn = (allgather_fn)dlsym(h, "ncclAllGather");
                    if (fn) break;
                }
            }
        }
    }
    return fn;
}

int sift_hip_match_u8_sharded(sift_hip_ctx* ctx, void* comm, int nranks, const uint8_t* d_q, int nq,
                              const uint8_t* d_r_local, int nr_local, float ratio, int ratio_form, int32_t* d_best,
                              int32_t* d_d1, int32_t* d_d2) {
    if (bad_ctx(ctx) || !comm || nranks < 1 || nq < 0 || nr_local < 0 || (nq > 0 && (!d_q || !d_best)) ||
        (nr_local > 0 && !d_r_local))
        return SIFT_HIP_ERR_ARG;
    if (ratio_form != 0 && ratio_form != 1) return SIFT_HIP_ERR_ARG;
    if ((long long)nr_local * nranks > INT32_MAX) return SIFT_HIP_ERR_LIMIT;
    const int nr_all = nr_local * nranks;
    int rc = ensure(ctx, ctx->gathered, std::max<size_t>(1, (size_t)nr_all * 128));
    if (rc) return rc;
    if (nr_local > 0) {
        allgather_fn ag = find_allgather();
        if (!ag) {
            ctx->last_error = "ncclAllGather not found in this process (load RCCL before calling)";
            return SIFT_HIP_ERR_HIP;
        }
        const int nrc = ag(d_r_local, ctx->gathered.p, (size_t)nr_local * 128, /*ncclUint8*/ 1, comm, ctx->stream);
        if (nrc != 0) {
            ctx->last_error = "ncclAllGather failed with code " + std::to_string(nrc);
            return SIFT_HIP_ERR_HIP;
        }
    }
    if (nq == 0) return SIFT_HIP_OK;
    rc = run_match(ctx, d_q, nq, (const uint8_t*)ctx->gathered.p, nr_all, ratio, ratio_form, d_best, d_d1, d_d2);
    if (rc) return rc;
    return check_async(ctx, "sharded match launch");
}

int sift_hip_ragged_offsets(const int32_t* nr_of_rank, int nranks, int64_t* offsets) {
    if (!nr_of_rank || !offsets || nranks < 1) return SIFT_HIP_ERR_ARG;
    int64_t run = 0;
    int32_t largest = 0;
    for (int k = 0; k < nranks; k++) {
        if (nr_of_rank[k] < 0) return SIFT_HIP_ERR_ARG;
        offsets[k] = run;
        run += nr_of_rank[k];
        largest = std::max(largest, nr_of_rank[k]);
    }
    offsets[nranks] = run;
    if (run > INT32_MAX) return SIFT_HIP_ERR_LIMIT;
    return largest;
}

int sift_hip_match_u8_sharded_ragged(sift_hip_ctx* ctx, void* comm, int nranks, int rank, const uint8_t* d_q, int nq,
                                     const uint8_t* d_r_local, const int32_t* nr_of_rank, float ratio, int ratio_form,
                                     int32_t* d_best, int32_t* d_d1, int32_t* d_d2) {
    if (bad_ctx(ctx) || !comm || nranks < 1 || rank < 0 || rank >= nranks || !nr_of_rank || nq < 0 ||
        (nq > 0 && (!d_q || !d_best)))
        return SIFT_HIP_ERR_ARG;
    if (ratio_form != 0 && ratio_form != 1) return SIFT_HIP_ERR_ARG;
    std::vector<int64_t> off((size_t)nranks + 1);
    const int largest = sift_hip_ragged_offsets(nr_of_rank, nranks, off.data());
    if (largest < 0) return largest;
    const int nr_local = nr_of_rank[rank];
    if (nr_local > 0 && !d_r_local) return SIFT_HIP_ERR_ARG;
    const int nr_all = (int)off[nranks];
    int rc;
    if ((rc = ensure(ctx, ctx->gathered, std::max<size_t>(1, (size_t)nr_all * 128)))) return rc;
    if (largest > 0) {
        // every rank sends `largest` rows (its own, then padding whose content nobody reads) ...
        const size_t slot = (size_t)largest * 128;
        if ((rc = ensure(ctx, ctx->gather_in, slot))) return rc;
        if ((rc = ensure(ctx, ctx->gather_pad, slot * nranks))) return rc;
        allgather_fn ag = find_allgather();
        if (!ag) {
            ctx->last_error = "ncclAllGather not found in this process (load RCCL before calling)";
            return SIFT_HIP_ERR_HIP;
        }
        if (nr_local > 0 &&
            !hip_ok(ctx, hipMemcpyAsync(ctx->gather_in.p, d_r_local, (size_t)nr_local * 128, hipMemcpyDeviceToDevice,
                                        ctx->stream), "stage references"))
            return SIFT_HIP_ERR_HIP;
        const int nrc = ag(ctx->gather_in.p, ctx->gather_pad.p, slot, /*ncclUint8*/ 1, comm, ctx->stream);
        if (nrc != 0) {
            ctx->last_error = "ncclAllGather failed with code " + std::to_string(nrc);
            return SIFT_HIP_ERR_HIP;
        }
        // ... and the padding is squeezed out: rank k's rows move to offset off[k]
        for (int k = 0; k < nranks; k++) {
            if (nr_of_rank[k] == 0) continue;
            if (!hip_ok(ctx, hipMemcpyAsync((uint8_t*)ctx->gathered.p + (size_t)off[k] * 128,
                                            (const uint8_t*)ctx->gather_pad.p + slot * k, (size_t)nr_of_rank[k] * 128,
                                            hipMemcpyDeviceToDevice, ctx->stream), "compact references"))
                return SIFT_HIP_ERR_HIP;
        }
    }
    if (nq == 0) return SIFT_HIP_OK;
    rc = run_match(ctx, d_q, nq, (const uint8_t*)ctx->gathered.p, nr_all, ratio, ratio_form, d_best, d_d1, d_d2);
    if (rc) return rc;
    return check_async(ctx, "sharded match launch");
}

int sift_hip_match_u8(sift_hip_ctx* ctx, const uint8_t* q, int nq, const uint8_t* r, int nr, float ratio,
                      int ratio_form, int32_t* best, int32_t* d1, int32_t* d2) {
    if (bad_ctx(ctx) || nq < 0 || nr < 0 || (nq > 0 && (!q || !best)) || (nr > 0 && !r)) return SIFT_HIP_ERR_ARG;
    if (ratio_form != 0 && ratio_form != 1) return SIFT_HIP_ERR_ARG;
    if (nq == 0) return SIFT_HIP_OK;
    int rc;
    if ((rc = ensure(ctx, ctx->io_a, (size_t)nq * 128))) return rc;
    if ((rc = ensure(ctx, ctx->io_b, (size_t)std::max(nr, 1) * 128))) return rc;
    if ((rc = ensure(ctx, ctx->best, (size_t)nq * 4))) return rc;
    if ((rc = ensure(ctx, ctx->d1, (size_t)nq * 4))) return rc;
    if ((rc = ensure(ctx, ctx->d2, (size_t)nq * 4))) return rc;
    hipMemcpyAsync(ctx->io_a.p, q, (size_t)nq * 128, hipMemcpyHostToDevice, ctx->stream);
    if (nr > 0) hipMemcpyAsync(ctx->io_b.p, r, (size_t)nr * 128, hipMemcpyHostToDevice, ctx->stream);
    rc = run_match(ctx, (const uint8_t*)ctx->io_a.p, nq, (const uint8_t*)ctx->io_b.p, nr, ratio, ratio_form,
                   (int32_t*)ctx->best.p, (int32_t*)ctx->d1.p, (int32_t*)ctx->d2.p);
    if (rc) return rc;
    hipMemcpyAsync(best, ctx->best.p, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (d1) hipMemcpyAsync(d1, ctx->d1.p, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (d2) hipMemcpyAsync(d2, ctx->d2.p, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "match")) return SIFT_HIP_ERR_HIP;
    return check_async(ctx, "match");
}

// Largest image the warps index with 32-bit pixel counts.
static bool warp_size_bad(int cols, int rows) {
    return cols <= 0 || rows <= 0 || (size_t)cols * (size_t)rows > ((size_t)1 << 30);
}

int sift_hip_warp_perspective_dev(sift_hip_ctx* ctx, const uint8_t* d_src_rgb, int s_col, int s_row, const float* H9,
                                  uint8_t* d_dst_rgb, int d_col, int d_row) {
    if (bad_ctx(ctx) || !d_src_rgb || !d_dst_rgb || !H9 || d_src_rgb == d_dst_rgb) return SIFT_HIP_ERR_ARG;
    if (warp_size_bad(s_col, s_row) || warp_size_bad(d_col, d_row)) return SIFT_HIP_ERR_ARG;
    launch_warp_perspective(ctx, ctx->stream, d_src_rgb, s_col, s_row, H9, d_dst_rgb, d_col, d_row);
    return check_async(ctx, "warp_perspective launch");
}

int sift_hip_warp_perspective(sift_hip_ctx* ctx, const uint8_t* src_rgb, int s_col, int s_row, const float* H9,
                              uint8_t* dst_rgb, int d_col, int d_row) {
    if (bad_ctx(ctx) || !src_rgb || !dst_rgb || !H9 || src_rgb == dst_rgb) return SIFT_HIP_ERR_ARG;
    if (warp_size_bad(s_col, s_row) || warp_size_bad(d_col, d_row)) return SIFT_HIP_ERR_ARG;
    const size_t sb = (size_t)s_col * s_row * 3, db = (size_t)d_col * d_row * 3;
    int rc;
    if ((rc = ensure(ctx, ctx->warp_src, sb))) return rc;
    if ((rc = ensure(ctx, ctx->warp_dst, db))) return rc;
    hipMemcpyAsync(ctx->warp_src.p, src_rgb, sb, hipMemcpyHostToDevice, ctx->stream);
    hipMemcpyAsync(ctx->warp_dst.p, dst_rgb, db, hipMemcpyHostToDevice, ctx->stream);  // in/out
    launch_warp_perspective(ctx, ctx->stream, (const uint8_t*)ctx->warp_src.p, s_col, s_row, H9,
                            (uint8_t*)ctx->warp_dst.p, d_col, d_row);
    hipMemcpyAsync(dst_rgb, ctx->warp_dst.p, db, hipMemcpyDeviceToHost, ctx->stream);
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "warp_perspective")) return SIFT_HIP_ERR_HIP;
    return check_async(ctx, "warp_perspective");
}

int sift_hip_warp_cylinder_dev(sift_hip_ctx* ctx, const uint8_t* d_src_rgb, int w, int h, float focal,
                               uint8_t* d_out_rgb, float* d_upedge, float* d_downedge) {
    if (bad_ctx(ctx) || !d_src_rgb || !d_out_rgb || d_src_rgb == d_out_rgb || warp_size_bad(w, h) || !(focal > 0.f))
        return SIFT_HIP_ERR_ARG;
    launch_warp_cylinder(ctx, ctx->stream, d_src_rgb, w, h, focal, d_out_rgb, d_upedge, d_downedge);
    return check_async(ctx, "warp_cylinder launch");
}

int sift_hip_warp_cylinder(sift_hip_ctx* ctx, const uint8_t* src_rgb, int w, int h, float focal, uint8_t* out_rgb,
                           float* upedge, float* downedge) {
    if (bad_ctx(ctx) || !src_rgb || !out_rgb || src_rgb == out_rgb || warp_size_bad(w, h) || !(focal > 0.f))
        return SIFT_HIP_ERR_ARG;
    const size_t nb = (size_t)w * h * 3, eb = (size_t)w * 2 * sizeof(float);
    int rc;
    if ((rc = ensure(ctx, ctx->warp_src, nb))) return rc;
    if ((rc = ensure(ctx, ctx->warp_dst, nb))) return rc;
    if ((rc = ensure(ctx, ctx->warp_edges, 2 * eb))) return rc;
    float* d_up = (float*)ctx->warp_edges.p;
    float* d_down = d_up + (size_t)w * 2;
    hipMemcpyAsync(ctx->warp_src.p, src_rgb, nb, hipMemcpyHostToDevice, ctx->stream);
    hipMemsetAsync(ctx->warp_edges.p, 0, 2 * eb, ctx->stream);  // a one-row image has no separate last row
    launch_warp_cylinder(ctx, ctx->stream, (const uint8_t*)ctx->warp_src.p, w, h, focal, (uint8_t*)ctx->warp_dst.p, d_up,
                         d_down);
    hipMemcpyAsync(out_rgb, ctx->warp_dst.p, nb, hipMemcpyDeviceToHost, ctx->stream);
    if (upedge) hipMemcpyAsync(upedge, d_up, eb, hipMemcpyDeviceToHost, ctx->stream);
    if (downedge) hipMemcpyAsync(downedge, d_down, eb, hipMemcpyDeviceToHost, ctx->stream);
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "warp_cylinder")) return SIFT_HIP_ERR_HIP;
    return check_async(ctx, "warp_cylinder");
}

static bool blend_args_bad(int lw, int lh, int rw, int rh, int dx, int dy) {
    return warp_size_bad(lw, lh) || warp_size_bad(rw, rh) || dx < 1 || dx > 16384 || dy == 0 || dy > 16384 || dy < -16384;
}

int sift_hip_multiband_blend_dev(sift_hip_ctx* ctx, uint8_t* d_l_rgb, int lw, int lh, uint8_t* d_r_rgb, int rw, int rh,
                                 int dx, int dy) {
    if (bad_ctx(ctx) || !d_l_rgb || !d_r_rgb || d_l_rgb == d_r_rgb || blend_args_bad(lw, lh, rw, rh, dx, dy))
        return SIFT_HIP_ERR_ARG;
    int rc;
    {
        ProfScope ps(ctx, SIFT_HIP_K_BLEND, ctx->stream);
        rc = run_multiband_blend(ctx, d_l_rgb, lw, lh, d_r_rgb, rw, rh, dx, dy);
    }
    if (rc) return rc;
    return check_async(ctx, "multiband_blend launch");
}

int sift_hip_multiband_blend(sift_hip_ctx* ctx, uint8_t* l_rgb, int lw, int lh, uint8_t* r_rgb, int rw, int rh, int dx,
                             int dy) {
    if (bad_ctx(ctx) || !l_rgb || !r_rgb || l_rgb == r_rgb || blend_args_bad(lw, lh, rw, rh, dx, dy)) return SIFT_HIP_ERR_ARG;
    const size_t lb = (size_t)lw * lh * 3, rb = (size_t)rw * rh * 3;
    int rc;
    if ((rc = ensure(ctx, ctx->blend_l, lb))) return rc;
    if ((rc = ensure(ctx, ctx->blend_r, rb))) return rc;
    if (!hip_ok(ctx, hipMemcpyAsync(ctx->blend_l.p, l_rgb, lb, hipMemcpyHostToDevice, ctx->stream), "upload left image") ||
        !hip_ok(ctx, hipMemcpyAsync(ctx->blend_r.p, r_rgb, rb, hipMemcpyHostToDevice, ctx->stream), "upload right image"))
        return SIFT_HIP_ERR_HIP;
    rc = sift_hip_multiband_blend_dev(ctx, (uint8_t*)ctx->blend_l.p, lw, lh, (uint8_t*)ctx->blend_r.p, rw, rh, dx, dy);
    if (rc) return rc;
    if (!hip_ok(ctx, hipMemcpyAsync(l_rgb, ctx->blend_l.p, lb, hipMemcpyDeviceToHost, ctx->stream), "download left image") ||
        !hip_ok(ctx, hipMemcpyAsync(r_rgb, ctx->blend_r.p, rb, hipMemcpyDeviceToHost, ctx->stream), "download right image"))
        return SIFT_HIP_ERR_HIP;
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "multiband_blend")) return SIFT_HIP_ERR_HIP;
    return check_async(ctx, "multiband_blend");
}

// Shared part of the two panorama entry points: projections of the two 8-bit images already on the device, blend,
// canvas, download.
static int panorama_from_device(sift_hip_ctx* ctx, const uint8_t* d_src1, int w1, int h1, const uint8_t* d_src2, int w2, int h2,
                                float focal, int dx, int blend_dy, int top0, int left1, int top1, uint8_t* canvas_rgb, int cols,
                                int rows) {
    hipStream_t st = ctx->stream;
    // warping (Blend.cpp:974): both projections stay on the device
    launch_warp_cylinder(ctx, st, d_src1, w1, h1, focal, (uint8_t*)ctx->blend_l.p, nullptr, nullptr);
    launch_warp_cylinder(ctx, st, d_src2, w2, h2, focal, (uint8_t*)ctx->blend_r.p, nullptr, nullptr);
    // multiBandBlend (Blend.cpp:994-996) on the projections, then the canvas (Blend.cpp:1010-1031)
    const int rc = sift_hip_multiband_blend_dev(ctx, (uint8_t*)ctx->blend_l.p, w1, h1, (uint8_t*)ctx->blend_r.p, w2, h2, dx, blend_dy);
    if (rc) return rc;
    launch_compose_pair(ctx, st, (const uint8_t*)ctx->blend_l.p, w1, h1, (const uint8_t*)ctx->blend_r.p, w2, h2, top0, left1,
                        top1, (uint8_t*)ctx->pano_canvas.p, cols, rows);
    if (!hip_ok(ctx, hipMemcpyAsync(canvas_rgb, ctx->pano_canvas.p, (size_t)cols * rows * 3, hipMemcpyDeviceToHost, st),
                "download canvas") ||
        !hip_ok(ctx, hipStreamSynchronize(st), "panorama_pair"))
        return SIFT_HIP_ERR_HIP;
    return check_async(ctx, "panorama_pair");
}

static int panorama_buffers(sift_hip_ctx* ctx, size_t b1, size_t b2, size_t bc, size_t* off2) {
    *off2 = (b1 + 255) / 256 * 256;  // the second image starts on an aligned address of the shared buffer
    int rc;
    if ((rc = ensure(ctx, ctx->pano_src, *off2 + b2))) return rc;
    if ((rc = ensure(ctx, ctx->blend_l, b1))) return rc;
    if ((rc = ensure(ctx, ctx->blend_r, b2))) return rc;
    return ensure(ctx, ctx->pano_canvas, bc);
}

int sift_hip_panorama_pair(sift_hip_ctx* ctx, const uint8_t* img1_rgb, int w1, int h1, const uint8_t* img2_rgb, int w2,
                           int h2, float focal, int dx, int blend_dy, int top0, int left1, int top1, uint8_t* canvas_rgb,
                           int cols, int rows) {
    if (bad_ctx(ctx) || !img1_rgb || !img2_rgb || !canvas_rgb || !(focal > 0.f) || blend_args_bad(w1, h1, w2, h2, dx, blend_dy) ||
        warp_size_bad(cols, rows))
        return SIFT_HIP_ERR_ARG;
    const size_t b1 = (size_t)w1 * h1 * 3, b2 = (size_t)w2 * h2 * 3, bc = (size_t)cols * rows * 3;
    size_t off2;
    int rc;
    if ((rc = panorama_buffers(ctx, b1, b2, bc, &off2))) return rc;
    uint8_t* d_src1 = (uint8_t*)ctx->pano_src.p;
    uint8_t* d_src2 = d_src1 + off2;
    hipStream_t st = ctx->stream;
    if (!hip_ok(ctx, hipMemcpyAsync(d_src1, img1_rgb, b1, hipMemcpyHostToDevice, st), "upload image 1") ||
        !hip_ok(ctx, hipMemcpyAsync(d_src2, img2_rgb, b2, hipMemcpyHostToDevice, st), "upload image 2"))
        return SIFT_HIP_ERR_HIP;
    return panorama_from_device(ctx, d_src1, w1, h1, d_src2, w2, h2, focal, dx, blend_dy, top0, left1, top1, canvas_rgb, cols,
                                rows);
}

int sift_hip_panorama_pair_f32(sift_hip_ctx* ctx, const float* img1_rgb, int w1, int h1, const float* img2_rgb, int w2, int h2,
                               int normalised, float focal, int dx, int blend_dy, int top0, int left1, int top1,
                               uint8_t* canvas_rgb, int cols, int rows) {
    if (bad_ctx(ctx) || !img1_rgb || !img2_rgb || !canvas_rgb || !(focal > 0.f) || blend_args_bad(w1, h1, w2, h2, dx, blend_dy) ||
        warp_size_bad(cols, rows))
        return SIFT_HIP_ERR_ARG;
    const size_t b1 = (size_t)w1 * h1 * 3, b2 = (size_t)w2 * h2 * 3, bc = (size_t)cols * rows * 3;
    size_t off2;
    int rc;
    if ((rc = panorama_buffers(ctx, b1, b2, bc, &off2))) return rc;
    // one staging buffer for the float samples of either image: the stream orders the second upload behind the
    // first conversion
    if ((rc = ensure(ctx, ctx->pano_f32, std::max(b1, b2) * sizeof(float)))) return rc;
    uint8_t* d_src1 = (uint8_t*)ctx->pano_src.p;
    uint8_t* d_src2 = d_src1 + off2;
    float* d_f = (float*)ctx->pano_f32.p;
    hipStream_t st = ctx->stream;
    if (!hip_ok(ctx, hipMemcpyAsync(d_f, img1_rgb, b1 * sizeof(float), hipMemcpyHostToDevice, st), "upload image 1"))
        return SIFT_HIP_ERR_HIP;
    launch_float_to_u8(ctx, st, d_f, b1, normalised != 0, d_src1);
    if (!hip_ok(ctx, hipMemcpyAsync(d_f, img2_rgb, b2 * sizeof(float), hipMemcpyHostToDevice, st), "upload image 2"))
        return SIFT_HIP_ERR_HIP;
    launch_float_to_u8(ctx, st, d_f, b2, normalised != 0, d_src2);
    return panorama_from_device(ctx, d_src1, w1, h1, d_src2, w2, h2, focal, dx, blend_dy, top0, left1, top1, canvas_rgb, cols,
                                rows);
}

static int ransac_args_bad(sift_hip_ctx* ctx, const void* q, int nq, const void* r, int nr, const void* best, int m,
                           float p_badxform, float err_tol, const int32_t* n_in) {
    if (bad_ctx(ctx) || nq < 0 || nr < 0 || !n_in || (nq > 0 && (!q || !best)) || (nr > 0 && !r)) return 1;
    return m < 4 || m > 8 || !(p_badxform > 0.f) || !(p_badxform < 1.f) || !(err_tol >= 0.f);
}

int sift_hip_ransac_homog_dev(sift_hip_ctx* ctx, const sift_hip_feat* d_q_feats, int nq,
                              const sift_hip_feat* d_r_feats, int nr, const int32_t* d_best, int m, float p_badxform,
                              float err_tol, unsigned seed, float* H9, int32_t* inlier_idx, int32_t* n_in,
                              int32_t* iterations) {
    if (ransac_args_bad(ctx, d_q_feats, nq, d_r_feats, nr, d_best, m, p_badxform, err_tol, n_in))
        return SIFT_HIP_ERR_ARG;
    const int rc = run_ransac(ctx, d_q_feats, nq, d_r_feats, nr, d_best, m, p_badxform, err_tol, seed, H9,
                              inlier_idx, n_in, iterations);
    if (rc) return rc;
    return check_async(ctx, "ransac");
}

int sift_hip_ransac_homog(sift_hip_ctx* ctx, const sift_hip_feat* q_feats, int nq, const sift_hip_feat* r_feats,
                          int nr, const int32_t* best, int m, float p_badxform, float err_tol, unsigned seed,
                          float* H9, int32_t* inlier_idx, int32_t* n_in, int32_t* iterations) {
    if (ransac_args_bad(ctx, q_feats, nq, r_feats, nr, best, m, p_badxform, err_tol, n_in)) return SIFT_HIP_ERR_ARG;
    sifthip::RansacWs& ws = ctx->ransac;
    int rc;
    if ((rc = ensure(ctx, ws.qf, (size_t)std::max(nq, 1) * sizeof(sift_hip_feat)))) return rc;
    if ((rc = ensure(ctx, ws.rf, (size_t)std::max(nr, 1) * sizeof(sift_hip_feat)))) return rc;
    if ((rc = ensure(ctx, ws.best, (size_t)std::max(nq, 1) * 4))) return rc;
    if (nq > 0) {
        hipMemcpyAsync(ws.qf.p, q_feats, (size_t)nq * sizeof(sift_hip_feat), hipMemcpyHostToDevice, ctx->stream);
        hipMemcpyAsync(ws.best.p, best, (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream);
    }
    if (nr > 0) hipMemcpyAsync(ws.rf.p, r_feats, (size_t)nr * sizeof(sift_hip_feat), hipMemcpyHostToDevice, ctx->stream);
    return sift_hip_ransac_homog_dev(ctx, (const sift_hip_feat*)ws.qf.p, nq, (const sift_hip_feat*)ws.rf.p, nr,
                                     (const int32_t*)ws.best.p, m, p_badxform, err_tol, seed, H9, inlier_idx, n_in,
                                     iterations);
}

int sift_hip_ransac_homog_pairs(sift_hip_ctx* ctx, const float* xyuv, int n, int m, float p_badxform, float err_tol,
                                unsigned seed, float* H9, int32_t* inlier_idx, int32_t* n_in, int32_t* iterations) {
    if (bad_ctx(ctx) || n < 0 || !n_in || (n > 0 && !xyuv)) return SIFT_HIP_ERR_ARG;
    if (m < 4 || m > 8 || !(p_badxform > 0.f) || !(p_badxform < 1.f) || !(err_tol >= 0.f)) return SIFT_HIP_ERR_ARG;
    const int rc = run_ransac_pairs(ctx, xyuv, n, m, p_badxform, err_tol, seed, H9, inlier_idx, n_in, iterations);
    if (rc) return rc;
    return check_async(ctx, "ransac");
}

int sift_hip_last_offsets(sift_hip_ctx* ctx, float* xyz, int cap) {
    if (!ctx || cap < 0 || (cap > 0 && !xyz)) return SIFT_HIP_ERR_ARG;
    const int n = (int)std::min<size_t>(ctx->h_sub.size() / 3, (size_t)cap);
    if (n > 0) std::memcpy(xyz, ctx->h_sub.data(), (size_t)n * 3 * sizeof(float));
    return n;
}

struct sift_hip_descr_set {
    uint8_t* d = nullptr;  // n x 128 bytes
    int n = 0;
    int device = 0;
};

int sift_hip_keep_descriptors(sift_hip_ctx* ctx, sift_hip_descr_set** set, int* n) {
    if (bad_ctx(ctx) || !set) return SIFT_HIP_ERR_ARG;
    *set = nullptr;
    if (ctx->last_feat_ptr.empty()) {
        ctx->last_error = "keep_descriptors: no extract of this context has delivered features since the last one began";
        return SIFT_HIP_ERR_ARG;
    }
    long long total = 0;
    for (int c : ctx->last_feat_cnt) total += c;
    sift_hip_descr_set* ds = new (std::nothrow) sift_hip_descr_set;
    if (!ds) return SIFT_HIP_ERR_HIP;
    ds->n = (int)total;
    ds->device = ctx->device;
    if (!hip_ok(ctx, hipMalloc((void**)&ds->d, (size_t)std::max<long long>(total, 1) * 128), "keep_descriptors")) {
        delete ds;
        return SIFT_HIP_ERR_HIP;
    }
    size_t done = 0;
    for (size_t oi = 0; oi < ctx->last_feat_ptr.size(); oi++) {
        launch_gather_descr(ctx, ctx->stream, ctx->last_feat_ptr[oi], ctx->last_feat_cnt[oi], ds->d + done * 128);
        done += (size_t)ctx->last_feat_cnt[oi];
    }
    // the record arrays may be rewritten by the next extract as soon as this call returns
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "keep_descriptors")) {
        hipFree(ds->d);
        delete ds;
        return SIFT_HIP_ERR_HIP;
    }
    const int rc = check_async(ctx, "keep_descriptors");
    if (rc) {
        hipFree(ds->d);
        delete ds;
        return rc;
    }
    *set = ds;
    if (n) *n = ds->n;
    return SIFT_HIP_OK;
}

void sift_hip_release_descriptors(sift_hip_ctx* ctx, sift_hip_descr_set* set) {
    if (!set) return;
    if (ctx && ctx->device == set->device) hipSetDevice(ctx->device);
    if (set->d) hipFree(set->d);
    delete set;
}

int sift_hip_descriptor_count(const sift_hip_descr_set* set) { return set ? set->n : SIFT_HIP_ERR_ARG; }

int sift_hip_match_kept(sift_hip_ctx* ctx, const sift_hip_descr_set* q, const sift_hip_descr_set* r, float ratio,
                        int ratio_form, int32_t* best, int32_t* d1, int32_t* d2) {
    if (bad_ctx(ctx) || !q || !r || (q->n > 0 && !best)) return SIFT_HIP_ERR_ARG;
    if (q->device != ctx->device || r->device != ctx->device) return SIFT_HIP_ERR_ARG;
    if (ratio_form != 0 && ratio_form != 1) return SIFT_HIP_ERR_ARG;
    const int nq = q->n, nr = r->n;
    if (nq == 0) return SIFT_HIP_OK;
    int rc;
    if ((rc = ensure(ctx, ctx->best, (size_t)nq * 4))) return rc;
    if ((rc = ensure(ctx, ctx->d1, (size_t)nq * 4))) return rc;
    if ((rc = ensure(ctx, ctx->d2, (size_t)nq * 4))) return rc;
    rc = run_match(ctx, q->d, nq, r->d, nr, ratio, ratio_form, (int32_t*)ctx->best.p, (int32_t*)ctx->d1.p, (int32_t*)ctx->d2.p);
    if (rc) return rc;
    if (!hip_ok(ctx, hipMemcpyAsync(best, ctx->best.p, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream), "read matches"))
        return SIFT_HIP_ERR_HIP;
    if (d1 && !hip_ok(ctx, hipMemcpyAsync(d1, ctx->d1.p, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream), "read matches"))
        return SIFT_HIP_ERR_HIP;
    if (d2 && !hip_ok(ctx, hipMemcpyAsync(d2, ctx->d2.p, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream), "read matches"))
        return SIFT_HIP_ERR_HIP;
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "match")) return SIFT_HIP_ERR_HIP;
    return check_async(ctx, "match");
}

int sift_hip_profile(sift_hip_ctx* ctx, int enable) {
    if (!ctx) return SIFT_HIP_ERR_ARG;
    ctx->profiling = enable != 0;
    return SIFT_HIP_OK;
}

int sift_hip_profile_read(sift_hip_ctx* ctx, float* ms, int* launches, int reset) {
    if (bad_ctx(ctx)) return SIFT_HIP_ERR_ARG;
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "profile_read")) return SIFT_HIP_ERR_HIP;
    if (ctx->timeline && !ctx->events.empty()) {
        hipStreamSynchronize(ctx->side_stream);
        hipStreamSynchronize(ctx->side_stream2);
        hipStreamSynchronize(ctx->side_stream3);
        // a frame starts at a "resample" group on the main stream; times are relative to the last one
        size_t first = 0;
        for (size_t i = 0; i < ctx->events.size(); i++)
            if (ctx->events[i].cls == SIFT_HIP_K_RESAMPLE && ctx->events[i].st == ctx->stream) first = i;
        for (size_t i = first; i < ctx->events.size(); i++) {
            const auto& e = ctx->events[i];
            float t0 = 0.f, t1 = 0.f;
            hipEventElapsedTime(&t0, ctx->events[first].a, e.a);
            hipEventElapsedTime(&t1, ctx->events[first].a, e.b);
            const int q = e.st == ctx->stream ? 0 : (e.st == ctx->side_stream ? 1 : (e.st == ctx->side_stream2 ? 2 : 3));
            std::fprintf(stderr, "timeline q%d %*s%8.1f + %7.1f  %s\n", q, 4 * q, "", t0 * 1e3f, (t1 - t0) * 1e3f,
                         sift_hip_kernel_class_name(e.cls));
        }
    }
    for (auto& e : ctx->events) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) {
            ctx->prof_ms[e.cls] += t;
            ctx->prof_n[e.cls] += 1;
        }
        ctx->free_events.push_back(e.a);
        ctx->free_events.push_back(e.b);
    }
    ctx->events.clear();
    for (int k = 0; k < SIFT_HIP_K_COUNT; k++) {
        if (ms) ms[k] = ctx->prof_ms[k];
        if (launches) launches[k] = ctx->prof_n[k];
        if (reset) { ctx->prof_ms[k] = 0.f; ctx->prof_n[k] = 0; }
    }
    return SIFT_HIP_OK;
}

const char* sift_hip_kernel_class_name(int k) {
    static const char* names[SIFT_HIP_K_COUNT] = {"resample", "blur_row", "blur_col", "detect", "compact", "orient",
                                                  "emit",     "descr",    "match",    "gray",   "blur_row_small",
                                                  "blur_col_small", "ransac", "warp", "blend", "blur_fused"};
    return (k >= 0 && k < SIFT_HIP_K_COUNT) ? names[k] : "?";
}

}  // extern "C"
