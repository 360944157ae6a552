// RANSAC homography with the hypotheses fitted and scored on the GPU.
//
// The reference's loop (Sift/ransac/xform.cpp:64-95) is sequential only in how
// many samples it draws: sample k never depends on the outcome of sample k-1,
// the outcomes only decide when to stop.  So the host draws a batch of samples
// from the same libc stream, the device fits every sample (one thread each, the
// QR solve of homog.h) and counts every model's consensus over all matches
// (find_consensus, xform.cpp:365-395), and the host replays the loop's
// bookkeeping over the counts in sample order, stopping where the reference
// would.  Models, counts and therefore the result equal the host
// implementation's (sift_amd/host/xform.cpp) bit for bit on the same seed.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include "homog.h"
#include "internal.h"

namespace sifthip {
namespace {

constexpr int PB = 256;        // matches per block in the gather and score kernels
constexpr int HB = 32;         // models a score block holds in LDS
constexpr int MAX_M = 8;       // largest sample size the fit kernel has room for
constexpr int MAX_BATCH = 2048;
constexpr float kProbBadSupp = 0.10f;  // RANSAC_PROB_BAD_SUPP, xform.cpp:24
constexpr float kInlierFracEst = 0.25f;  // RANSAC_INLIER_FRAC_EST, xform.cpp:22

__device__ inline bool has_match(const int32_t* best, int i, int nq, int nr) {
    return i < nq && (uint32_t)best[i] < (uint32_t)nr;
}

// get_matched_features (xform.cpp:222-244), step 1: matches per block of PB queries.
__global__ __launch_bounds__(PB) void pair_count_kernel(const int32_t* __restrict__ best, int nq, int nr,
                                                        int32_t* __restrict__ blk_cnt) {
    const int c = __syncthreads_count(has_match(best, blockIdx.x * PB + threadIdx.x, nq, nr));
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = c;
}

// Step 2: the matched pairs in query order.  pairs[j] = (x, y, u, v): the query
// feature's source-image position and its match's (Feature::rX()/rY(),
// imagedata.hpp:82-83: octave coordinate / octave scale); src[j] = query index.
__global__ __launch_bounds__(PB) void pair_write_kernel(const sift_hip_feat* __restrict__ q,
                                                        const sift_hip_feat* __restrict__ r,
                                                        const int32_t* __restrict__ best, int nq, int nr,
                                                        const int32_t* __restrict__ blk_cnt, float4* __restrict__ pairs,
                                                        int32_t* __restrict__ src, int32_t* __restrict__ total) {
    __shared__ int32_t s_part[PB / 64], s_wave[PB / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int acc = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += PB) acc += blk_cnt[b];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    const int i = blockIdx.x * PB + threadIdx.x;
    const bool f = has_match(best, i, nq, nr);
    const uint64_t bal = __ballot(f);
    if (lane == 0) {
        s_part[wv] = acc;
        s_wave[wv] = __popcll(bal);
    }
    __syncthreads();
    int off = 0;
    for (int w = 0; w < PB / 64; w++) off += s_part[w] + (w < wv ? s_wave[w] : 0);
    if (f) {
        const int j = off + __popcll(bal & ((1ull << lane) - 1ull));
        const sift_hip_feat* a = q + i;
        const sift_hip_feat* c = r + best[i];
        pairs[j] = make_float4(a->x / a->size, a->y / a->size, c->x / c->size, c->y / c->size);
        src[j] = i;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == PB - 1) *total = off + s_wave[PB / 64 - 1];
}

// lsq_homog on each sample of m pairs (xform.cpp:66-70): one thread per sample.
__global__ __launch_bounds__(64) void fit_kernel(const float4* __restrict__ pairs, const int32_t* __restrict__ samples,
                                                 int m, int nsamp, float* __restrict__ H, int32_t* __restrict__ valid) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= nsamp) return;
    double M[2 * MAX_M * sift_homog::NAUG];
    for (int i = 0; i < m; i++) {
        const float4 p = pairs[samples[s * m + i]];
        sift_homog::dlt_rows(p.x, p.y, p.z, p.w, &M[(2 * i) * sift_homog::NAUG], &M[(2 * i + 1) * sift_homog::NAUG]);
    }
    double h[sift_homog::NCOL];
    const int ok = sift_homog::solve_lsq(M, 2 * m, h);
    for (int k = 0; k < 8; k++) H[s * 9 + k] = ok ? (float)h[k] : 0.f;
    H[s * 9 + 8] = 1.f;
    valid[s] = ok;
}

// Consensus size of every model (xform.cpp:74): a block takes PB matches and HB
// models; each thread keeps its match in registers and walks the models in LDS.
__global__ __launch_bounds__(PB) void score_kernel(const float4* __restrict__ pairs, int nm, const float* __restrict__ H,
                                                   int nmodels, float err_tol, int32_t* __restrict__ counts) {
    __shared__ float s_H[HB * 9];
    __shared__ int32_t s_cnt[HB];
    const int h0 = blockIdx.y * HB;
    const int nh = min(HB, nmodels - h0);
    for (int t = threadIdx.x; t < nh * 9; t += PB) s_H[t] = H[h0 * 9 + t];
    if (threadIdx.x < HB) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * PB + threadIdx.x;
    const bool live = i < nm;
    const float4 p = live ? pairs[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < nh; j++) {
        const bool in = live && sift_homog::homog_xfer_err(&s_H[j * 9], p.x, p.y, p.z, p.w) <= err_tol;
        const uint64_t bal = __ballot(in);
        if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&s_cnt[j], (int32_t)__popcll(bal));
    }
    __syncthreads();
    if ((int)threadIdx.x < nh && s_cnt[threadIdx.x]) atomicAdd(&counts[h0 + threadIdx.x], s_cnt[threadIdx.x]);
}

struct Model {
    float h[9];
};

// The consensus of one model as a flag per match (the host lists them in order).
__global__ __launch_bounds__(PB) void flag_kernel(const float4* __restrict__ pairs, int nm, Model T, float err_tol,
                                                  uint8_t* __restrict__ flags) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= nm) return;
    const float4 p = pairs[i];
    flags[i] = sift_homog::homog_xfer_err(T.h, p.x, p.y, p.z, p.w) <= err_tol ? 1 : 0;
}

}  // namespace

// The loop over device-resident pairs d_pairs[0..nm) (host copy: xyuv); srand() was
// called by the entry point.  src maps pair -> reported index (null: identity).
static int ransac_core(sift_hip_ctx* ctx, const float4* d_pairs, int nm, const float* xyuv, const int32_t* src, int m,
                       float p_badxform, float err_tol, float* H9, int32_t* inlier_idx, int32_t* n_in,
                       int32_t* iterations) {
    hipStream_t st = ctx->stream;
    RansacWs& ws = ctx->ransac;
    int rc;
    if ((rc = ensure(ctx, ws.flags, (size_t)nm))) return rc;
    if ((rc = ensure(ctx, ws.samples, (size_t)MAX_BATCH * MAX_M * 4))) return rc;
    if ((rc = ensure(ctx, ws.models, (size_t)MAX_BATCH * 9 * 4))) return rc;
    if ((rc = ensure(ctx, ws.counts, (size_t)MAX_BATCH * 2 * 4))) return rc;

    // ---- the sampling loop, in batches -----------------------------------------------
    const int in_min = sift_homog::ransac_min_inliers(nm, m, kProbBadSupp, p_badxform);
    float in_frac = kInlierFracEst;
    int k = 0, in_max = 0;
    float p = sift_homog::ransac_p_bad(in_frac, m, k);
    Model best_model{};
    std::vector<int32_t> samples, counts;
    std::vector<float> models;
    int32_t* d_samples = (int32_t*)ws.samples.p;
    float* d_models = (float*)ws.models.p;
    int32_t* d_counts = (int32_t*)ws.counts.p;
    int32_t* d_valid = d_counts + MAX_BATCH;
    while (p > p_badxform) {
        // samples the loop still needs if the consensus fraction stays where it is
        int batch = MAX_BATCH;
        const double q = 1.0 - std::pow((double)in_frac, (double)m);
        if (q <= 0.0) {
            batch = 1;
        } else if (q < 1.0) {
            const double need = std::ceil(std::log((double)p_badxform) / std::log(q)) - (double)k + 1.0;
            batch = (int)std::min<double>(std::max(need, 1.0), (double)MAX_BATCH);
        }
        samples.resize((size_t)batch * m);
        for (int s = 0; s < batch; s++) {
            // draw_ransac_sample (xform.cpp:284-306): m distinct pairs, redrawing on repeats
            int32_t* sm = &samples[(size_t)s * m];
            for (int i = 0; i < m; i++) {
                int x;
                bool again;
                do {
                    x = rand() % nm;
                    again = false;
                    for (int t = 0; t < i; t++) again |= sm[t] == x;
                } while (again);
                sm[i] = x;
            }
        }
        hipMemcpyAsync(d_samples, samples.data(), samples.size() * 4, hipMemcpyHostToDevice, st);
        hipMemsetAsync(d_counts, 0, (size_t)batch * 4, st);
        {
            ProfScope ps(ctx, SIFT_HIP_K_RANSAC, st);
            fit_kernel<<<(batch + 63) / 64, 64, 0, st>>>(d_pairs, d_samples, m, batch, d_models, d_valid);
            score_kernel<<<dim3((nm + PB - 1) / PB, (batch + HB - 1) / HB), PB, 0, st>>>(d_pairs, nm, d_models, batch,
                                                                                         err_tol, d_counts);
        }
        counts.resize((size_t)batch * 2);
        models.resize((size_t)batch * 9);
        hipMemcpyAsync(counts.data(), d_counts, (size_t)batch * 4, hipMemcpyDeviceToHost, st);
        hipMemcpyAsync(counts.data() + batch, d_valid, (size_t)batch * 4, hipMemcpyDeviceToHost, st);
        hipMemcpyAsync(models.data(), d_models, (size_t)batch * 36, hipMemcpyDeviceToHost, st);
        if (!hip_ok(ctx, hipStreamSynchronize(st), "ransac batch")) return SIFT_HIP_ERR_HIP;
        for (int s = 0; s < batch && p > p_badxform; s++) {
            if (counts[(size_t)batch + s] && counts[(size_t)s] > in_max) {  // strictly larger (xform.cpp:77)
                in_max = counts[(size_t)s];
                std::memcpy(best_model.h, &models[(size_t)s * 9], 36);
                in_frac = (float)in_max / (float)nm;
            }
            p = sift_homog::ransac_p_bad(in_frac, m, ++k);
        }
    }
    if (!hip_ok(ctx, hipStreamSynchronize(st), "ransac")) return SIFT_HIP_ERR_HIP;
    if (iterations) *iterations = k;
    if (in_max < in_min) return SIFT_HIP_OK;  // xform.cpp:150-156

    // ---- xform.cpp:101-146: refit on the best consensus, that model's consensus, last fit ----
    std::vector<uint8_t> flags((size_t)nm);
    auto consensus_of = [&](const Model& T, std::vector<int>& out) -> bool {
        {
            ProfScope ps(ctx, SIFT_HIP_K_RANSAC, st);
            flag_kernel<<<(nm + PB - 1) / PB, PB, 0, st>>>(d_pairs, nm, T, err_tol, (uint8_t*)ws.flags.p);
        }
        hipMemcpyAsync(flags.data(), ws.flags.p, (size_t)nm, hipMemcpyDeviceToHost, st);
        if (!hip_ok(ctx, hipStreamSynchronize(st), "ransac consensus")) return false;
        out.clear();
        for (int i = 0; i < nm; i++)
            if (flags[(size_t)i]) out.push_back(i);
        return true;
    };
    std::vector<int> best, fin;
    if (!consensus_of(best_model, best)) return SIFT_HIP_ERR_HIP;
    Model M{};
    if (!sift_homog::fit_pairs(xyuv, best.data(), (int)best.size(), M.h)) return SIFT_HIP_OK;
    if (!consensus_of(M, fin)) return SIFT_HIP_ERR_HIP;
    const std::vector<int> order = sift_homog::last_fit_order(xyuv, fin);
    const bool ok = sift_homog::fit_pairs(xyuv, order.data(), (int)order.size(), M.h);
    if (!ok) return SIFT_HIP_OK;
    if (H9) std::memcpy(H9, M.h, sizeof M.h);
    if (inlier_idx)
        for (size_t i = 0; i < fin.size(); i++) inlier_idx[i] = src ? src[(size_t)fin[i]] : (int32_t)fin[i];
    *n_in = (int32_t)fin.size();
    return SIFT_HIP_OK;
}


int run_ransac(sift_hip_ctx* ctx, const sift_hip_feat* d_q, int nq, const sift_hip_feat* d_r, int nr,
               const int32_t* d_best, int m, float p_badxform, float err_tol, unsigned seed, float* H9,
               int32_t* inlier_idx, int32_t* n_in, int32_t* iterations) {
    hipStream_t st = ctx->stream;
    *n_in = -1;
    if (iterations) *iterations = 0;
    srand(seed ? seed : (unsigned)time(nullptr));  // xform.cpp:42
    if (nq == 0 || nr == 0) return SIFT_HIP_OK;

    // ---- matched pairs, in query order ---------------------------------------------
    const int nblk = (nq + PB - 1) / PB;
    int rc;
    RansacWs& ws = ctx->ransac;
    if ((rc = ensure(ctx, ws.blk, (size_t)(nblk + 1) * 4))) return rc;
    if ((rc = ensure(ctx, ws.pairs, (size_t)nq * sizeof(float4)))) return rc;
    if ((rc = ensure(ctx, ws.src, (size_t)nq * 4))) return rc;
    int32_t* d_blk = (int32_t*)ws.blk.p;
    int32_t* d_total = d_blk + nblk;
    float4* d_pairs = (float4*)ws.pairs.p;
    int32_t* d_src = (int32_t*)ws.src.p;
    {
        ProfScope ps(ctx, SIFT_HIP_K_RANSAC, st);
        pair_count_kernel<<<nblk, PB, 0, st>>>(d_best, nq, nr, d_blk);
        pair_write_kernel<<<nblk, PB, 0, st>>>(d_q, d_r, d_best, nq, nr, d_blk, d_pairs, d_src, d_total);
    }
    int32_t nm = 0;
    hipMemcpyAsync(&nm, d_total, 4, hipMemcpyDeviceToHost, st);
    if (!hip_ok(ctx, hipStreamSynchronize(st), "ransac pairs")) return SIFT_HIP_ERR_HIP;
    if (nm < m) return SIFT_HIP_OK;  // "not enough matches to compute xform" (xform.cpp:53-56)
    std::vector<float> xyuv((size_t)nm * 4);
    std::vector<int32_t> src((size_t)nm);
    hipMemcpyAsync(xyuv.data(), d_pairs, (size_t)nm * 16, hipMemcpyDeviceToHost, st);
    hipMemcpyAsync(src.data(), d_src, (size_t)nm * 4, hipMemcpyDeviceToHost, st);
    if (!hip_ok(ctx, hipStreamSynchronize(st), "ransac pairs")) return SIFT_HIP_ERR_HIP;
    return ransac_core(ctx, d_pairs, nm, xyuv.data(), src.data(), m, p_badxform, err_tol, H9, inlier_idx, n_in,
                       iterations);
}

// The same over pairs the caller has already formed: xyuv = n x (x, y, u, v) in host memory.
int run_ransac_pairs(sift_hip_ctx* ctx, const float* xyuv, int n, int m, float p_badxform, float err_tol,
                     unsigned seed, float* H9, int32_t* inlier_idx, int32_t* n_in, int32_t* iterations) {
    *n_in = -1;
    if (iterations) *iterations = 0;
    srand(seed ? seed : (unsigned)time(nullptr));  // xform.cpp:42
    if (n < m) return SIFT_HIP_OK;
    int rc;
    if ((rc = ensure(ctx, ctx->ransac.pairs, (size_t)n * sizeof(float4)))) return rc;
    hipMemcpyAsync(ctx->ransac.pairs.p, xyuv, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream);
    return ransac_core(ctx, (const float4*)ctx->ransac.pairs.p, n, xyuv, nullptr, m, p_badxform, err_tol, H9,
                       inlier_idx, n_in, iterations);
}

}  // namespace sifthip
