// Exact nearest / second-nearest neighbour search over 128-byte descriptors on the
// gfx950 matrix cores (v_mfma_i32_32x32x32_i8), with the ratio test fused.
//
// Semantics: forceMatchFeat (Sift.cpp:537-567) -- every query against every
// reference, strict '<' updates (lowest reference index wins ties, an equal
// runner-up distance makes d2 == d1) -- with the acceptance test of the live
// call site (Sift.cpp:573-578) or of forceMatchFeat (:563), chosen by ratio_form.
//
// Arithmetic: descriptors are bytes; x - 128 fits int8 and leaves distances
// unchanged, so |a-b|^2 = |a'|^2 + |b'|^2 - 2 a'.b' with the dot product on the
// int8 MFMA and everything else in int32: exact.  The search maximises
// s = 2 a'.b' - |b'|^2 per query through keys key = (dot << 9) + c_ref with
// c_ref = -|b'|^2 * 256 + 255 - slot (slot = reference index mod 256): the low
// byte makes the lowest index win ties and is decoded every 256 references.  The
// running top-2 update is m2 = med3(m1, m2, key), m1 = max(m1, key), but it is
// reached only by tiles that pass a max-over-tile threshold test (see match_kernel).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>

#include "internal.h"

namespace sifthip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kNone = (int)0x80000000;  // "no key yet"
#ifndef MATCH_RT
#define MATCH_RT 256
#endif
constexpr int RT = MATCH_RT;            // references per LDS stage
constexpr int GROUP = 256;              // references per key-decoding group
// NB = 32-query blocks per wave (template parameter): 4 when there are enough
// queries to fill the chip with 512-query workgroups, else 2.

// bytes -> int8 (x - 128), row norms; mode 0: norm[i] = |row|^2 (queries),
// mode 1: norm[i] = c_ref >> 9 and norm[n_pad + i] = c_ref & 511 (references).
// Rows >= n are zero padding.
__global__ __launch_bounds__(256) void match_prep_kernel(const uint8_t* __restrict__ src, int n, int n_pad,
                                                         int8_t* __restrict__ dst, int* __restrict__ norm, int mode) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int row = t >> 3, ch = t & 7;
    if (row >= n_pad) return;  // n_pad*8 is a multiple of 64: whole waves leave together
    v4i v = {0, 0, 0, 0};
    int part = 0;
    if (row < n) {
        v = *reinterpret_cast<const v4i*>(src + (size_t)row * 128 + ch * 16);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const unsigned u = (unsigned)v[w] ^ 0x80808080u;
            v[w] = (int)u;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int x = (int)(signed char)((u >> (8 * b)) & 0xff);
                part += x * x;
            }
        }
    }
    *reinterpret_cast<v4i*>(dst + (size_t)row * 128 + ch * 16) = v;
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    if (ch == 0) {
        if (mode == 0) {
            norm[row] = part;
        } else {
            const int c = (row < n) ? (-part * 256 + 255 - (row & (GROUP - 1))) : kNone;
            norm[row] = c >> 9;
            norm[n_pad + row] = c & 511;
        }
    }
}

// median of three; the backend selects v_med3_i32 for this min/max pattern
__device__ __forceinline__ int med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

struct Top2 {
    int b1, i1, b2;  // best s, its reference index, second-best s (kNone = none)
};

// Pseudo-key that stands for "the global second best" inside a group: any real
// key of this group that could change the global top-2 is larger than it.
__device__ __forceinline__ int floor_key(const Top2& t) { return (t.b2 == kNone) ? kNone : ((t.b2 << 8) | 255); }

__device__ __forceinline__ void fold_group(Top2& t, int& m1, int& m2, int gbase) {
    const int F = floor_key(t);
    if (m1 > F) {
        const int s1 = m1 >> 8;
        const int idx = gbase + 255 - (m1 & 255);
        const int s2 = (m2 > F) ? (m2 >> 8) : kNone;
        if (s1 > t.b1) {
            t.b2 = max(t.b1, s2);
            t.b1 = s1;
            t.i1 = idx;
        } else {
            t.b2 = max(t.b2, s1);
        }
    }
    m1 = m2 = floor_key(t);
}

// The accumulators start from ci = c_ref >> 9, so an MFMA result is
// dot + ci = key >> 9.  A tile can only change a query's top-2 if one of its 16
// values reaches thr = m2 >> 9; that test (8 v_max3 per tile) replaces the
// 3-instructions-per-value update, which runs only for the rare tiles that pass.
template <int NB>
__global__ __launch_bounds__(256) void match_kernel(const int8_t* __restrict__ q, const int* __restrict__ qnorm,
                                                    int nq, const int8_t* __restrict__ r,
                                                    const int* __restrict__ rc, int nr, int nr_pad, float ratio,
                                                    int ratio_form, int* __restrict__ best, int* __restrict__ d1o,
                                                    int* __restrict__ d2o) {
    // two LDS stages of RT references: [bytes RT*128][ci RT][lo RT]
    constexpr int STAGE = RT * 128 + RT * 8;
    __shared__ __attribute__((aligned(1024))) unsigned char s_all[2 * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jq = lane & 31, h = lane >> 5;
    constexpr int QPW = 32 * NB;  // queries per wave
    const int qbase = blockIdx.x * (4 * QPW) + wave * QPW;

    // this wave's queries stay in registers as MFMA B operands for the whole sweep
    v4i bq[NB][4];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int8_t* row = q + (size_t)(qbase + b * 32 + jq) * 128;
#pragma unroll
        for (int s = 0; s < 4; s++) bq[b][s] = *reinterpret_cast<const v4i*>(row + 32 * s + 16 * h);
    }
    Top2 top[NB];
    int m1[NB], m2[NB], thr[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        top[b] = Top2{kNone, -1, kNone};
        m1[b] = kNone;
        m2[b] = kNone;
        thr[b] = kNone >> 9;
    }

    // Stage loads go global -> LDS directly (no registers, no ds_write).  One wave
    // instruction fills 1 KB = 8 reference rows; lane i lands at row i>>3, 16-byte
    // slot i&7, so the XOR swizzle that makes the operand reads conflict-free is
    // applied to the SOURCE chunk each lane fetches.
    constexpr int NPIECE = RT / 8 / 4;  // 1-KB pieces per wave and stage
    auto fetch = [&](int st, int buf) {
        unsigned char* base = s_all + buf * STAGE;
#pragma unroll
        for (int u = 0; u < NPIECE; u++) {
            const int piece = wave * NPIECE + u;
            const int row = piece * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((row >> 1) & 7);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(r + (size_t)(st + row) * 128 + ch * 16),
                (__attribute__((address_space(3))) void*)(base + piece * 1024), 16, 0, 0);
        }
        // the per-reference constants (ci then lo, RT ints each) the same way, 64 ints
        // per wave instruction, so that the loop holds no register-destination load
        // whose wait would also drain the DMA
        constexpr int NCI = 2 * RT / 64;  // wave instructions for ci + lo
#pragma unroll
        for (int u = 0; u < (NCI + 3) / 4; u++) {
            const int k = wave + 4 * u;  // wave-uniform
            if (k < NCI) {
                const int* src = (k < NCI / 2) ? (rc + st + k * 64) : (rc + nr_pad + st + (k - NCI / 2) * 64);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane),
                                                 (__attribute__((address_space(3))) void*)(base + RT * 128 + k * 256),
                                                 4, 0, 0);
            }
        }
    };
    fetch(0, 0);
    __syncthreads();
    int buf = 0;
    for (int st = 0; st < nr_pad; st += RT, buf ^= 1) {
        const unsigned char* s_ref = s_all + buf * STAGE;
        const int* s_ci = reinterpret_cast<const int*>(s_ref + RT * 128);
        const int* s_lo = s_ci + RT;
        if (st + RT < nr_pad) fetch(st + RT, buf ^ 1);

        // Fast pass over the stage's tiles, branch-free so that LDS reads, MFMA
        // chains and the threshold tests of neighbouring tiles overlap: it only
        // records WHICH tiles could change some query's top-2.  thr is not raised
        // inside the pass (a stale, lower threshold only flags more tiles).
        unsigned hotmask = 0;
        // operands of tile t+1 are read from LDS before the MFMAs of tile t are issued
        auto load_tile = [&](int tile, v4i (&a)[4], v16i& ci) {
            const int row = tile * 32 + jq;
#pragma unroll
            for (int s = 0; s < 4; s++)
                a[s] = *reinterpret_cast<const v4i*>(s_ref + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4));
            // accumulator register e holds reference row (e&3) + 8*(e>>2) + 4*h of the tile
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const v4i t = *reinterpret_cast<const v4i*>(&s_ci[tile * 32 + 8 * g + 4 * h]);
                ci[4 * g + 0] = t[0];
                ci[4 * g + 1] = t[1];
                ci[4 * g + 2] = t[2];
                ci[4 * g + 3] = t[3];
            }
        };
        v4i a_cur[4], a_nxt[4];
        v16i ci_cur, ci_nxt;
        load_tile(0, a_cur, ci_cur);
#pragma unroll
        for (int tile = 0; tile < RT / 32; tile++) {
            if (tile + 1 < RT / 32) load_tile(tile + 1, a_nxt, ci_nxt);
            // the threshold test of chain b is issued after the MFMAs of chain b+1, so
            // the wave never idles on a chain's result while the matrix pipe is free
            auto chain = [&](int b) {
                v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_cur[0], bq[b][0], ci_cur, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 4; s++)
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_cur[s], bq[b][s], acc, 0, 0, 0);
                return acc;
            };
            auto test = [&](const v16i& acc, int b) {
                int mx = max(max(acc[0], acc[1]), acc[2]);
#pragma unroll
                for (int e = 3; e + 1 < 16; e += 2) mx = max(max(mx, acc[e]), acc[e + 1]);
                mx = max(mx, acc[15]);
                // one flag per (tile, chain): the slow pass redoes only the 32 x 32 blocks
                // that can change a top-2, not all NB chains of a flagged tile
                hotmask |= (__ballot(mx >= thr[b]) != 0ull) ? (1u << (tile * NB + b)) : 0u;
            };
            v16i acc_prev = chain(0);
#pragma unroll
            for (int b = 1; b < NB; b++) {
                const v16i acc_now = chain(b);
                __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs ahead of the previous chain's test
                test(acc_prev, b - 1);
                acc_prev = acc_now;
            }
            test(acc_prev, NB - 1);
            if (tile + 1 < RT / 32) {
#pragma unroll
                for (int s = 0; s < 4; s++) a_cur[s] = a_nxt[s];
                ci_cur = ci_nxt;
            }
        }
        // Slow pass (rare once a few thousand references are in): redo the flagged
        // tiles and run the exact top-2 update on every value.
        if (hotmask) {
            for (int tile = 0; tile < RT / 32; tile++) {
                const unsigned tb = (hotmask >> (tile * NB)) & ((1u << NB) - 1u);
                if (!tb) continue;
                const int row = tile * 32 + jq;
                v4i a[4];
#pragma unroll
                for (int s = 0; s < 4; s++)
                    a[s] = *reinterpret_cast<const v4i*>(s_ref + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4));
                v16i ci;
                v4i lo[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const v4i t = *reinterpret_cast<const v4i*>(&s_ci[tile * 32 + 8 * g + 4 * h]);
                    lo[g] = *reinterpret_cast<const v4i*>(&s_lo[tile * 32 + 8 * g + 4 * h]);
                    ci[4 * g + 0] = t[0];
                    ci[4 * g + 1] = t[1];
                    ci[4 * g + 2] = t[2];
                    ci[4 * g + 3] = t[3];
                }
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if (!((tb >> b) & 1u)) continue;  // wave-uniform
                    v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], bq[b][0], ci, 0, 0, 0);
#pragma unroll
                    for (int s = 1; s < 4; s++)
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[b][s], acc, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int key = (acc[e] << 9) + lo[e >> 2][e & 3];
                        m2[b] = med3i(m1[b], m2[b], key);
                        m1[b] = max(m1[b], key);
                    }
                    thr[b] = m2[b] >> 9;
                }
            }
        }
        if (((st + RT) & (GROUP - 1)) == 0) {
            const int gbase = (st + RT) - GROUP;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                fold_group(top[b], m1[b], m2[b], gbase);
                thr[b] = m2[b] >> 9;
            }
        }
        __syncthreads();
    }
    if (nr_pad & (GROUP - 1)) {
        const int gbase = nr_pad & ~(GROUP - 1);
#pragma unroll
        for (int b = 0; b < NB; b++) fold_group(top[b], m1[b], m2[b], gbase);
    }

    // the two half-waves hold disjoint reference rows of the same queries: merge
#pragma unroll
    for (int b = 0; b < NB; b++) {
        Top2 a = top[b];
        Top2 o;
        o.b1 = __shfl_xor(a.b1, 32);
        o.i1 = __shfl_xor(a.i1, 32);
        o.b2 = __shfl_xor(a.b2, 32);
        if (o.b1 > a.b1 || (o.b1 == a.b1 && o.b1 != kNone && o.i1 < a.i1)) {
            const Top2 t = a;
            a = o;
            o = t;
        }
        a.b2 = max(a.b2, o.b1);
        const int qi = qbase + b * 32 + jq;
        if (h == 0 && qi < nq) {
            const int nqv = qnorm[qi];
            const int dd1 = (a.b1 == kNone) ? INT32_MAX : nqv - a.b1;
            const int dd2 = (a.b2 == kNone) ? INT32_MAX : nqv - a.b2;
            const float f1 = (dd1 == INT32_MAX) ? 3.402823466e+38f : (float)dd1;
            const float f2 = (dd2 == INT32_MAX) ? 3.402823466e+38f : (float)dd2;
            bool ok;
            if (ratio_form == 1) ok = (f1 / f2) < ratio;
            else ok = (nr >= 2) && (f1 < f2 * ratio);
            best[qi] = (ok && a.b1 != kNone) ? a.i1 : -1;
            if (d1o) d1o[qi] = dd1;
            if (d2o) d2o[qi] = dd2;
        }
    }
}

// Launch sequence shared by both entry-point families: the int8 copies, norms and
// per-reference constants go to the four work areas the caller names.
static void launch_match(hipStream_t st, bool big, const uint8_t* d_q, int nq, int nq_pad, const uint8_t* d_r, int nr,
                         int nr_pad, int8_t* qi8, int* qnorm, int8_t* ri8, int* rconst, float ratio, int ratio_form,
                         int32_t* d_best, int32_t* d_d1, int32_t* d_d2) {
    const int qpb = big ? 512 : 256;
    hipLaunchKernelGGL(match_prep_kernel, dim3((nq_pad * 8 + 255) / 256), dim3(256), 0, st, d_q, nq, nq_pad, qi8, qnorm,
                       0);
    if (nr_pad > 0)
        hipLaunchKernelGGL(match_prep_kernel, dim3((nr_pad * 8 + 255) / 256), dim3(256), 0, st, d_r, nr, nr_pad, ri8,
                           rconst, 1);
    if (big)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(match_kernel<4>), dim3(nq_pad / qpb), dim3(256), 0, st, (const int8_t*)qi8,
                           (const int*)qnorm, nq, (const int8_t*)ri8, (const int*)rconst, nr, nr_pad, ratio,
                           ratio_form, d_best, d_d1, d_d2);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(match_kernel<2>), dim3(nq_pad / qpb), dim3(256), 0, st, (const int8_t*)qi8,
                           (const int*)qnorm, nq, (const int8_t*)ri8, (const int*)rconst, nr, nr_pad, ratio,
                           ratio_form, d_best, d_d1, d_d2);
}

static bool match_big(const sift_hip_ctx* ctx, int nq) {
    if (ctx->tune.match_nb) return ctx->tune.match_nb == 4;
    return nq >= 512 * 512;  // >= 2 workgroups of 512 queries per CU
}

int run_match(sift_hip_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, float ratio, int ratio_form,
              int32_t* d_best, int32_t* d_d1, int32_t* d_d2) {
    const bool big = match_big(ctx, nq);
    const int qpb = big ? 512 : 256;
    const int nq_pad = (nq + qpb - 1) / qpb * qpb;
    const int nr_pad = (nr + RT - 1) / RT * RT;
    int rc;
    if ((rc = ensure(ctx, ctx->qi8, (size_t)nq_pad * 128))) return rc;
    if ((rc = ensure(ctx, ctx->ri8, (size_t)std::max(nr_pad, RT) * 128))) return rc;
    if ((rc = ensure(ctx, ctx->qnorm, (size_t)nq_pad * 4))) return rc;
    if ((rc = ensure(ctx, ctx->rconst, (size_t)std::max(nr_pad, RT) * 8))) return rc;
    ProfScope ps(ctx, SIFT_HIP_K_MATCH, ctx->stream);
    launch_match(ctx->stream, big, d_q, nq, nq_pad, d_r, nr, nr_pad, (int8_t*)ctx->qi8.p, (int*)ctx->qnorm.p,
                 (int8_t*)ctx->ri8.p, (int*)ctx->rconst.p, ratio, ratio_form, d_best, d_d1, d_d2);
    return SIFT_HIP_OK;
}

// ---- the matcher for callers that cannot allocate ---------------------------------
// Its work areas (int8 copies of both descriptor sets, query norms, reference
// constants) live in one file-scope device array instead of buffers of the context.
// Each GPU holds its own copy of the array at its own address; `scratch` itself is
// only the host-side handle of the symbol, so the address is asked for once per
// device (match_scratch_init, with that device current) and only that pointer is
// ever handed to kernels.
constexpr int kNoAllocMax = SIFT_HIP_NOALLOC_MAX;                      // rows per side (a multiple of 512)
constexpr size_t kScrQ = 0;                                            // int8 queries
constexpr size_t kScrR = kScrQ + (size_t)kNoAllocMax * 128;            // int8 references
constexpr size_t kScrQn = kScrR + (size_t)kNoAllocMax * 128;           // query norms
constexpr size_t kScrRc = kScrQn + (size_t)kNoAllocMax * 4;            // reference constants (two ints per row)
constexpr size_t kScrBytes = kScrRc + (size_t)kNoAllocMax * 8;
static_assert(kNoAllocMax % 512 == 0 && kNoAllocMax >= RT && kNoAllocMax % RT == 0, "padding must stay inside the array");
static __device__ __align__(256) unsigned char scratch[kScrBytes];

static unsigned char* g_scratch_ptr[64];  // per device: address of that device's copy of `scratch`
static std::mutex g_scratch_mutex;

int match_scratch_init(sift_hip_ctx* ctx) {
    const int dev = ctx->device;
    if (dev < 0 || dev >= 64) return SIFT_HIP_ERR_LIMIT;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        if (!hip_ok(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)), "address of the matcher's static work area") || !p)
            return SIFT_HIP_ERR_HIP;
        g_scratch_ptr[dev] = (unsigned char*)p;
    }
    ctx->match_scratch = g_scratch_ptr[dev];
    return SIFT_HIP_OK;
}

int run_match_noalloc(sift_hip_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, float ratio,
                      int ratio_form, int32_t* d_best, int32_t* d_d1, int32_t* d_d2) {
    if (nq > kNoAllocMax || nr > kNoAllocMax) {
        ctx->last_error = "sift_hip_match_u8_noalloc takes at most " + std::to_string(kNoAllocMax) +
                          " queries and as many references";
        return SIFT_HIP_ERR_LIMIT;
    }
    unsigned char* w = ctx->match_scratch;
    if (!w) return SIFT_HIP_ERR_HIP;
    const bool big = false;  // 256-query workgroups: at most 256 of them at the size limit
    const int nq_pad = (nq + 255) / 256 * 256;
    const int nr_pad = (nr + RT - 1) / RT * RT;
    // Every row the kernels read (up to nq_pad / nr_pad) is rewritten by the two
    // preparation launches, padding rows included, so nothing of an earlier call's
    // contents is ever looked at and nothing has to be cleared.
    ProfScope ps(ctx, SIFT_HIP_K_MATCH, ctx->stream);  // does nothing unless profiling is on (which creates events)
    launch_match(ctx->stream, big, d_q, nq, nq_pad, d_r, nr, nr_pad, (int8_t*)(w + kScrQ), (int*)(w + kScrQn),
                 (int8_t*)(w + kScrR), (int*)(w + kScrRc), ratio, ratio_form, d_best, d_d1, d_d2);
    return SIFT_HIP_OK;
}

// ---- descriptors kept on the device (sift_hip_keep_descriptors) -------------------------------------
// The 128 descriptor bytes of each record, packed: one 16-byte piece per thread.
__global__ __launch_bounds__(256) void gather_descr_kernel(const sift_hip_feat* __restrict__ feat, int n,
                                                           uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * 8) return;
    const size_t f = i >> 3;
    const int piece = (int)(i & 7);
    const uint8_t* src = feat[f].descr + piece * 16;  // records are 4-byte aligned
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
    uint4 v = make_uint4(s4[0], s4[1], s4[2], s4[3]);
    *reinterpret_cast<uint4*>(out + f * 128 + piece * 16) = v;
}

void launch_gather_descr(sift_hip_ctx* ctx, hipStream_t st, const sift_hip_feat* d_feat, int n, uint8_t* d_out) {
    if (n <= 0) return;
    const size_t threads = (size_t)n * 8;
    gather_descr_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(d_feat, n, d_out);
}

}  // namespace sifthip
