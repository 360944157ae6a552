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
// s = 2 a'.b' - |b'|^2 per query.  Per MFMA output element the epilogue is three
// VALU instructions: key = (dot << 9) + c_ref (c_ref = -|b'|^2 * 256 + 255 - slot,
// slot = reference index mod 256), m2 = med3(m1, m2, key), m1 = max(m1, key); the
// low byte makes the lowest index win ties and is decoded every 256 references.
#include "internal.h"

namespace sifthip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kNone = (int)0x80000000;  // "no key yet"
constexpr int RT = 128;                 // references per LDS stage
constexpr int NB = 2;                   // 32-query blocks per wave
constexpr int QPW = 32 * NB;            // queries per wave
constexpr int QPB = 4 * QPW;            // queries per workgroup
constexpr int GROUP = 256;              // references per key-decoding group

// bytes -> int8 (x - 128), row norms; mode 0: norm[i] = |row|^2 (queries),
// mode 1: norm[i] = c_ref (references).  Rows >= n are zero padding.
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
        if (mode == 0) norm[row] = part;
        else norm[row] = (row < n) ? (-part * 256 + 255 - (row & (GROUP - 1))) : kNone;
    }
}

// median of three; the backend selects v_med3_i32 for this min/max pattern
__device__ __forceinline__ int med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

struct Top2 {
    int b1, i1, b2;
};

__device__ __forceinline__ void fold_group(Top2& t, int& m1, int& m2, int gbase) {
    if (m1 != kNone) {
        const int s1 = m1 >> 8;
        const int idx = gbase + 255 - (m1 & 255);
        const int s2 = (m2 == kNone) ? kNone : (m2 >> 8);
        if (s1 > t.b1) {
            t.b2 = max(t.b1, s2);
            t.b1 = s1;
            t.i1 = idx;
        } else {
            t.b2 = max(t.b2, s1);
        }
    }
    m1 = kNone;
    m2 = kNone;
}

__global__ __launch_bounds__(256) void match_kernel(const int8_t* __restrict__ q, const int* __restrict__ qnorm,
                                                    int nq, const int8_t* __restrict__ r,
                                                    const int* __restrict__ rc, int nr, int nr_pad, float ratio,
                                                    int ratio_form, int* __restrict__ best, int* __restrict__ d1o,
                                                    int* __restrict__ d2o) {
    __shared__ __attribute__((aligned(16))) unsigned char s_ref[RT * 128];
    __shared__ __attribute__((aligned(16))) int s_c[RT];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int jq = lane & 31, h = lane >> 5;
    const int qbase = blockIdx.x * QPB + wave * QPW;

    // this wave's queries stay in registers as MFMA B operands for the whole sweep
    v4i bq[NB][4];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int8_t* row = q + (size_t)(qbase + b * 32 + jq) * 128;
#pragma unroll
        for (int s = 0; s < 4; s++) bq[b][s] = *reinterpret_cast<const v4i*>(row + 32 * s + 16 * h);
    }
    Top2 top[NB];
    int m1[NB], m2[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        top[b] = Top2{kNone, -1, kNone};
        m1[b] = kNone;
        m2[b] = kNone;
    }

    for (int st = 0; st < nr_pad; st += RT) {
        // stage RT references: 16-byte chunks, XOR-swizzled so that the operand
        // reads below (32 rows x one chunk column) are bank-conflict free
#pragma unroll
        for (int u = 0; u < (RT * 8) / 256; u++) {
            const int t = tid + 256 * u;
            const int row = t >> 3, ch = t & 7;
            const v4i v = *reinterpret_cast<const v4i*>(r + (size_t)(st + row) * 128 + ch * 16);
            *reinterpret_cast<v4i*>(s_ref + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = v;
        }
        if (tid < RT) s_c[tid] = rc[st + tid];
        __syncthreads();
#pragma unroll
        for (int tile = 0; tile < RT / 32; tile++) {
            const int row = tile * 32 + jq;
            v4i a[4];
#pragma unroll
            for (int s = 0; s < 4; s++)
                a[s] = *reinterpret_cast<const v4i*>(s_ref + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4));
            v4i cv[4];
#pragma unroll
            for (int g = 0; g < 4; g++) cv[g] = *reinterpret_cast<const v4i*>(&s_c[tile * 32 + 8 * g + 4 * h]);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[b][s], acc, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    // accumulator register e holds reference row (e&3) + 8*(e>>2) + 4*h of the tile
                    const int key = (acc[e] << 9) + cv[e >> 2][e & 3];
                    m2[b] = med3i(m1[b], m2[b], key);
                    m1[b] = max(m1[b], key);
                }
            }
        }
        if (((st + RT) & (GROUP - 1)) == 0) {
            const int gbase = (st + RT) - GROUP;
#pragma unroll
            for (int b = 0; b < NB; b++) fold_group(top[b], m1[b], m2[b], gbase);
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

int run_match(sift_hip_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, float ratio, int ratio_form,
              int32_t* d_best, int32_t* d_d1, int32_t* d_d2) {
    const int nq_pad = (nq + QPB - 1) / QPB * QPB;
    const int nr_pad = (nr + RT - 1) / RT * RT;
    int rc;
    if ((rc = ensure(ctx, ctx->qi8, (size_t)nq_pad * 128))) return rc;
    if ((rc = ensure(ctx, ctx->ri8, (size_t)nr_pad * 128))) return rc;
    if ((rc = ensure(ctx, ctx->qnorm, (size_t)nq_pad * 4))) return rc;
    if ((rc = ensure(ctx, ctx->rconst, (size_t)nr_pad * 4))) return rc;
    ProfScope ps(ctx, SIFT_HIP_K_MATCH, ctx->stream);
    hipLaunchKernelGGL(match_prep_kernel, dim3((nq_pad * 8 + 255) / 256), dim3(256), 0, ctx->stream, d_q, nq, nq_pad,
                       (int8_t*)ctx->qi8.p, (int*)ctx->qnorm.p, 0);
    hipLaunchKernelGGL(match_prep_kernel, dim3((nr_pad * 8 + 255) / 256), dim3(256), 0, ctx->stream, d_r, nr, nr_pad,
                       (int8_t*)ctx->ri8.p, (int*)ctx->rconst.p, 1);
    hipLaunchKernelGGL(match_kernel, dim3(nq_pad / QPB), dim3(256), 0, ctx->stream, (const int8_t*)ctx->qi8.p,
                       (const int*)ctx->qnorm.p, nq, (const int8_t*)ctx->ri8.p, (const int*)ctx->rconst.p, nr, nr_pad,
                       ratio, ratio_form, d_best, d_d1, d_d2);
    return SIFT_HIP_OK;
}

}  // namespace sifthip
