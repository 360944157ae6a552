// Scale-space kernels for gfx950: gray conversion, bilinear octave resample,
// separable Gaussian blur (row pass, column pass).  DoG layers are not stored by
// the extract path; the detector subtracts adjacent levels on the fly.
//
// Numerics follow the reference exactly in blur_mode 0: every tap is a float
// product rounded to float, accumulated in double in ascending tap order, and the
// sum is rounded to float once per pass (imglib/imglib.cpp:85-117).  That makes
// the blur VALU-bound (one v_mul_f32, one v_cvt_f64_f32, one v_add_f64 per tap)
// rather than HBM-bound; the kernels therefore never recompute a tap: rows and
// columns run as two launches with the intermediate image in HBM / Infinity Cache.
// Each thread produces P consecutive outputs ALONG the filter direction and keeps
// a sliding window of P inputs in registers, so a tap step costs one LDS read and
// one scalar tap load for P multiply-accumulates.
#include <algorithm>
#include <cstdlib>

#include "internal.h"

namespace sifthip {

// ---- ImgRaw::ConverGray, Imgraw.cpp:257-269 --------------------------------------
__global__ __launch_bounds__(256) void gray_kernel(const float* __restrict__ rgb, float* __restrict__ gray, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float R = rgb[i * 3 + 0], G = rgb[i * 3 + 1], B = rgb[i * 3 + 2];
    // double products and sums, left to right, one rounding to float at the end
    double v = (double)R * 0.299;
    v = v + (double)G * 0.587;
    v = v + (double)B * 0.114;
    gray[i] = (float)v;
}

void launch_gray(sift_hip_ctx* ctx, hipStream_t st, const float* rgb, float* gray, size_t n) {
    ProfScope ps(ctx, SIFT_HIP_K_GRAY, st);
    hipLaunchKernelGGL(gray_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rgb, gray, n);
}

// ---- Scaling::first + Scaling::bilinear, imglib.cpp:203-252 ----------------------
// Source coordinate of destination index i along an axis of `n` source and `nn`
// destination samples, and the bilinear terms derived from it (floor/ceil
// neighbours, the neighbour index clamped where the reference reads past the end).
struct Axis {
    int i0, i1;
    float d1, d2;
};
__device__ __forceinline__ Axis first_axis(int i, int n, int nn, float ratio) {
    float src;
    if (ratio < 1) {
        // ((i+0.5f)/Ratio) is a float quotient; the "- 0.5" is done in double
        src = (float)((double)(((float)i + 0.5f) / ratio) - 0.5);
    } else {
        src = (float)i * ((float)n - 1.f) / ((float)nn - 1.f);
    }
    const float f0 = floorf(src);
    Axis a;
    a.i0 = (int)f0;
    a.i1 = min((int)ceilf(src), n - 1);
    a.d1 = src - f0;  // x - x0 with x0 converted back to float
    a.d2 = 1.f - a.d1;
    return a;
}

// A workgroup produces a 256 x FIRST_ROWS tile: every thread keeps the terms of its
// column, the terms of the tile's rows are computed once into LDS, so a pixel costs
// four loads, six float operations and a store (the two IEEE divisions per pixel
// of the direct form made the kernel VALU-bound at 2x its memory time).
constexpr int FIRST_ROWS = 16;
__global__ __launch_bounds__(256) void first_kernel(const float* __restrict__ src, int w, int h, float ratio,
                                                    float* __restrict__ dst, int nw, int nh) {
    __shared__ Axis s_row[FIRST_ROWS];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * FIRST_ROWS;
    if (threadIdx.x < FIRST_ROWS && j0 + threadIdx.x < nh) s_row[threadIdx.x] = first_axis(j0 + threadIdx.x, h, nh, ratio);
    __syncthreads();
    if (i >= nw) return;
    const Axis ax = first_axis(i, w, nw, ratio);
    const int rows = min(FIRST_ROWS, nh - j0);
#pragma unroll 4
    for (int r = 0; r < rows; r++) {
        const Axis ay = s_row[r];
        const float* r0 = src + (size_t)ay.i0 * w;
        const float* r1 = src + (size_t)ay.i1 * w;
        const float A = r0[ax.i0];
        const float B = r0[ax.i1];
        const float C = r1[ax.i0];
        const float D = r1[ax.i1];
        const float AB = A * ax.d2 + B * ax.d1;
        const float CD = C * ax.d2 + D * ax.d1;
        dst[(size_t)(j0 + r) * nw + i] = AB * ay.d2 + CD * ay.d1;
    }
}

void launch_first(sift_hip_ctx* ctx, hipStream_t st, const float* src, int w, int h, float ratio, float* dst, int nw,
                  int nh) {
    ProfScope ps(ctx, SIFT_HIP_K_RESAMPLE, st);
    dim3 grid((nw + 255) / 256, (nh + FIRST_ROWS - 1) / FIRST_ROWS);
    hipLaunchKernelGGL(first_kernel, grid, dim3(256), 0, st, src, w, h, ratio, dst, nw, nh);
}

constexpr int kMaxColLds = 64 * 1024;   // largest LDS tile a blur pass requests without asking for more
constexpr int kBigColLds = 150 * 1024;  // gfx950 lets a workgroup address 160 KB once the kernel opts in

// Workgroup trace for tools/microbench_sweep.hip (compiled in only with -DWG_TRACE): per workgroup its start and
// end on the shader clock and on the 100 MHz clock all compute units share, and where it ran.
#ifdef WG_TRACE
__device__ unsigned long long g_wg_trace[6 * 32768];
#define TRACE_BEGIN() const unsigned long long t_begin__ = __builtin_readcyclecounter(), r_begin__ = __builtin_amdgcn_s_memrealtime()
#define TRACE_END(wg__)                                                                                          \
    do {                                                                                                         \
        if (threadIdx.x == 0 && (unsigned)(wg__) < 32768u) {                                                     \
            unsigned long long* t__ = sifthip::g_wg_trace + 6 * (size_t)(wg__);                                  \
            t__[0] = t_begin__;                                                                                  \
            t__[1] = __builtin_readcyclecounter();                                                               \
            t__[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  /* HW_ID */                   \
            t__[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); /* XCC_ID */                  \
            t__[4] = r_begin__;                                                                                  \
            t__[5] = __builtin_amdgcn_s_memrealtime();                                                           \
        }                                                                                                        \
    } while (0)
#else
#define TRACE_BEGIN() ((void)0)
#define TRACE_END(wg__) ((void)0)
#endif

// ---- separable blur ----------------------------------------------------------------
// P = outputs per thread along the filter direction: 8 for large octaves (one LDS
// read per 8 MACs), 4 and 2 for smaller ones, which otherwise cannot fill 1024 SIMDs.

// The sliding window of a thread: P input values in a ring (slot = input index
// mod P) and one tap step over all of them: acc[p] += win[(p + u) % P] * g.
typedef float float_pair __attribute__((ext_vector_type(2)));
// Float accumulation (blur_mode 1): fused multiply-add, two outputs per instruction
// (v_pk_fma_f32).  Output pair (2j, 2j+1) needs the input pair (2j+k, 2j+k+1) at tap k:
// for even k that is a pair of ring slots (2m, 2m+1), for odd k a pair (2m+1, 2m+2), so the
// ring is kept in both pairings and every new input is written into one pair of each.
template <typename ACC, int P>
struct Window {
    float_pair e[P / 2], o[P / 2];
    __device__ __forceinline__ void set(int slot, float x) {
        if (slot & 1) {
            e[slot >> 1].y = x;
            o[slot >> 1].x = x;
        } else {
            e[slot >> 1].x = x;
            o[((slot + P - 2) % P) >> 1].y = x;
        }
    }
    __device__ __forceinline__ void step(ACC (&acc)[P], int u, float g) const {
        const float_pair gg = {g, g};
#pragma unroll
        for (int j = 0; j < P / 2; j++) {
            const float_pair in = (u & 1) ? o[(j + (u >> 1)) % (P / 2)] : e[(j + (u >> 1)) % (P / 2)];
            float_pair a = {acc[2 * j], acc[2 * j + 1]};
            a = __builtin_elementwise_fma(in, gg, a);
            acc[2 * j] = a.x;
            acc[2 * j + 1] = a.y;
        }
    }
};
// Exact mode: the product is rounded to float first, as `img[..] * gau_mat[k]` is,
// then added in double.  The slots are kept as register pairs so that one packed
// multiply (v_pk_mul_f32, same rounding as two v_mul_f32) serves two outputs;
// conversion and addition stay one instruction per output.
template <int P>
struct Window<double, P> {
    float_pair v[P / 2];
    __device__ __forceinline__ void set(int slot, float x) {
        if (slot & 1) v[slot >> 1].y = x; else v[slot >> 1].x = x;
    }
    __device__ __forceinline__ void step(double (&acc)[P], int u, float g) const {
#pragma unroll
        for (int i = 0; i < P / 2; i++) {
            const float_pair prod = v[i] * g;
            acc[(2 * i - u + P) % P] += (double)prod.x;      // slot s feeds output (s - u) mod P
            acc[(2 * i + 1 - u + P) % P] += (double)prod.y;
        }
    }
};

// K taps over P consecutive outputs in blocks of TB taps: the TB+P-1 inputs of the NEXT block are
// read (through rd(j) = input j of the run, j = 0 .. K+P-2; rd must tolerate j up to K+P+TB-2) before
// the current block is consumed, so that the read latency is paid once per TB taps, not once per
// tap as with a P-slot ring when P is small and few waves are resident.  Per output the products
// are still added in ascending tap order.
template <typename ACC, int P, int TB = 16, typename RD>
__device__ __forceinline__ void conv_blocked(RD rd, const float* __restrict__ taps, int K, ACC (&acc)[P]) {
    constexpr int NV = TB + P - 1;
#pragma unroll
    for (int p = 0; p < P; p++) acc[p] = 0;
    float v[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] = rd(j);
    int k = 0;
    for (; k + TB <= K; k += TB) {
        float vn[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) vn[j] = rd(k + TB + j);
#pragma unroll
        for (int t = 0; t < TB; t++) {
            const float g = taps[k + t];
#pragma unroll
            for (int p = 0; p < P; p++) {
                if constexpr (sizeof(ACC) == sizeof(double)) {
                    const float prod = v[t + p] * g;
                    acc[p] += (double)prod;
                } else {
                    acc[p] = fmaf(v[t + p], g, acc[p]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = vn[j];
    }
#pragma unroll
    for (int t = 0; t < TB; t++) {
        if (k + t < K) {  // wave-uniform
            const float g = taps[k + t];
#pragma unroll
            for (int p = 0; p < P; p++) {
                if constexpr (sizeof(ACC) == sizeof(double)) {
                    const float prod = v[t + p] * g;
                    acc[p] += (double)prod;
                } else {
                    acc[p] = fmaf(v[t + p], g, acc[p]);
                }
            }
        }
    }
}

// LDS index with one pad word per P so that lanes P floats apart hit distinct banks
// (lane stride P+1 words, odd).
template <int P>
__host__ __device__ __forceinline__ int padx(int i) { return i + i / P; }

// Row pass: one wave per 512-pixel run of one row.  LDS holds the run plus halo.
template <typename ACC, int P>
__global__ __launch_bounds__(64) void blur_row_kernel(const float* __restrict__ in, float* __restrict__ out, int W,
                                                      int H, const float* __restrict__ taps, int K) {
    extern __shared__ float s[];
    const int lane = threadIdx.x;
    TRACE_BEGIN();
    const int r = K >> 1;
    const int y = blockIdx.y;
    const int x0 = blockIdx.x * (64 * P);
    const int n_in = 64 * P + K - 1;
    const float* row = in + (size_t)y * W;
    if (x0 - r >= 0 && x0 - r + n_in <= W) {
        // interior run: no clamping; loads and LDS writes differ by constants only
        const float* src = row + (x0 - r) + lane;
        float* dst = s + padx<P>(lane);
        constexpr int step = 64 + 64 / P;  // padx(lane + 64) - padx(lane)
        int i = lane;
        for (; i + 64 * 3 < n_in; i += 64 * 4, src += 256, dst += 4 * step) {
            const float a = src[0], b = src[64], c = src[128], d = src[192];
            dst[0] = a;
            dst[step] = b;
            dst[2 * step] = c;
            dst[3 * step] = d;
        }
        for (; i < n_in; i += 64, src += 64, dst += step) dst[0] = src[0];
    } else {
        for (int i = lane; i < n_in; i += 64) {
            int gx = x0 + i - r;
            gx = max(0, min(gx, W - 1));
            s[padx<P>(i)] = row[gx];
        }
    }
    __syncthreads();
    const int base = lane * P;
    if (x0 + base >= W) return;
    ACC acc[P];
    {
        Window<ACC, P> win;
#pragma unroll
        for (int p = 0; p < P; p++) {
            acc[p] = 0;
            win.set(p, s[padx<P>(base + p)]);
        }
        int k = 0;
        for (; k + P <= K; k += P) {
#pragma unroll
            for (int u = 0; u < P; u++) {
                const float g = taps[k + u];
                win.step(acc, u, g);
                win.set(u, s[padx<P>(base + k + u + P)]);
            }
        }
#pragma unroll
        for (int u = 0; u < P; u++) {
            if (k + u < K) {  // wave-uniform
                const float g = taps[k + u];
                win.step(acc, u, g);
                win.set(u, s[padx<P>(base + k + u + P)]);
            }
        }
    }
    float* o = out + (size_t)y * W + x0 + base;
    if constexpr (P == 8) {
        if ((W & 3) == 0 && x0 + base + P <= W) {
            float4 a = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
            float4 b = make_float4((float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]);
            reinterpret_cast<float4*>(o)[0] = a;
            reinterpret_cast<float4*>(o)[1] = b;
            return;
        }
    }
    {
#pragma unroll
        for (int p = 0; p < P; p++)
            if (x0 + base + p < W) o[p] = (float)acc[p];
    }
    TRACE_END(blockIdx.y * gridDim.x + blockIdx.x);
}

// Row pass for kernels too long for an LDS run (deep octaves of very large images:
// thousands of taps over a few pixels): the window slides over global memory.
template <typename ACC, int P>
__global__ __launch_bounds__(64) void blur_row_global_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int W, int H, const float* __restrict__ taps, int K) {
    const int lane = threadIdx.x;
    const int r = K >> 1;
    const int y = blockIdx.y;
    const int xb = (blockIdx.x * 64 + lane) * P;
    if (xb >= W) return;
    const float* row = in + (size_t)y * W;
    auto rd = [&](int xx) -> float { return row[max(0, min(xx, W - 1))]; };
    ACC acc[P];
    Window<ACC, P> win;
#pragma unroll
    for (int p = 0; p < P; p++) {
        acc[p] = 0;
        win.set(p, rd(xb + p - r));
    }
    int k = 0;
    for (; k + P <= K; k += P) {
#pragma unroll
        for (int u = 0; u < P; u++) {
            const float g = taps[k + u];
            win.step(acc, u, g);
            win.set(u, rd(xb + k + u + P - r));
        }
    }
#pragma unroll
    for (int u = 0; u < P; u++) {
        if (k + u < K) {
            const float g = taps[k + u];
            win.step(acc, u, g);
            win.set(u, rd(xb + k + u + P - r));
        }
    }
#pragma unroll
    for (int p = 0; p < P; p++)
        if (xb + p < W) out[(size_t)y * W + xb + p] = (float)acc[p];
}

// Column pass: a 256-thread workgroup owns 64 columns x TH rows; LDS holds the
// TH + K - 1 input rows.  Each wave takes P-row chunks.
template <typename ACC, int P>
__global__ __launch_bounds__(256) void blur_col_kernel(const float* __restrict__ in, float* __restrict__ out, int W,
                                                       int H, const float* __restrict__ taps, int K, int TH) {
    extern __shared__ float s[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = K >> 1;
    // Tiles that share halo rows are vertical neighbours.  Workgroup ids are dealt round-robin to
    // the 8 XCDs (one L2 each): remapped so that one XCD owns a contiguous run of tiles in
    // column-major order, the K-1 halo rows of a tile are L2 hits instead of fabric reads.
    const int ntile = gridDim.x * gridDim.y;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = ntile >> 3, r8 = ntile & 7, xcd = bid & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tbx = lin / (int)gridDim.y, tby = lin - tbx * (int)gridDim.y;
    const int x = tbx * 64 + lane;
    const int xc = min(x, W - 1);
    const int y0 = tby * TH;
    const int rows_in = TH + K - 1;
    if (y0 - r >= 0 && y0 - r + rows_in <= H) {
        // interior tile: no clamping, four rows of loads in flight per wave
        const float* src = in + (size_t)(y0 - r + wave) * W + xc;
        float* dst = s + wave * 64 + lane;
        const size_t step = (size_t)4 * W;
        int i = wave;
        for (; i + 12 < rows_in; i += 16, src += 4 * step, dst += 4 * 256) {
            const float a = src[0], b = src[step], c = src[2 * step], d = src[3 * step];
            dst[0] = a;
            dst[256] = b;
            dst[512] = c;
            dst[768] = d;
        }
        for (; i < rows_in; i += 4, src += step, dst += 256) dst[0] = src[0];
    } else {
        for (int i = wave; i < rows_in; i += 4) {
            int gy = y0 + i - r;
            gy = max(0, min(gy, H - 1));
            s[i * 64 + lane] = in[(size_t)gy * W + xc];
        }
    }
    __syncthreads();
    for (int c = wave * P; c < TH; c += 4 * P) {
        if (y0 + c >= H) break;  // wave-uniform
        ACC acc[P];
        {
            Window<ACC, P> win;
#pragma unroll
            for (int p = 0; p < P; p++) {
                acc[p] = 0;
                win.set(p, s[(c + p) * 64 + lane]);
            }
            int k = 0;
            for (; k + P <= K; k += P) {
#pragma unroll
                for (int u = 0; u < P; u++) {
                    const float g = taps[k + u];
                    win.step(acc, u, g);
                    win.set(u, s[(c + k + u + P) * 64 + lane]);
                }
            }
#pragma unroll
            for (int u = 0; u < P; u++) {
                if (k + u < K) {
                    const float g = taps[k + u];
                    win.step(acc, u, g);
                    win.set(u, s[(c + k + u + P) * 64 + lane]);
                }
            }
        }
        if (x < W) {
#pragma unroll
            for (int p = 0; p < P; p++) {
                const int y = y0 + c + p;
                if (y < H) {
                    out[(size_t)y * W + x] = (float)acc[p];
                }
            }
        }
    }
}

// Column pass for kernels too long for an LDS tile: the window slides over rows
// read straight from global memory (L1/L2 serve the overlap between row chunks).
template <typename ACC, int P>
__global__ __launch_bounds__(256) void blur_col_global_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int W, int H, const float* __restrict__ taps, int K) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = K >> 1;
    const int x = blockIdx.x * 64 + lane;
    if (x >= W) return;
    const int yb = (blockIdx.y * 4 + wave) * P;
    if (yb >= H) return;
    auto rd = [&](int yy) -> float {
        yy = max(0, min(yy, H - 1));
        return in[(size_t)yy * W + x];
    };
    ACC acc[P];
    Window<ACC, P> win;
#pragma unroll
    for (int p = 0; p < P; p++) {
        acc[p] = 0;
        win.set(p, rd(yb + p - r));
    }
    int k = 0;
    for (; k + P <= K; k += P) {
#pragma unroll
        for (int u = 0; u < P; u++) {
            const float g = taps[k + u];
            win.step(acc, u, g);
            win.set(u, rd(yb + k + u + P - r));
        }
    }
#pragma unroll
    for (int u = 0; u < P; u++) {
        if (k + u < K) {
            const float g = taps[k + u];
            win.step(acc, u, g);
            win.set(u, rd(yb + k + u + P - r));
        }
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int y = yb + p;
        if (y < H) {
            out[(size_t)y * W + x] = (float)acc[p];
        }
    }
}


// Kernels longer than the image (deep octaves: a few thousand pixels under kernels of hundreds
// to thousands of taps): most taps read a clamped index, so instead of staging K-1+n clamped
// copies the whole row (column strip) goes to LDS once and the index is clamped on the LDS read.
template <typename ACC, int P>
__global__ __launch_bounds__(64) void blur_row_whole_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int W, int H, const float* __restrict__ taps, int K) {
    extern __shared__ float s[];  // the W pixels of the row
    const int lane = threadIdx.x;
    const int r = K >> 1;
    const int y = blockIdx.y;
    const float* row = in + (size_t)y * W;
    for (int i = lane; i < W; i += 64) s[i] = row[i];
    __syncthreads();
    const int xb = (blockIdx.x * 64 + lane) * P;
    if (xb >= W) return;
    auto rd = [&](int xx) -> float { return s[max(0, min(xx, W - 1))]; };
    ACC acc[P];
    conv_blocked<ACC, P>([&](int j) { return rd(xb - r + j); }, taps, K, acc);
#pragma unroll
    for (int p = 0; p < P; p++)
        if (xb + p < W) out[(size_t)y * W + xb + p] = (float)acc[p];
}

template <typename ACC, int P>
__global__ __launch_bounds__(256) void blur_col_whole_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int W, int H, const float* __restrict__ taps, int K) {
    extern __shared__ float s[];  // all H rows of the workgroup's 64 columns
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = K >> 1;
    const int x = blockIdx.x * 64 + lane;
    const int xc = min(x, W - 1);
    for (int i = wave; i < H; i += 4) s[i * 64 + lane] = in[(size_t)i * W + xc];
    __syncthreads();
    const int yb = (blockIdx.y * 4 + wave) * P;
    if (yb >= H) return;
    auto rd = [&](int yy) -> float { return s[max(0, min(yy, H - 1)) * 64 + lane]; };
    ACC acc[P];
    conv_blocked<ACC, P>([&](int j) { return rd(yb - r + j); }, taps, K, acc);
    if (x < W) {
#pragma unroll
        for (int p = 0; p < P; p++)
            if (yb + p < H) out[(size_t)(yb + p) * W + x] = (float)acc[p];
    }
}

// Images of a few thousand pixels under kernels of hundreds to thousands of taps (the deepest
// octaves): there are too few outputs to fill the chip whatever the mapping, so the time of a pass
// is the length of one output's chain of K ordered additions.  One output per thread keeps that
// chain as short as the arithmetic allows: per tap one product, one conversion, one addition, and
// reads that need no index arithmetic in the loop.
//   row pass: the run of 64 outputs plus its K-1 neighbours is staged with clamped indices once;
//             lane l then reads staged[l + j] (an immediate offset per tap);
//   column pass: a wave owns one output row, so the clamped source row of every tap is the same
//             for all lanes (scalar arithmetic) and the whole 64-column strip sits in LDS.
// One output per thread, inputs and taps both read a block of TB ahead (rd(j) = input j, tp(k) =
// tap k; both must tolerate indices up to K + 2*TB - 1): nothing in the loop waits for memory, so
// a pass takes the K ordered additions of its outputs and little more.
template <typename ACC, int TB, typename RD, typename TP>
__device__ __forceinline__ ACC conv_chain(RD rd, TP tp, int K, ACC acc = 0) {
    float va[TB], ga[TB], vb[TB], gb[TB];  // two blocks in turn: one is consumed while the other is fetched
    auto fetch = [&](float (&v)[TB], float (&g)[TB], int k) {
#pragma unroll
        for (int j = 0; j < TB; j++) {
            v[j] = rd(k + j);
            g[j] = tp(k + j);
        }
    };
    auto consume = [&](const float (&v)[TB], const float (&g)[TB], int n) {  // n = TB: the whole block
#pragma unroll
        for (int t = 0; t < TB; t++) {
            if (t < n) {  // wave-uniform
                if constexpr (sizeof(ACC) == sizeof(double)) {
                    const float prod = v[t] * g[t];
                    acc += (double)prod;
                } else {
                    acc = fmaf(v[t], g[t], acc);
                }
            }
        }
    };
    fetch(va, ga, 0);
    int k = 0;
    for (; k + 2 * TB <= K; k += 2 * TB) {
        fetch(vb, gb, k + TB);
        consume(va, ga, TB);
        fetch(va, ga, k + 2 * TB);
        consume(vb, gb, TB);
    }
    if (k + TB <= K) {
        fetch(vb, gb, k + TB);
        consume(va, ga, TB);
        consume(vb, gb, K - k - TB);
    } else {
        consume(va, ga, K - k);
    }
    return acc;
}

template <typename ACC>
__global__ __launch_bounds__(64) void blur_row_tiny_kernel(const float* __restrict__ in, float* __restrict__ out, int W,
                                                           int H, const float* __restrict__ taps, int K) {
    extern __shared__ float s[];  // the K taps (+ 64 of read-ahead slack), then the staged inputs of the middle run
    const int lane = threadIdx.x;
    const int r = K >> 1;
    const int y = blockIdx.y;
    const int x0 = blockIdx.x * 64;
    const int xmax = min(x0 + 63, W - 1);
    const float* row = in + (size_t)y * W;
    // Output x reads column x + k - r at tap k, clamped: for the 64 outputs of this wave the first `a`
    // taps all read column 0 and the taps from `c` on all read column W-1 (for these octaves nearly all
    // of them); only the run between them fetches staged inputs.  Same products, same order.
    const int a = min(max(r - xmax + 1, 0), K), c = min(max(W - 1 - x0 + r, a), K);
    float* s_taps = s;
    float* s_in = s + K + 64;
    for (int i = lane; i < K + 64; i += 64) s_taps[i] = i < K ? taps[i] : 0.f;
    const int n_in = (c - a) + 64 + 64;  // lane l reads staged[l + j], j < (c - a) + 64 with the read-ahead
    for (int i0 = lane; i0 < n_in; i0 += 64 * 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = row[max(0, min(x0 + a + i0 + 64 * u - r, W - 1))];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + 64 * u < n_in) s_in[i0 + 64 * u] = t[u];
    }
    const float left = row[0], right = row[W - 1];
    __syncthreads();
    ACC acc = conv_chain<ACC, 32>([&](int) { return left; }, [&](int k) { return s_taps[k]; }, a);
    acc = conv_chain<ACC, 32>([&](int j) { return s_in[lane + j]; }, [&](int k) { return s_taps[a + k]; }, c - a, acc);
    acc = conv_chain<ACC, 32>([&](int) { return right; }, [&](int k) { return s_taps[c + k]; }, K - c, acc);
    if (x0 + lane < W) out[(size_t)y * W + x0 + lane] = (float)acc;
}

template <typename ACC>
__global__ __launch_bounds__(256) void blur_col_tiny_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int W, int H, const float* __restrict__ taps, int K) {
    extern __shared__ float s[];  // all H rows of the workgroup's 64 columns, then the K taps (+ 64)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = K >> 1;
    const int x = blockIdx.x * 64 + lane;
    const int xc = min(x, W - 1);
    for (int i0 = wave; i0 < H; i0 += 4 * 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = in[(size_t)min(i0 + 4 * u, H - 1) * W + xc];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + 4 * u < H) s[(i0 + 4 * u) * 64 + lane] = t[u];
    }
    float* s_taps = s + (size_t)H * 64;
    for (int i = threadIdx.x; i < K + 64; i += 256) s_taps[i] = i < K ? taps[i] : 0.f;
    __syncthreads();
    const int yb = blockIdx.y * 4 + wave;  // this wave's output row
    if (yb >= H) return;
    // Tap k reads row yb - r + k, clamped: the first `a` taps all read row 0 and the taps from `c` on
    // all read row H-1 (for these octaves that is nearly all of them), so those two runs multiply one
    // value by successive taps and only the short middle run fetches rows.  Same products, same order.
    const int a = min(max(r - yb + 1, 0), K), c = min(max(r - yb + H - 1, a), K);
    const float top = s[lane], bot = s[(H - 1) * 64 + lane];
    ACC acc = conv_chain<ACC, 32>([&](int) { return top; }, [&](int k) { return s_taps[k]; }, a);
    acc = conv_chain<ACC, 32>([&](int j) { return s[min(yb - r + a + j, H - 1) * 64 + lane]; },
                              [&](int k) { return s_taps[a + k]; }, c - a, acc);
    acc = conv_chain<ACC, 32>([&](int) { return bot; }, [&](int k) { return s_taps[c + k]; }, K - c, acc);
    if (x < W) out[(size_t)yb * W + x] = (float)acc;
}

// ---- exact column pass for short kernels: input-stationary ------------------------
// The taps are symmetric bit for bit (g[k] == g[K-1-k]: the kernel is built from its
// centre outwards), so the float product of an input with tap m serves two outputs,
// and so does its conversion to double: (K+1)/2 multiplies and conversions per pixel
// instead of K, the K additions staying.  Every output must still receive its
// addends in ascending tap order, which for output row o is ascending input row
// (addend k comes from row o + k - r).  A thread therefore walks down ONE column,
// input row by input row, and keeps the K outputs that are in flight (rows iv-r ..
// iv+r around input row iv) as double accumulators in registers: input iv adds
// q[r-|d|] = (double)(float)(x * g[r-|d|]) to output iv+d for d = -r..r, which
// completes output iv-r.  The accumulators form a ring (slot = row mod K); the loop
// body covers K input rows, so every slot index in it is a compile-time constant.
// No LDS: a wave reads 64 consecutive floats of each input row once (the next K
// rows are fetched while the current K are consumed) and writes 64 floats per row.
// A column is cut into segments of `seg` output rows, one wave each; a segment
// first consumes the K-1 rows above it (its share of redundant work).

// Buffer resource over `bytes` bytes at p (raw 32-bit elements, bounds-checked).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// Rows fetched ahead of their use: a ring of L registers with L a divisor of K (slot = step mod L
// stays a compile-time constant in a body of K steps), long enough to cover the memory latency.
template <int K>
struct StreamGeom {
    static constexpr int L = K == 55 ? 11 : K == 39 ? 13 : K == 21 ? 7 : K;
    static constexpr int WAVES = K <= 11 ? 8 : K <= 21 ? 5 : K <= 39 ? 3 : 2;  // waves a SIMD is to hold (bounds the register count)
};

template <int K>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(StreamGeom<K>::WAVES, 8))) void blur_col_stream_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int W, int H, const float* __restrict__ taps, int seg) {
    constexpr int R = K / 2, L = StreamGeom<K>::L;
    const int lane = threadIdx.x;
    const int x = blockIdx.x * 64 + lane;
    const int xc = min(x, W - 1);  // lanes past the edge redo the last column and store nothing
    const int y0 = blockIdx.y * seg;
    const int y1 = min(y0 + seg, H);
    const int nb = (y1 - y0 + K - 1) / K;  // the segment as blocks of K output rows (the last one may hang over y1)
    float g[R + 1];
#pragma unroll
    for (int m = 0; m <= R; m++) g[m] = taps[m];
    double acc[K];
#pragma unroll
    for (int j = 0; j < K; j++) acc[j] = 0.0;
    // Buffer addressing: one resource per wave over exactly the rows the segment may
    // read (resp. write), a lane offset and a wave-uniform row offset per access, all
    // below 2^31 whatever the image size.  Loads clamp the row offset to the resource
    // (clamp to edge); stores are never branched around: a row outside the segment or
    // a lane past the right edge yields an offset outside the resource, which the
    // hardware drops.
    const int yb = max(0, y0 - R);                   // first row the segment reads
    const int rows_in = min(H, y1 + R) - yb;         // rows it may read
    const __amdgpu_buffer_rsrc_t rin = buffer_of(in + (size_t)yb * W, (unsigned)((size_t)rows_in * W * 4));
    const __amdgpu_buffer_rsrc_t rout = buffer_of(out + (size_t)y0 * W, (unsigned)((size_t)(y1 - y0) * W * 4));
    const int row_bytes = W * 4;
    const int voff = xc * 4;
    const int last_in = (rows_in - 1) * row_bytes;
    // rows above the image clamp to row 0, rows below it to row H-1: both are ends of the resource
    // whenever the clamp can apply (yb == 0 resp. yb + rows_in == H)
    int in_off = (y0 - R - yb) * row_bytes;  // offset of the next input row to fetch
    auto load_next = [&]() __attribute__((always_inline)) -> float {
        const int off = max(0, min(in_off, last_in));
        in_off += row_bytes;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, voff, off, 0));
    };
    unsigned st_off = x < W ? (unsigned)voff : 0x80000000u;  // byte offset of the next output row to complete
    float cur[L];  // input row of step s sits in cur[s mod L]
#pragma unroll
    for (int t = 0; t < L; t++) cur[t] = load_next();
    // Step s takes input row y0 - R + s and adds q[R-|d|] to output row y0 - R + s + d for d = dlo..dhi;
    // the accumulator of output row o is slot (o - y0) mod K.  A full step (d = -R..R) completes output
    // row y0 - 2R + s.  The first K-1 steps of a segment only feed rows of the segment (d >= R - s) and
    // the last K-1 only rows above its end, so a segment costs exactly its own additions: segments can
    // be short, which is what lets every SIMD hold its share of waves.
    // The float products of a step are formed one step ahead (a conversion issued right
    // behind the multiply that feeds it would stall the wave for the multiply's latency,
    // and so would an addition right behind its conversion); each step runs conversions of the
    // outer taps -> their additions -> conversions of the inner taps -> next step's products ->
    // additions of the inner taps.
    constexpr int NP = (R + 2) / 2;  // packed multiplies per input
    float_pair pr[NP];
    auto products = [&](float xv, int mhi) __attribute__((always_inline)) {  // taps 0..mhi
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (2 * i <= mhi) {
                const float_pair gp = {g[2 * i], (2 * i + 1 <= R) ? g[2 * i + 1] : 0.f};
                pr[i] = gp * xv;  // one packed multiply, each half rounded to float as xv * g[m] is
            }
        }
    };
    // t = s mod K; mhi / next_mhi = highest tap index this / the next step uses (-1: no next step)
    auto step = [&](int t, int dlo, int dhi, bool completes, int mhi, int next_mhi, bool refill) __attribute__((always_inline)) {
        // in two halves (outer taps, then inner taps), so that only half of the converted products are live
        constexpr int H1 = R / 2;
        double q[R + 1];
#pragma unroll
        for (int m = 0; m <= H1; m++)
            if (m <= mhi) q[m] = (double)((m & 1) ? pr[m >> 1].y : pr[m >> 1].x);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = -R; d <= R; d++)
            if (d >= dlo && d <= dhi && R - (d < 0 ? -d : d) <= H1) acc[(t + d + R + 1) % K] += q[R - (d < 0 ? -d : d)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = H1 + 1; m <= R; m++)
            if (m <= mhi) q[m] = (double)((m & 1) ? pr[m >> 1].y : pr[m >> 1].x);
        __builtin_amdgcn_sched_barrier(0);
        if (next_mhi >= 0) products(cur[(t + 1) % L], next_mhi);
        if (refill) cur[(t + 1) % L] = load_next();  // the row of step s + 1 + L
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = -R; d <= R; d++)
            if (d >= dlo && d <= dhi && R - (d < 0 ? -d : d) > H1) acc[(t + d + R + 1) % K] += q[R - (d < 0 ? -d : d)];
        if (completes) {
            const int slot = (t + 1) % K;  // output row y0 - 2R + s
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)acc[slot]), rout, (int)st_off, 0, 0);
            st_off += (unsigned)row_bytes;
            acc[slot] = 0.0;
        }
        // Pin every accumulator here: left alone the compiler moves additions and
        // conversions across steps and keeps hundreds of converted products alive.
#pragma unroll
        for (int j = 0; j < K; j++) asm volatile("" : "+v"(acc[j]));
        __builtin_amdgcn_sched_barrier(0);
    };
    products(cur[0], 0);
    cur[0] = load_next();  // the row of step L
    // lead-in: steps 0 .. K-2
#pragma unroll
    for (int t = 0; t < K - 1; t++)
        step(t, R - t > -R ? R - t : -R, R, false, t < R ? t : R, t + 1 < R ? t + 1 : R, true);
    // full steps K-1 .. nb*K - 2, a body of K at a time
    for (int b = 1; b < nb; b++) {
#pragma unroll
        for (int u = 0; u < K; u++) step((K - 1 + u) % K, -R, R, true, R, R, true);
    }
    // the last full step, then the K-1 steps that finish the rows still open
    step(K - 1, -R, R, true, R, R, L <= K - 2);
#pragma unroll
    for (int j = 0; j < K - 1; j++)
        step(j, -R, R - 1 - j, true, 2 * R - 1 - j < R ? 2 * R - 1 - j : R,
             j + 1 < K - 1 ? (2 * R - 2 - j < R ? 2 * R - 2 - j : R) : -1, j + 1 + L <= K - 2);
}

// Segment length of the streaming column pass, in whole blocks of K rows.  A SIMD issues for its
// oldest wave first and fills the gaps with the others, so waves on one SIMD finish one after the
// other and the pass takes as long as the fullest SIMD: the segments are cut so that the waves
// come to about one full set of resident waves, rather a little less than a little more.
template <int K>
static int stream_segment(const sift_hip_ctx* ctx, int W, int H) {
    if (ctx->tune.blur_seg > 0) return (ctx->tune.blur_seg + K - 1) / K * K;
    const long cols = (W + 63) / 64;
    const long slots = 1024L * StreamGeom<K>::WAVES;
    auto waves = [&](long m) { return cols * ((H + m * K - 1) / (m * K)); };
    long m = std::max(1L, (long)H * cols / (slots * K));
    if (waves(m) > slots && waves(m + 1) * 100 >= slots * 85) m++;
    return (int)(m * K);
}

template <int K>
static bool launch_col_stream(const sift_hip_ctx* ctx, hipStream_t st, const float* tmp, float* dst, int W, int H,
                              const float* d_taps) {
    int seg = stream_segment<K>(ctx, W, H);
    // the kernel addresses the rows of a segment with 32-bit byte offsets: very wide images get shorter segments
    const long max_rows = (((long)1 << 31) - 1) / ((long)W * 4) - K;
    if (max_rows < K) return false;
    if (seg > max_rows) seg = (int)(max_rows / K * K);
    dim3 grid((W + 63) / 64, (H + seg - 1) / seg);
    if (grid.y > 65535) return false;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_col_stream_kernel<K>), grid, dim3(64), 0, st, tmp, dst, W, H, d_taps, seg);
    return true;
}

// true if a streaming kernel exists for this tap count (the pyramid's own short kernels)
static bool col_stream(const sift_hip_ctx* ctx, hipStream_t st, const float* tmp, float* dst, int W, int H,
                       const float* d_taps, int K) {
    if (ctx->tune.blur_stream >= 0) {
        if (ctx->tune.blur_stream == 0) return false;
    } else if ((size_t)((W + 63) / 64) * ((H + K - 1) / K) < 2048) {
        // even the shortest segments give fewer than two waves per SIMD: the tiled pass spreads better
        // (measured on 1920 x 1080: 39 taps 14 us tiled, 13-18 streaming; 55 taps 17 against 30)
        return false;
    }
    switch (K) {
        case 11: return launch_col_stream<11>(ctx, st, tmp, dst, W, H, d_taps);
        case 15: return launch_col_stream<15>(ctx, st, tmp, dst, W, H, d_taps);
        case 21: return launch_col_stream<21>(ctx, st, tmp, dst, W, H, d_taps);
        case 29: return launch_col_stream<29>(ctx, st, tmp, dst, W, H, d_taps);
        case 39: return launch_col_stream<39>(ctx, st, tmp, dst, W, H, d_taps);
        case 55: return launch_col_stream<55>(ctx, st, tmp, dst, W, H, d_taps);
        default: return false;
    }
}

// ---- both passes in one launch (short kernels on large images) ----------------------
// A 256-thread workgroup produces a 64 x 64 tile: its 64+K-1 input rows and columns go to
// LDS once, the row pass runs over all 64+K-1 rows of the tile into a second LDS array
// (the values the two-pass form would round to float and store in `tmp`), and the column
// pass reads that.  Every output receives the same products in the same order as in the
// two-pass form, so the results are identical; what changes is that the level is read
// and written once instead of twice, at the price of (64+K-1)/64 times the row-pass
// arithmetic.  K is a template parameter: both loops are fully unrolled.
template <typename ACC, int K, int P>
__device__ __forceinline__ void conv_run(const float* __restrict__ src, int stride, const float* __restrict__ taps,
                                         ACC (&acc)[P]) {
    Window<ACC, P> win;
#pragma unroll
    for (int p = 0; p < P; p++) {
        acc[p] = 0;
        win.set(p, src[p * stride]);
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const float g = taps[k];
        win.step(acc, k % P, g);
        if (k + P < K + P - 1) win.set(k % P, src[(k + P) * stride]);
    }
}

// conv_run for float accumulators with the taps in registers, the symmetric half only (g[m] = tap m = tap K-1-m)
template <int K, int P>
__device__ __forceinline__ void conv_run_sym(const float* __restrict__ src, const float (&g)[K / 2 + 1], float (&acc)[P]) {
    Window<float, P> win;
#pragma unroll
    for (int p = 0; p < P; p++) {
        acc[p] = 0;
        win.set(p, src[p]);
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        win.step(acc, k % P, g[k <= K / 2 ? k : K - 1 - k]);
        if (k + P < K + P - 1) win.set(k % P, src[k + P]);
    }
}

template <int K>
struct FusedGeom {
    static constexpr int T = 64;                          // tile edge
    static constexpr int NIN = T + K - 1;                 // input rows and columns of a tile
    static constexpr int PIN = ((NIN + 6) / 8) * 8 + 1;   // LDS pitch, 1 mod 8: 8 rows x 8 eight-column blocks hit 64 banks
    static constexpr size_t lds_bytes = (size_t)NIN * PIN * sizeof(float);
};

template <typename ACC, int K>
__global__ __launch_bounds__(256) void blur_fused_kernel(const float* __restrict__ in, float* __restrict__ out, int W,
                                                         int H, const float* __restrict__ taps) {
    using G = FusedGeom<K>;
    constexpr int R = K >> 1, NIN = G::NIN, PIN = G::PIN;
    extern __shared__ float s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2: remapped so that one
    // XCD owns a contiguous run of tiles, the halo a tile shares with its neighbours is an L2 hit
    // instead of a second fetch over the fabric (bijective for any tile count).
    const int ntile = gridDim.x * gridDim.y;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = ntile >> 3, r8 = ntile & 7, xcd = bid & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int x0 = (lin % (int)gridDim.x) * 64, y0 = (lin / (int)gridDim.x) * 64;
    // Input tile, indices clamped as both passes of the two-pass form clamp them.  The loads go
    // global -> LDS directly (64 consecutive words per wave instruction), so all of a wave's rows
    // are in flight at once and cost no registers.
    {
        const int c0 = min(max(x0 - R + lane, 0), W - 1);
        const int c1 = min(max(x0 - R + 64 + lane, 0), W - 1);
        const bool second = lane + 64 < NIN;
        for (int i = wave; i < NIN; i += 4) {
            const float* row = in + (size_t)min(max(y0 - R + i, 0), H - 1) * W;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + c0),
                                             (__attribute__((address_space(3))) void*)(s + i * PIN), 4, 0, 0);
            if (second)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + c1),
                                                 (__attribute__((address_space(3))) void*)(s + i * PIN + 64), 4, 0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's tile rows have landed
    __syncthreads();
    // Row pass: unit = (tile row, block of 8 output columns).  The eight units of a row sit in one
    // wave, which reads every input of the row before it writes, so the results replace the
    // inputs in place (columns 0..63 of the row).
    for (int u = tid; u < 64 * 8; u += 256) {
        const int row = u >> 3, xb = u & 7;
        ACC acc[8];
        float* d = s + row * PIN + xb * 8;
        conv_run<ACC, K, 8>(d, 1, taps, acc);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int p = 0; p < 8; p++) d[p] = (float)acc[p];
    }
    // the K-1 rows left over would fill only part of a third round of 8-column units:
    // 4-column units spread them over twice as many threads at half the length each
    // (pitch 1 mod 4: 4 rows x 16 four-column blocks hit 64 banks)
    constexpr int PR = ((K - 1) * 8 <= 128 || ((K - 1) * 8 > 256 && (K - 1) * 8 <= 384)) ? 4 : 8;
    for (int u = tid; u < (K - 1) * (64 / PR); u += 256) {
        const int row = 64 + u / (64 / PR), xb = u % (64 / PR);
        ACC acc[PR];
        float* d = s + row * PIN + xb * PR;
        conv_run<ACC, K, PR>(d, 1, taps, acc);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int p = 0; p < PR; p++) d[p] = (float)acc[p];
    }
    __syncthreads();
    // column pass: unit = (column, block of 8 output rows)
    const int x = x0 + lane;
    for (int yb = wave; yb < 8; yb += 4) {
        if (y0 + yb * 8 >= H) break;  // wave-uniform
        ACC acc[8];
        conv_run<ACC, K, 8>(s + (yb * 8) * PIN + lane, PIN, taps, acc);
        if (x < W) {
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int y = y0 + yb * 8 + p;
                if (y < H) out[(size_t)y * W + x] = (float)acc[p];
            }
        }
    }
}

template <typename ACC, int K>
static bool launch_fused(sift_hip_ctx* ctx, hipStream_t st, const float* src, float* dst, int W, int H,
                         const float* d_taps) {
    using G = FusedGeom<K>;
    static signed char state[64] = {0};  // per device: the kernel's LDS request above 64 KB needs an opt-in
    signed char& ok = state[ctx->device & 63];
    if (ok == 0)
        ok = (G::lds_bytes <= (size_t)kMaxColLds ||
              hipFuncSetAttribute(reinterpret_cast<const void*>(&blur_fused_kernel<ACC, K>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes) == hipSuccess) ? 1 : -1;
    if (ok != 1) return false;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_fused_kernel<ACC, K>), dim3((W + 63) / 64, (H + 63) / 64), dim3(256),
                       G::lds_bytes, st, src, dst, W, H, d_taps);
    return true;
}

// whether the one-launch form is the choice for this level
template <typename ACC>
static bool fused_wanted(const sift_hip_ctx* ctx, int W, int H, int K) {
    // exact arithmetic: up to 29 taps (beyond, the extra row-pass work costs more than the second
    // trip through memory saves); float arithmetic: every length that has an instance
    const size_t px = (size_t)W * H;
    // (images of 16 Mpixel and more have the streaming column pass, which moves the break-even down)
    int max_k = sizeof(ACC) == sizeof(double) ? (px >= ((size_t)16 << 20) ? 15 : 29) : 39;
    size_t min_px = (size_t)3 << 20;  // smaller octaves: too few 64 x 64 tiles to fill the chip
    if (ctx->tune.blur_fused >= 0) {  // A/B and test switch: longest kernel, any image size
        max_k = ctx->tune.blur_fused;
        min_px = 0;
    }
    return K <= max_k && px >= min_px && (K == 11 || K == 15 || K == 21 || K == 29 || K == 39);
}

// true if the level was produced by the one-launch form
template <typename ACC>
static bool blur_fused(sift_hip_ctx* ctx, hipStream_t st, const float* src, float* dst, int W, int H,
                       const float* d_taps, int K) {
    switch (K) {
        case 11: return launch_fused<ACC, 11>(ctx, st, src, dst, W, H, d_taps);
        case 15: return launch_fused<ACC, 15>(ctx, st, src, dst, W, H, d_taps);
        case 21: return launch_fused<ACC, 21>(ctx, st, src, dst, W, H, d_taps);
        case 29: return launch_fused<ACC, 29>(ctx, st, src, dst, W, H, d_taps);
        case 39: return launch_fused<ACC, 39>(ctx, st, src, dst, W, H, d_taps);
        default: return false;
    }
}

// ---- both passes in one launch, sweeping down the image (float arithmetic) ------------
// The one-launch form above pays (64+K-1)^2 / 64^2 times the loads and (64+K-1)/64 times the row-pass
// arithmetic of a level, which at 39 taps is 2.5x and 1.6x.  Here a 256-thread workgroup owns a strip of
// TW = 256*NU output columns and a segment of `seg` output rows and walks down it RS = 8 input rows at a
// time: the 8 rows (plus K-1 columns of halo) go global -> LDS directly, two buffers in turn so that the
// next step's rows are in flight while this step computes; the row pass (the same sliding window over 8
// consecutive outputs as everywhere else; a unit = one of the 8 rows x one block of 8 columns, pitch 1 mod 8
// so that the 8 x 8 units of a wave hit 64 banks) writes the 8 row-blurred rows to a third LDS array; for
// the column pass a thread owns NU columns and keeps the last KR >= K row-blurred values of each in
// REGISTERS (a ring whose slot is the row's number mod KR: the loop body covers KR rows, so every slot
// index is a compile-time constant), pushes the 8 new values one by one and, once K rows are in, forms
// an output row from the ring after every push.  Loads: (TW+K-1)/TW x (seg+K-1)/seg of the level; row-pass
// arithmetic: (seg+K-1)/seg; the column pass costs exactly its own multiply-adds and reads nothing but its
// 8 new values per step.  Every output receives the same fused multiply-adds in the same (ascending tap)
// order as in the other float kernels, so the results are identical to theirs.
template <int K, int NU>
struct SweepGeom {
    static constexpr int RS = 8;                           // input rows per step
    static constexpr int KR = (K + RS - 1 + RS - 1) / RS * RS;  // ring length: K values of the oldest pending output + the RS-1 newer rows
    static constexpr int TW = 256 * NU;                    // output columns of a strip
    static constexpr int NINW = TW + K - 1;                // input columns of a strip
    static constexpr int PIN = ((NINW + 6) / 8) * 8 + 1;   // pitch of the input rows
    static constexpr int POUT = TW + 1;                    // pitch of the row-blurred rows
    static constexpr int NCH = (NINW + 63) / 64;           // 64-column chunks of an input row
    static constexpr size_t lds_bytes = (size_t)(2 * RS * PIN + RS * POUT) * sizeof(float);
};

template <int K, int NU>
__global__ __launch_bounds__(256) void blur_sweep_kernel(const float* __restrict__ in, float* __restrict__ out, int W,
                                                         int H, const float* __restrict__ taps, int seg) {
    using G = SweepGeom<K, NU>;
    constexpr int R = K >> 1, RS = G::RS, KR = G::KR, PIN = G::PIN, POUT = G::POUT, NCH = G::NCH, NINW = G::NINW;
    extern __shared__ float s[];
    float* s_rb = s + 2 * RS * PIN;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    TRACE_BEGIN();
    const int x0 = blockIdx.x * G::TW;
    const int y0 = blockIdx.y * seg, y1 = min(y0 + seg, H);
    const int npush = (y1 - y0) + K - 1;  // input rows y0-R .. y1-1+R, clamped to the image as the column pass clamps them
    const int nsteps = (npush + RS - 1) / RS;
    int col[NCH];  // this lane's column in each chunk, clamped as the row pass clamps it
#pragma unroll
    for (int ci = 0; ci < NCH; ci++) col[ci] = min(max(x0 - R + 64 * ci + lane, 0), W - 1);
    auto load_step = [&](int st) __attribute__((always_inline)) {
        float* buf = s + (st & 1) * (RS * PIN);
#pragma unroll
        for (int jj = 0; jj < RS / 4; jj++) {
            const int j = wave + 4 * jj;
            const float* row = in + (size_t)min(max(y0 - R + st * RS + j, 0), H - 1) * W;
#pragma unroll
            for (int ci = 0; ci < NCH; ci++)
                if (64 * ci + lane < NINW)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + col[ci]),
                                                     (__attribute__((address_space(3))) void*)(buf + j * PIN + 64 * ci), 4, 0, 0);
        }
    };
    // the taps are symmetric bit for bit (built from the centre outwards): R+1 scalars serve all K
    float g[R + 1];
#pragma unroll
    for (int m = 0; m <= R; m++) g[m] = taps[m];
    // the ring in both pairings, (2m, 2m+1) and (2m+1, 2m+2), so that the output rows j and j+1, whose tap k reads
    // the consecutive slots i and i+1, share one packed multiply-add whatever the parity of i (cf. Window)
    float_pair re[NU][KR / 2], ro[NU][KR / 2];
#pragma unroll
    for (int n = 0; n < NU; n++)
#pragma unroll
        for (int i = 0; i < KR / 2; i++) re[n][i] = ro[n][i] = float_pair{0.f, 0.f};
    // Output rows wait in registers for one step: a wave must see its loads of the next rows complete (vmcnt(0), the
    // counter does not tell loads from stores) at the top of every step, and stores issued just before that wait
    // would put their whole round trip to memory into it.  Issued right behind the wait, they have a step to finish.
    float_pair pend[NU][RS / 2];
    int pend_st = -1;  // step whose outputs are pending (uniform)
    auto flush = [&]() __attribute__((always_inline)) {
        if (pend_st < 0) return;
#pragma unroll
        for (int n = 0; n < NU; n++) {
            if (x0 + tid + 256 * n < W) {
#pragma unroll
                for (int j = 0; j < RS; j++) {
                    const int sidx = pend_st * RS + j;
                    const int yo = y0 + sidx - (K - 1);
                    if (sidx >= K - 1 && yo < y1)
                        out[(size_t)yo * W + x0 + tid + 256 * n] = (j & 1) ? pend[n][j >> 1].y : pend[n][j >> 1].x;
                }
            }
        }
        pend_st = -1;
    };
    load_step(0);
    for (int st0 = 0; st0 < nsteps; st0 += KR / RS) {
#pragma unroll
        for (int q = 0; q < KR / RS; q++) {
            const int st = st0 + q;
            if (st >= nsteps) break;  // uniform
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's rows of step st have landed
            __syncthreads();                      // ... and everybody's; the column pass of step st-1 is done with s_rb
            flush();
            if (st + 1 < nsteps) load_step(st + 1);
            const float* buf = s + (st & 1) * (RS * PIN);
#pragma unroll
            for (int n = 0; n < NU; n++) {
                const int j = lane & 7, xb = (lane >> 3) + 8 * wave + 32 * n;
                float acc[8];
                conv_run_sym<K, 8>(buf + j * PIN + xb * 8, g, acc);
                float* d = s_rb + j * POUT + xb * 8;
#pragma unroll
                for (int p = 0; p < 8; p++) d[p] = acc[p];
            }
            __syncthreads();
            // the 8 new rows enter the ring (slot = q*RS + j, a compile-time constant after unrolling) ...
#pragma unroll
            for (int j = 0; j < RS; j++)
#pragma unroll
                for (int n = 0; n < NU; n++) {
                    const float v = s_rb[j * POUT + tid + 256 * n];
                    const int slot = q * RS + j;
                    if (slot & 1) {
                        re[n][slot >> 1].y = v;
                        ro[n][slot >> 1].x = v;
                    } else {
                        re[n][slot >> 1].x = v;
                        ro[n][((slot + KR - 2) % KR) >> 1].y = v;
                    }
                }
            // ... and the 8 output rows they complete are formed side by side: independent chains, two rows per
            // packed multiply-add, each adding its products in ascending tap order
            if (st * RS + RS - 1 >= K - 1) {  // uniform: at least the last of them has all its K rows
#pragma unroll
                for (int n = 0; n < NU; n++) {
#pragma unroll
                    for (int j = 0; j < RS / 2; j++) pend[n][j] = float_pair{0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        const float gk = g[k <= R ? k : K - 1 - k];
                        const float_pair gg = {gk, gk};
#pragma unroll
                        for (int j = 0; j < RS / 2; j++) {
                            const int i = (q * RS + 2 * j - (K - 1) + k + 2 * KR) % KR;  // slot of row 2j's tap k; row 2j+1 reads i+1
                            const float_pair v = (i & 1) ? ro[n][i >> 1] : re[n][i >> 1];
                            pend[n][j] = __builtin_elementwise_fma(v, gg, pend[n][j]);
                        }
                    }
                }
                pend_st = st;
            }
        }
    }
    flush();
    TRACE_END(blockIdx.y * gridDim.x + blockIdx.x);
}

// Rows per segment of the sweeping form: about `per_cu` workgroups per compute unit, in whole steps.
template <int K, int NU>
static bool launch_sweep(sift_hip_ctx* ctx, hipStream_t st, const float* src, float* dst, int W, int H,
                         const float* d_taps, int seg_rows) {
    using G = SweepGeom<K, NU>;
    static_assert(G::lds_bytes <= (size_t)kMaxColLds, "strip buffers must fit the default LDS limit");
    const int strips = (W + G::TW - 1) / G::TW;
    int seg = seg_rows;
    if (seg <= 0) {
        const int segs = std::max(1, 256 * 3 / strips);
        seg = (H + segs - 1) / segs;
    }
    seg = std::max(G::RS, (seg + G::RS - 1) / G::RS * G::RS);
    const int nseg = (H + seg - 1) / seg;
    if (nseg > 65535) return false;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_sweep_kernel<K, NU>), dim3(strips, nseg), dim3(256), G::lds_bytes, st, src,
                       dst, W, H, d_taps, seg);
    return true;
}

// Whether the sweeping one-launch form takes this level (float arithmetic only); *seg = rows per segment (0: chosen
// at the launch).
static bool sweep_wanted(const sift_hip_ctx* ctx, int W, int H, int K, int* seg) {
    const int mode = ctx->tune.blur_sweep;  // -1: where it pays, 0: never, n > 0: always, n rows per segment
    if (mode == 0 || !(K == 21 || K == 29 || K == 39)) return false;
    *seg = 0;
    if (mode > 0) {
        *seg = mode;
        return true;
    }
    // About three workgroups per compute unit, but no segment shorter than 64 rows; the segments must be long
    // against the K-1 rows a segment consumes before its first output.  (Measured on 3840 x 2160: 39 taps 34.9 us
    // in segments of 64 rows against 40.0 for the tiled one-launch form, 21 and 29 taps even with it.)
    const int strips = (W + 255) / 256;
    const int segs = std::max(1, 256 * 3 / strips);
    const int rows = std::max(64, (H + segs - 1) / segs);
    *seg = rows;
    if ((size_t)W * H < ((size_t)3 << 20) || H < rows) return false;
    return rows >= 3 * K || K == 39;
}

static bool blur_sweep(sift_hip_ctx* ctx, hipStream_t st, const float* src, float* dst, int W, int H,
                       const float* d_taps, int K, int seg) {
    switch (K) {
        case 21: return launch_sweep<21, 1>(ctx, st, src, dst, W, H, d_taps, seg);
        case 29: return launch_sweep<29, 1>(ctx, st, src, dst, W, H, d_taps, seg);
        case 39: return launch_sweep<39, 1>(ctx, st, src, dst, W, H, d_taps, seg);
        default: return false;
    }
}

template <typename ACC, int P>
static void launch_blur_tp(sift_hip_ctx* ctx, hipStream_t st, const float* src, float* tmp, float* dst, int W, int H,
                           const float* d_taps, int K) {
    if constexpr (sizeof(ACC) == sizeof(float)) {
        int seg = 0;
        if (src != dst && sweep_wanted(ctx, W, H, K, &seg)) {
            ProfScope ps(ctx, SIFT_HIP_K_BLUR_FUSED, st);
            if (blur_sweep(ctx, st, src, dst, W, H, d_taps, K, seg)) return;
        }
    }
    if (src != dst && fused_wanted<ACC>(ctx, W, H, K)) {
        ProfScope ps(ctx, SIFT_HIP_K_BLUR_FUSED, st);
        if (blur_fused<ACC>(ctx, st, src, dst, W, H, d_taps, K)) return;
    }
    {
        ProfScope ps(ctx, (size_t)W * H >= ((size_t)3 << 20) ? SIFT_HIP_K_BLUR_ROW : SIFT_HIP_K_BLUR_ROW_SMALL, st);
        const int n_in = 64 * P + K - 1 + P;
        const size_t lds = (size_t)(n_in + n_in / P + 1) * sizeof(float);
        dim3 grid((W + 64 * P - 1) / (64 * P), H);
        const bool tiny = (size_t)W * H < ((size_t)1 << 16);  // the deepest octaves: one output per thread
        // (LDS: K + 64 taps; the middle run is at most min(K, W + 62) taps long, plus 128 of slack)
        const size_t tiny_lds = (size_t)(K + 64 + std::min(K, W + 62) + 128) * sizeof(float);
        if (tiny && tiny_lds <= (size_t)kMaxColLds)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_row_tiny_kernel<ACC>), dim3((W + 63) / 64, H), dim3(64), tiny_lds, st, src,
                               tmp, W, H, d_taps, K);
        else if (W <= K && (size_t)W * sizeof(float) <= (size_t)kMaxColLds)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_row_whole_kernel<ACC, P>), grid, dim3(64), (size_t)W * sizeof(float), st,
                               src, tmp, W, H, d_taps, K);
        else if (lds <= (size_t)kMaxColLds)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_row_kernel<ACC, P>), grid, dim3(64), lds, st, src, tmp, W, H,
                               d_taps, K);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_row_global_kernel<ACC, P>), grid, dim3(64), 0, st, src, tmp, W, H,
                               d_taps, K);
    }
    {
        ProfScope ps(ctx, (size_t)W * H >= ((size_t)3 << 20) ? SIFT_HIP_K_BLUR_COL : SIFT_HIP_K_BLUR_COL_SMALL, st);
        if constexpr (sizeof(ACC) == sizeof(double)) {
            if (col_stream(ctx, st, tmp, dst, W, H, d_taps, K)) return;
        }
        if ((size_t)W * H < ((size_t)1 << 16) && (size_t)H * 256 + (size_t)(K + 64) * sizeof(float) <= (size_t)kMaxColLds) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_col_tiny_kernel<ACC>), dim3((W + 63) / 64, (H + 3) / 4), dim3(256),
                               (size_t)H * 256 + (size_t)(K + 64) * sizeof(float), st, tmp, dst, W, H, d_taps, K);
            return;
        }
        if (H <= K && (size_t)H * 256 <= (size_t)kMaxColLds) {
            dim3 grid((W + 63) / 64, (H + 4 * P - 1) / (4 * P));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_col_whole_kernel<ACC, P>), grid, dim3(256), (size_t)H * 256, st, tmp,
                               dst, W, H, d_taps, K);
            return;
        }
        // tile height: as tall as keeps the LDS tile modest (occupancy) for this K
        int TH = 16 * P;
        while (TH > 4 * P && (size_t)(TH + K - 1 + P) * 256 > 32 * 1024) TH >>= 1;
        size_t lds = (size_t)(TH + K - 1 + P) * 64 * sizeof(float);
        // Kernels of several hundred taps (deep octaves): a tile of up to 150 KB still beats
        // sliding the window over global memory; taller tiles amortise the K-1 halo rows.
        // (the opt-in is per device: once for each device a context of this process runs on)
        static signed char big_state[64] = {0};  // 0 = not asked yet, 1 = granted, -1 = refused
        signed char& bs = big_state[ctx->device & 63];
        if (bs == 0)
            bs = hipFuncSetAttribute(reinterpret_cast<const void*>(&blur_col_kernel<ACC, P>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kBigColLds) == hipSuccess ? 1 : -1;
        const bool big_ok = bs == 1;
        if (lds > (size_t)kMaxColLds && big_ok) {
            while (TH < 16 * P && (size_t)(2 * TH + K - 1 + P) * 256 <= (size_t)kBigColLds) TH <<= 1;
            lds = (size_t)(TH + K - 1 + P) * 64 * sizeof(float);
        }
        // Small images: a wave alone on its SIMD issues one vector instruction per ~16 cycles, so
        // prefer shorter tiles until there are about four workgroups per compute unit.
        while (TH > 4 * P && (size_t)((W + 63) / 64) * ((H + TH - 1) / TH) < 1024) {
            TH >>= 1;
            lds = (size_t)(TH + K - 1 + P) * 64 * sizeof(float);
        }
        if (lds <= (size_t)(big_ok ? kBigColLds : kMaxColLds)) {
            dim3 grid((W + 63) / 64, (H + TH - 1) / TH);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_col_kernel<ACC, P>), grid, dim3(256), lds, st, tmp, dst, W, H,
                               d_taps, K, TH);
        } else {
            dim3 grid((W + 63) / 64, (H + 4 * P - 1) / (4 * P));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_col_global_kernel<ACC, P>), grid, dim3(256), 0, st, tmp, dst, W, H,
                               d_taps, K);
        }
    }
}

template <typename ACC>
static void launch_blur_t(sift_hip_ctx* ctx, hipStream_t st, const float* src, float* tmp, float* dst, int W, int H,
                          const float* d_taps, int K) {
    // outputs per thread: fewer for small octaves, which otherwise cannot fill 1024 SIMDs
    // (measured on 960 x 540, five levels of 79...309 taps: 4 outputs per thread 191 us against 195 with 2 in
    // exact arithmetic, 127 against 150 in float arithmetic)
    if ((size_t)W * H < (size_t)1 << 18)
        launch_blur_tp<ACC, 2>(ctx, st, src, tmp, dst, W, H, d_taps, K);
    else if ((size_t)W * H < (size_t)3 << 20)
        launch_blur_tp<ACC, 4>(ctx, st, src, tmp, dst, W, H, d_taps, K);
    else
        launch_blur_tp<ACC, 8>(ctx, st, src, tmp, dst, W, H, d_taps, K);
}

void launch_blur(sift_hip_ctx* ctx, hipStream_t st, const float* src, float* tmp, float* dst, int W, int H,
                 const float* d_taps, int K, int blur_mode) {
    if (blur_mode == 1)
        launch_blur_t<float>(ctx, st, src, tmp, dst, W, H, d_taps, K);
    else
        launch_blur_t<double>(ctx, st, src, tmp, dst, W, H, d_taps, K);
}

// DoG images for callers that want them stored (Sift.cpp:161-168): dog[k] = gau[k] - gau[k+1].
__global__ __launch_bounds__(256) void dog_kernel(const float* __restrict__ gau, float* __restrict__ dog, size_t PX,
                                                  int nlev) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= PX) return;
    float a = gau[i];
    for (int k = 0; k + 1 < nlev; k++) {
        const float b = gau[(size_t)(k + 1) * PX + i];
        dog[(size_t)k * PX + i] = a - b;
        a = b;
    }
}

void launch_dog(sift_hip_ctx* ctx, hipStream_t st, const float* gau, float* dog, size_t PX, int nlev) {
    hipLaunchKernelGGL(dog_kernel, dim3((unsigned)((PX + 255) / 256)), dim3(256), 0, st, gau, dog, PX, nlev);
}

}  // namespace sifthip
