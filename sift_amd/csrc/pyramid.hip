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

// ---- separable blur ----------------------------------------------------------------
// P = outputs per thread along the filter direction: 8 for large octaves (one LDS
// read per 8 MACs), 4 and 2 for smaller ones, which otherwise cannot fill 1024 SIMDs.

// The sliding window of a thread: P input values in a ring (slot = input index
// mod P) and one tap step over all of them: acc[p] += win[(p + u) % P] * g.
template <typename ACC, int P>
struct Window {  // float accumulation (blur_mode 1): fused multiply-add
    float v[P];
    __device__ __forceinline__ void set(int slot, float x) { v[slot] = x; }
    __device__ __forceinline__ void step(ACC (&acc)[P], int u, float g) const {
#pragma unroll
        for (int p = 0; p < P; p++) acc[p] = fmaf(v[(p + u) % P], g, acc[p]);
    }
};
// Exact mode: the product is rounded to float first, as `img[..] * gau_mat[k]` is,
// then added in double.  The slots are kept as register pairs so that one packed
// multiply (v_pk_mul_f32, same rounding as two v_mul_f32) serves two outputs;
// conversion and addition stay one instruction per output.
typedef float float_pair __attribute__((ext_vector_type(2)));
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

// LDS index with one pad word per P so that lanes P floats apart hit distinct banks
// (lane stride P+1 words, odd).
template <int P>
__device__ __forceinline__ int padx(int i) { return i + i / P; }

// Row pass: one wave per 512-pixel run of one row.  LDS holds the run plus halo.
template <typename ACC, int P>
__global__ __launch_bounds__(64) void blur_row_kernel(const float* __restrict__ in, float* __restrict__ out, int W,
                                                      int H, const float* __restrict__ taps, int K) {
    extern __shared__ float s[];
    const int lane = threadIdx.x;
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
    const int x = blockIdx.x * 64 + lane;
    const int xc = min(x, W - 1);
    const int y0 = blockIdx.y * TH;
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


template <typename ACC, int P>
static void launch_blur_tp(sift_hip_ctx* ctx, hipStream_t st, const float* src, float* tmp, float* dst, int W, int H,
                           const float* d_taps, int K) {
    {
        ProfScope ps(ctx, (size_t)W * H >= ((size_t)3 << 20) ? SIFT_HIP_K_BLUR_ROW : SIFT_HIP_K_BLUR_ROW_SMALL, st);
        const int n_in = 64 * P + K - 1 + P;
        const size_t lds = (size_t)(n_in + n_in / P + 1) * sizeof(float);
        dim3 grid((W + 64 * P - 1) / (64 * P), H);
        if (lds <= (size_t)kMaxColLds)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_row_kernel<ACC, P>), grid, dim3(64), lds, st, src, tmp, W, H,
                               d_taps, K);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(blur_row_global_kernel<ACC, P>), grid, dim3(64), 0, st, src, tmp, W, H,
                               d_taps, K);
    }
    {
        ProfScope ps(ctx, (size_t)W * H >= ((size_t)3 << 20) ? SIFT_HIP_K_BLUR_COL : SIFT_HIP_K_BLUR_COL_SMALL, st);
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
    if ((size_t)W * H < (size_t)1 << 20)
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
