// multiBandBlend of the stitcher (Sift/stitch/Blend.cpp:638-740) on gfx950: the overlap
// of two 8-bit RGB images is cut out, its black border filled, turned into two
// five-level Laplacian pyramids (5-tap Gaussian + the reference's bicubic resampler),
// blended level by level, collapsed and written back into both images.
//
// Every kernel is elementwise or a short per-line scan over float RGB images of the
// overlap's size: byte and float work bound by memory, not arithmetic.  The
// arithmetic follows the reference expression by expression (float products, double
// sums where its expressions promote), including what it computes by accident: its
// column blur takes G from channel 2 and B from channel 3 (the next pixel's R) of the
// row-blurred image, and its resampler derives the output height from the input WIDTH.
// Memory the reference reads without having written it is taken as 0.0f.
#include <algorithm>
#include <cmath>
#include <vector>

#include "internal.h"

namespace sifthip {

constexpr int kPyrLevels = 5;  // PYR_OCTAVE, Blend.cpp:30
constexpr int kPyrRadius = 2;  // PYR_R, Blend.cpp:33

struct FImg {  // float RGB image in device memory
    float* p = nullptr;
    int w = 0, h = 0;
    size_t floats() const { return (size_t)std::max(w, 0) * (size_t)std::max(h, 0) * 3; }
};

// ---- overlap copy (buildLaplacianMap, Blend.cpp:519-536) ------------------------------
__global__ __launch_bounds__(256) void overlap_copy_kernel(const uint8_t* __restrict__ rgb, int imgW, int imgH, int disx,
                                                           int disy, float* __restrict__ tmp, uint8_t* __restrict__ flag,
                                                           int W, int H) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= W * H) return;
    const int j = e / W, i = e - j * W;
    float v[3] = {0.f, 0.f, 0.f};
    bool f = false;
    if (j + disy < imgH && j + disy >= 0 && i + disx < imgW) {
        const uint8_t* s = rgb + ((size_t)(j + disy) * imgW + (i + disx)) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            v[c] = (float)s[c];
            f = f || s[c] != 0;
        }
    }
    tmp[(size_t)e * 3 + 0] = v[0];
    tmp[(size_t)e * 3 + 1] = v[1];
    tmp[(size_t)e * 3 + 2] = v[2];
    flag[e] = f ? 1 : 0;
}

// ---- black-border extension (Blend.cpp:542-597) ---------------------------------------
// The reference walks the overlap column by column, then row by row.  In each line it
// copies the sample next to the first non-black one over everything before it, and the
// sample before the last non-black one over everything after it.  Its two indices (ur,
// dr) are ordinary locals that keep their value from line to line -- and from the
// column sweep to the row sweep -- when a line holds nothing but black.  Lines are
// independent apart from that carry, so each sweep runs as: per-line search (parallel),
// carry along the lines (a scan), per-line fill + second search (parallel), carry,
// per-line fill (parallel).  A line is `len` samples `si` floats apart; lines are `so`
// floats apart.
__device__ __forceinline__ bool lit(const float* p) { return p[0] != 0.f || p[1] != 0.f || p[2] != 0.f; }

__global__ __launch_bounds__(64) void line_first_kernel(const float* __restrict__ img, int nlines, int len, size_t so,
                                                        size_t si, int* __restrict__ first) {
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= nlines) return;
    const float* p = img + (size_t)l * so;
    int f = -1;
    for (int k = 0; k < len; k++)
        if (lit(p + (size_t)k * si)) {
            f = k;
            break;
        }
    first[l] = f;
}

// found[l] >= 0 replaces the carried value, -1 keeps it; at_least: lower bound applied to the
// carried value after every line (the reference's `dr = dr < 1 ? 1 : dr`), INT_MIN for none.
// One 256-thread workgroup: line l takes the value of the last line <= l that found something
// (a max-scan over "position of the last hit"), or the value carried in, 256 lines per step.
__global__ __launch_bounds__(256) void carry_kernel(int* __restrict__ found, int nlines, int* __restrict__ carried,
                                                    int at_least) {
    __shared__ int s_f[256];
    __shared__ int s_wave[4];
    __shared__ int s_prev;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_prev = *carried;
    __syncthreads();
    for (int base = 0; base < nlines; base += 256) {
        const int l = base + tid;
        const int f = l < nlines ? found[l] : -1;
        s_f[tid] = f;
        int pos = f >= 0 ? tid : -1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(pos, o);
            if (lane >= o) pos = max(pos, t);
        }
        if (lane == 63) s_wave[wave] = pos;
        __syncthreads();
        for (int w = 0; w < wave; w++) pos = max(pos, s_wave[w]);
        int v = pos >= 0 ? s_f[pos] : s_prev;
        if (v < at_least) v = at_least;
        if (l < nlines) found[l] = v;
        __syncthreads();
        if (l == min(base + 255, nlines - 1)) s_prev = v;
        __syncthreads();
    }
    if (tid == 0) *carried = s_prev;
}

__global__ __launch_bounds__(64) void line_upfill_last_kernel(float* __restrict__ img, int nlines, int len, size_t so,
                                                              size_t si, const int* __restrict__ ur,
                                                              int* __restrict__ last) {
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= nlines) return;
    float* p = img + (size_t)l * so;
    const int u = ur[l];
    const float* s = p + (size_t)min(u + 1, len - 1) * si;
    const float v0 = s[0], v1 = s[1], v2 = s[2];
    for (int k = min(u, len - 1); k >= 0; k--) {
        float* d = p + (size_t)k * si;
        d[0] = v0;
        d[1] = v1;
        d[2] = v2;
    }
    int f = -1;
    for (int k = len - 1; k >= 0; k--)
        if (lit(p + (size_t)k * si)) {
            f = k;
            break;
        }
    last[l] = f;
}

__global__ __launch_bounds__(64) void line_downfill_kernel(float* __restrict__ img, int nlines, int len, size_t so,
                                                           size_t si, const int* __restrict__ dr) {
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= nlines) return;
    float* p = img + (size_t)l * so;
    const int d0 = dr[l];  // >= 1
    if (d0 >= len) return;
    const float* s = p + (size_t)(d0 - 1) * si;
    const float v0 = s[0], v1 = s[1], v2 = s[2];
    for (int k = d0; k < len; k++) {
        float* d = p + (size_t)k * si;
        d[0] = v0;
        d[1] = v1;
        d[2] = v2;
    }
}

// ---- GauBlur3d (Blend.cpp:370-429) ------------------------------------------------------
// taps: gau_matrix(sigma, 5); float product, double sum in tap order, clamp to edge.
__global__ __launch_bounds__(256) void blur3_row_kernel(const float* __restrict__ in, float* __restrict__ out, int W,
                                                        int H, const float* __restrict__ g, int K) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;  // one channel of one pixel
    if (e >= (size_t)W * H * 3) return;
    const int c = (int)(e % 3);
    const size_t px = e / 3;
    const int i = (int)(px % W);
    const size_t j = px / W;
    const int r = K / 2;
    double s = 0;
    for (int k = 0; k < K; k++) {
        const int idx = min(max(i - r + k, 0), W - 1);
        const float prod = in[(j * W + idx) * 3 + c] * g[k];
        s += (double)prod;
    }
    out[e] = (float)s;
}

// Column pass as written in the reference: R from channel 0, G from channel 2, B from channel 3
// (the R of the next pixel; one float past the end of the image counts as 0).
__global__ __launch_bounds__(256) void blur3_col_kernel(const float* __restrict__ gx, float* __restrict__ out, int W,
                                                        int H, const float* __restrict__ g, int K) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)W * H * 3;
    if (e >= total) return;
    const int c = (int)(e % 3);
    const size_t px = e / 3;
    const int i = (int)(px % W);
    const int j = (int)(px / W);
    const int r = K / 2;
    const int off = c == 0 ? 0 : c + 1;
    double s = 0;
    for (int k = 0; k < K; k++) {
        const int idx = min(max(j - r + k, 0), H - 1);
        const size_t a = ((size_t)idx * W + i) * 3 + off;
        const float v = a < total ? gx[a] : 0.f;
        const float prod = v * g[k];
        s += (double)prod;
    }
    out[e] = (float)s;
}

// ---- Bicubic (Blend.cpp:188-367) for a diagonal scale s ----------------------------------
__device__ __forceinline__ float bicubic_poly(float x) {
    const float ax = fabsf(x);
    const float a = -0.5f;
    const double d = (double)ax;
    // pow(float, int) is evaluated in double by the reference; x*x is exact there and (x*x)*x rounds once
    if (ax <= 1.f) return (float)((double)(a + 2.f) * ((d * d) * d) - (double)(a + 3.f) * (d * d) + (double)1.f);
    if (ax < 2.0f)
        return (float)((double)a * ((d * d) * d) - (double)(5.f * a) * (d * d) + (double)(8.f * a * ax) - (double)(4.f * a));
    return 0.f;
}

__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ src, int sw, int sh, float s, float left,
                                                      float top, float* __restrict__ dst, int dw, int dh) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= dw * dh) return;
    const int i = e / dw, j = e - i * dw;
    const float x = ((float)j + left) / s;
    const float y = ((float)i + top) / s;
    const int x1 = (int)x, y1 = (int)y;
    const int x0 = x1 - 1, y0 = y1 - 1, x3 = x1 + 2, y3 = y1 + 2;
    float o[3] = {0.f, 0.f, 0.f};
    if (x0 >= 0 && x3 < sw && y0 >= 0 && y3 < sh) {
        float wx[4], wy[4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            wx[a] = bicubic_poly(x - (float)(x0 + a));
            wy[a] = bicubic_poly(y - (float)(y0 + a));
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float acc = 0.f;
            bool started = false;
            // the reference's sum: x-major, rows inside, left to right
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const float term = src[((size_t)(y0 + b) * sw + (x0 + a)) * 3 + c] * (wx[a] * wy[b]);
                    acc = started ? acc + term : term;
                    started = true;
                }
            o[c] = acc;
        }
        const float* q = src + ((size_t)y1 * sw + x1) * 3;
        if (fabsf(o[0] - q[0]) > 198.f || fabsf(o[1] - q[1]) > 198.f || fabsf(o[2] - q[2]) > 198.f) {
            o[0] = q[0];
            o[1] = q[1];
            o[2] = q[2];
        }
    }
    dst[(size_t)e * 3 + 0] = o[0];
    dst[(size_t)e * 3 + 1] = o[1];
    dst[(size_t)e * 3 + 2] = o[2];
}

// m -= n inside n's extent, 0 outside it (sub(), Blend.cpp:70-90: the reference leaves the rest of
// its new buffer unwritten)
__global__ __launch_bounds__(256) void sub_kernel(float* __restrict__ m, int mw, int mh, const float* __restrict__ n, int nw,
                                                  int nh) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= mw * mh) return;
    const int i = e / mw, j = e - i * mw;
    float* d = m + (size_t)e * 3;
    if (i < nh && j < nw) {
        const float* s = n + ((size_t)i * nw + j) * 3;
        d[0] = d[0] - s[0];
        d[1] = d[1] - s[1];
        d[2] = d[2] - s[2];
    } else {
        d[0] = d[1] = d[2] = 0.f;
    }
}

// Blend of one pyramid level (Blend.cpp:657-683): the coarsest level mixes with the column
// weights k, the others switch from left to right at the centre column.
__global__ __launch_bounds__(256) void level_blend_kernel(const float* __restrict__ l, const float* __restrict__ r,
                                                          float* __restrict__ ls, int w, int h, int coarsest,
                                                          const float* __restrict__ kcol) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= w * h) return;
    const int i = e % w;
    const int center = w / 2;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const size_t a = (size_t)e * 3 + c;
        float v;
        if (coarsest) {
            const float k = kcol[i];
            v = l[a] * k;
            v += r[a] * (1.f - k);
        } else if (i == center) {
            v = (l[a] + r[a]) / 2.f;
        } else if (i > center) {
            v = r[a];
        } else {
            v = l[a];
        }
        ls[a] = v;
    }
}

// lo += up inside up's extent, clamped to [0, 255] there (Blend.cpp:690-703)
__global__ __launch_bounds__(256) void collapse_kernel(float* __restrict__ lo, int lw, int lh, const float* __restrict__ up,
                                                       int uw, int uh) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= lw * lh) return;
    const int j = e / lw, i = e - j * lw;
    if (j >= uh || i >= uw) return;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v = lo[(size_t)e * 3 + c] + up[((size_t)j * uw + i) * 3 + c];
        v = v < 0.f ? 0.f : v > 255.f ? 255.f : v;
        lo[(size_t)e * 3 + c] = v;
    }
}

// blendImg (Blend.cpp:454-490)
__global__ __launch_bounds__(256) void write_back_kernel(uint8_t* __restrict__ rgb, int imgW, int imgH, int disx, int disy,
                                                         const float* __restrict__ ov, int W, int H,
                                                         const uint8_t* __restrict__ bol, const uint8_t* __restrict__ bor) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= W * H) return;
    const int j = e / W, i = e - j * W;
    if (i >= W - 1) return;
    if (!(bol[e] && bor[e] && bol[e + 1] && bor[e + 1])) return;
    if (!(j + disy < imgH && i + disx < imgW)) return;
    uint8_t* d = rgb + ((size_t)(j + disy) * imgW + (i + disx)) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float v = ov[(size_t)e * 3 + c];
        d[c] = (unsigned char)((v < 0.f) ? 0.f : (v > 255.f) ? 255.f : v);
    }
}

// ---- host side ----------------------------------------------------------------------------
static unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

// Output size and origin of Bicubic for the scale matrix diag(s, s) (Blend.cpp:204-231)
static void bicubic_geometry(int sw, int sh, float s, int* dw, int* dh, float* left, float* top) {
    float x = sw * 1.0f, y = 0.0f;
    const float u1 = x * s + y * 0.f, v1 = x * 0.f + y * s;
    x = sw * 1.0f;
    y = sh * 1.0f;
    const float u2 = x * s + y * 0.f, v2 = x * 0.f + y * s;
    x = 0.0f;
    y = sw * 1.0f;  // the reference takes the width here too (Blend.cpp:219)
    const float u3 = x * s + y * 0.f, v3 = x * 0.f + y * s;
    *left = std::min(std::min(std::min(0.0f, u1), u2), u3);
    const float right = std::max(std::max(std::max(0.0f, u1), u2), u3);
    *top = std::min(std::min(std::min(0.0f, v1), v2), v3);
    const float down = std::max(std::max(std::max(0.0f, v1), v2), v3);
    *dw = int(std::abs(right - *left));
    *dh = int(std::abs(down - *top));
}

static void launch_bicubic(hipStream_t st, const FImg& src, float s, const FImg& dst, float left, float top) {
    if (dst.w <= 0 || dst.h <= 0) return;
    hipLaunchKernelGGL(bicubic_kernel, dim3(blocks_for((size_t)dst.w * dst.h)), dim3(256), 0, st, src.p, src.w, src.h, s, left,
                       top, dst.p, dst.w, dst.h);
}

// Column weights of the coarsest level: getGaussianKernel_rr(x, y, x) (Blend.cpp:166-185), which
// depends on the column only.
static std::vector<float> coarsest_weights(int x) {
    std::vector<float> k((size_t)std::max(x, 0), 0.f);
    const int dx = x;
    const float half = (dx - 1) / 2.f;
    const float sigma = std::sqrt((-1.f) * std::pow((float)x - 1.f - half, 2.f) / (2.f * std::log(0.5f)));
    for (int i = (x - dx); i < x; i++) {
        float g;
        if (i <= (x - half)) g = std::exp((-1) * i * i / (2 * sigma * sigma));
        else g = 1.f - std::exp((-1.f) * std::pow(x - i - 1.f, 2.f) / (2.f * sigma * sigma));
        k[(size_t)i] = g;
    }
    return k;
}

// taps of the pyramid's blur: Gaus::gau_matrix(sigma, 2 * PYR_R + 1) through the library's own entry point
static int pyramid_taps(float* out) {
    const float sigma = std::sqrt((-1) * std::pow(2.f, 2.f) / (2.f * std::log(0.5f)));
    return sift_hip_gau_matrix(sigma, kPyrRadius * 2 + 1, out, 16);
}

struct Side {
    FImg lvl[kPyrLevels];
    uint8_t* flag = nullptr;
};

int run_multiband_blend(sift_hip_ctx* ctx, uint8_t* d_l, int lw, int lh, uint8_t* d_r, int rw, int rh, int dx, int dy) {
    hipStream_t st = ctx->stream;
    const int W = dx, H = std::abs(dy);
    // level sizes (the same for both sides) and the resampling geometry between them
    int w[kPyrLevels], h[kPyrLevels];
    float dl[kPyrLevels], dt[kPyrLevels];  // origin of the down-sampling that produces level i
    w[0] = W;
    h[0] = H;
    for (int i = 1; i < kPyrLevels; i++) bicubic_geometry(w[i - 1], h[i - 1], 0.5f, &w[i], &h[i], &dl[i], &dt[i]);
    int uw[kPyrLevels], uh[kPyrLevels];  // size of level i+1 scaled up by 2
    float ul[kPyrLevels], ut[kPyrLevels];
    for (int a = 0; a + 1 < kPyrLevels; a++) bicubic_geometry(w[a + 1], h[a + 1], 2.0f, &uw[a], &uh[a], &ul[a], &ut[a]);
    // one workspace: two sides x five levels, the blended levels, two scratch images, flags, line indices, taps
    size_t fl = 0;
    for (int i = 0; i < kPyrLevels; i++) fl += (size_t)std::max(w[i], 0) * std::max(h[i], 0) * 3;
    size_t scratch = (size_t)W * H * 3;
    for (int a = 0; a + 1 < kPyrLevels; a++) scratch = std::max(scratch, (size_t)std::max(uw[a], 0) * std::max(uh[a], 0) * 3);
    const int nline = std::max(W, H);
    const size_t bytes = (3 * fl + 2 * scratch) * sizeof(float) + 2 * (size_t)W * H + (size_t)(2 * nline + 8) * sizeof(int) +
                         (size_t)(std::max(w[kPyrLevels - 1], 0) + 32) * sizeof(float) + 1024;
    int rc = ensure(ctx, ctx->blend_ws, bytes);
    if (rc) return rc;
    float* fp = (float*)ctx->blend_ws.p;
    Side side[2];
    FImg ls[kPyrLevels];
    for (int s = 0; s < 3; s++)
        for (int i = 0; i < kPyrLevels; i++) {
            FImg& im = s < 2 ? side[s].lvl[i] : ls[i];
            im.p = fp;
            im.w = w[i];
            im.h = h[i];
            fp += im.floats();
        }
    float* scr_a = fp;
    fp += scratch;
    float* scr_b = fp;
    fp += scratch;
    float* d_taps = fp;
    fp += 16;
    float* d_kcol = fp;
    fp += std::max(w[kPyrLevels - 1], 0) + 16;
    int* d_idx = (int*)fp;         // [nline] per-line index
    int* d_carry = d_idx + nline;  // [2]: ur, dr carried from line to line and from columns to rows
    uint8_t* d_flags = (uint8_t*)(d_carry + 8 + nline);
    side[0].flag = d_flags;
    side[1].flag = d_flags + (size_t)W * H;

    float taps[16];
    const int K = pyramid_taps(taps);
    if (K <= 0) return SIFT_HIP_ERR_ARG;
    const std::vector<float> kcol = coarsest_weights(w[kPyrLevels - 1]);
    // (pageable sources: the copies below complete before the call returns to code that could free them)
    if (!hip_ok(ctx, hipMemcpyAsync(d_taps, taps, (size_t)K * 4, hipMemcpyHostToDevice, st), "upload taps")) return SIFT_HIP_ERR_HIP;
    if (!kcol.empty() &&
        !hip_ok(ctx, hipMemcpyAsync(d_kcol, kcol.data(), kcol.size() * 4, hipMemcpyHostToDevice, st), "upload weights"))
        return SIFT_HIP_ERR_HIP;
    if (!hip_ok(ctx, hipStreamSynchronize(st), "upload")) return SIFT_HIP_ERR_HIP;

    const size_t npx = (size_t)W * H;
    for (int s = 0; s < 2; s++) {
        const bool right = s == 1;
        const uint8_t* rgb = right ? d_r : d_l;
        const int imgW = right ? rw : lw, imgH = right ? rh : lh;
        int disx = right ? 0 : (imgW - dx);
        const int disy = (dy >= 0) ? (right ? 0 : (imgH - dy)) : (right ? (imgH + dy) : 0);
        if (disx < 0) disx = 0;
        FImg& base = side[s].lvl[0];
        hipLaunchKernelGGL(overlap_copy_kernel, dim3(blocks_for(npx)), dim3(256), 0, st, rgb, imgW, imgH, disx, disy, base.p,
                           side[s].flag, W, H);
        // border extension: columns (lines W apart... a column is a line of H samples 3*W floats apart), then rows
        hipMemsetAsync(d_carry, 0, 2 * sizeof(int), st);
        for (int sweep = 0; sweep < 2; sweep++) {
            const int nl = sweep == 0 ? W : H, len = sweep == 0 ? H : W;
            const size_t so = sweep == 0 ? 3 : (size_t)3 * W, si = sweep == 0 ? (size_t)3 * W : 3;
            const dim3 g((nl + 63) / 64);
            hipLaunchKernelGGL(line_first_kernel, g, dim3(64), 0, st, (const float*)base.p, nl, len, so, si, d_idx);
            hipLaunchKernelGGL(carry_kernel, dim3(1), dim3(256), 0, st, d_idx, nl, d_carry + 0, INT32_MIN);
            hipLaunchKernelGGL(line_upfill_last_kernel, g, dim3(64), 0, st, base.p, nl, len, so, si, (const int*)d_idx,
                               d_idx + nline + 8);
            hipLaunchKernelGGL(carry_kernel, dim3(1), dim3(256), 0, st, d_idx + nline + 8, nl, d_carry + 1, 1);
            hipLaunchKernelGGL(line_downfill_kernel, g, dim3(64), 0, st, base.p, nl, len, so, si,
                               (const int*)(d_idx + nline + 8));
        }
        // Gaussian pyramid: blur, then every second sample (Bicubic at scale 0.5)
        for (int i = 1; i < kPyrLevels; i++) {
            const FImg& src = side[s].lvl[i - 1];
            if (src.floats() == 0) break;
            hipLaunchKernelGGL(blur3_row_kernel, dim3(blocks_for(src.floats())), dim3(256), 0, st, (const float*)src.p, scr_a,
                               src.w, src.h, (const float*)d_taps, K);
            hipLaunchKernelGGL(blur3_col_kernel, dim3(blocks_for(src.floats())), dim3(256), 0, st, (const float*)scr_a, scr_b,
                               src.w, src.h, (const float*)d_taps, K);
            FImg blurred = src;
            blurred.p = scr_b;
            launch_bicubic(st, blurred, 0.5f, side[s].lvl[i], dl[i], dt[i]);
        }
        // Laplacian levels: level a minus level a+1 scaled up
        for (int a = 0; a + 1 < kPyrLevels; a++) {
            const FImg& m = side[s].lvl[a];
            if (m.floats() == 0) continue;
            FImg up;
            up.p = scr_a;
            up.w = uw[a];
            up.h = uh[a];
            launch_bicubic(st, side[s].lvl[a + 1], 2.0f, up, ul[a], ut[a]);
            hipLaunchKernelGGL(sub_kernel, dim3(blocks_for((size_t)m.w * m.h)), dim3(256), 0, st, m.p, m.w, m.h,
                               (const float*)up.p, std::max(up.w, 0), std::max(up.h, 0));
        }
    }
    for (int a = 0; a < kPyrLevels; a++) {
        if (ls[a].floats() == 0) continue;
        hipLaunchKernelGGL(level_blend_kernel, dim3(blocks_for((size_t)ls[a].w * ls[a].h)), dim3(256), 0, st,
                           (const float*)side[0].lvl[a].p, (const float*)side[1].lvl[a].p, ls[a].p, ls[a].w, ls[a].h,
                           a == kPyrLevels - 1 ? 1 : 0, (const float*)d_kcol);
    }
    for (int a = kPyrLevels - 1; a > 0; a--) {
        if (ls[a - 1].floats() == 0) continue;
        FImg up;
        up.p = scr_a;
        up.w = uw[a - 1];
        up.h = uh[a - 1];
        launch_bicubic(st, ls[a], 2.0f, up, ul[a - 1], ut[a - 1]);
        hipLaunchKernelGGL(collapse_kernel, dim3(blocks_for((size_t)ls[a - 1].w * ls[a - 1].h)), dim3(256), 0, st, ls[a - 1].p,
                           ls[a - 1].w, ls[a - 1].h, (const float*)up.p, std::max(up.w, 0), std::max(up.h, 0));
    }
    for (int s = 0; s < 2; s++) {
        const bool right = s == 1;
        uint8_t* rgb = right ? d_r : d_l;
        const int imgW = right ? rw : lw, imgH = right ? rh : lh;
        int disx = right ? 0 : (imgW - dx);
        int disy = dy >= 0 ? (right ? 0 : (imgH - dy)) : (right ? (imgH + dy) : 0);
        if (disy < 0) disy = 0;
        if (disx < 0) disx = 0;
        hipLaunchKernelGGL(write_back_kernel, dim3(blocks_for(npx)), dim3(256), 0, st, rgb, imgW, imgH, disx, disy,
                           (const float*)ls[0].p, W, H, (const uint8_t*)side[0].flag, (const uint8_t*)side[1].flag);
    }
    return SIFT_HIP_OK;
}

}  // namespace sifthip
