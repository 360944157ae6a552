// The two geometric warps between RANSAC and the blender (reference:
// Sift/stitch/stitch.cpp): inverse perspective mapping of image 2 into image
// 1's frame and the cylindrical projection of both images.  Per-pixel gathers
// over interleaved 8-bit RGB, bound by memory; every operation is a float
// add/mul/div/sqrt in the reference's order (the library is built with
// contraction off), so the bytes equal a CPU evaluation of the same expressions.
#include "internal.h"

namespace sifthip {
namespace {

constexpr int PXT = 4;  // consecutive pixels of a row per thread: 12 bytes = three aligned words

// stitch.cpp:19-63.  A sample whose 2x2 neighbourhood leaves the image is black.
__device__ __forceinline__ void bilinear(const uint8_t* __restrict__ rgb, int cols, int rows, float fx, float fy,
                                         uint8_t out[3]) {
    const int x = (int)fx, y = (int)fy;
    const float l_x = fx - (float)x, r_x = 1.f - l_x;
    const float t_y = fy - (float)y, b_y = 1.f - t_y;
    float c[3] = {0.f, 0.f, 0.f};
    if (x >= 0 && x < cols - 1 && y >= 0 && y < rows - 1) {
        // the six bytes of pixels (x, y), (x+1, y) and of the row below, as one word and one
        // half-word each (global memory takes unaligned accesses)
        const uint8_t* p0 = rgb + ((size_t)y * cols + x) * 3;
        const uint8_t* p1 = p0 + (size_t)cols * 3;
        uint32_t a0, a1;
        uint16_t b0, b1;
        __builtin_memcpy(&a0, p0, 4);
        __builtin_memcpy(&b0, p0 + 4, 2);
        __builtin_memcpy(&a1, p1, 4);
        __builtin_memcpy(&b1, p1 + 4, 2);
        const uint64_t top = (uint64_t)a0 | ((uint64_t)b0 << 32), bot = (uint64_t)a1 | ((uint64_t)b1 << 32);
        const float w00 = r_x * b_y, w01 = l_x * b_y, w10 = r_x * t_y, w11 = l_x * t_y;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float v = (float)(uint32_t)(top >> (8 * k) & 0xff) * w00;
            v += (float)(uint32_t)(top >> (8 * (k + 3)) & 0xff) * w01;
            v += (float)(uint32_t)(bot >> (8 * k) & 0xff) * w10;
            v += (float)(uint32_t)(bot >> (8 * (k + 3)) & 0xff) * w11;
            c[k] = v;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = (uint8_t)(c[k] > 255.f ? 255.f : c[k] < 0.f ? 0.f : c[k]);
}

__device__ __forceinline__ void store_run(uint8_t* __restrict__ row, int j0, int ncols, const uint8_t (&px)[PXT][3],
                                          unsigned valid) {
    uint8_t* o = row + (size_t)j0 * 3;
    if (valid == (1u << PXT) - 1u && j0 + PXT <= ncols && (reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
        uint32_t w[3];
        const uint8_t* b = &px[0][0];
#pragma unroll
        for (int t = 0; t < 3; t++)
            w[t] = (uint32_t)b[4 * t] | ((uint32_t)b[4 * t + 1] << 8) | ((uint32_t)b[4 * t + 2] << 16) |
                   ((uint32_t)b[4 * t + 3] << 24);
        uint32_t* ow = reinterpret_cast<uint32_t*>(o);
        ow[0] = w[0];
        ow[1] = w[1];
        ow[2] = w[2];
        return;
    }
#pragma unroll
    for (int p = 0; p < PXT; p++)
        if ((valid >> p & 1u) && j0 + p < ncols) {
            o[3 * p] = px[p][0];
            o[3 * p + 1] = px[p][1];
            o[3 * p + 2] = px[p][2];
        }
}

struct Homog {
    float h[9];
};

// _WarpPerspective, stitch.cpp:106-141: output pixel (j, i) takes the source
// position given by the closed-form inverse of H; pixels mapping outside the
// source keep their value.
__global__ __launch_bounds__(256) void warp_perspective_kernel(const uint8_t* __restrict__ src, int s_col, int s_row,
                                                               Homog T, uint8_t* __restrict__ dst, int d_col,
                                                               int d_row) {
    const int j0 = (blockIdx.x * 64 + threadIdx.x) * PXT;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (i >= d_row || j0 >= d_col) return;
    const float* H = T.h;
    const float fi = (float)i;
    const float b3 = H[3] - H[6] * fi, b4 = H[4] - H[7] * fi, b5 = H[5] - H[8] * fi;
    uint8_t px[PXT][3];
    unsigned valid = 0;
#pragma unroll
    for (int p = 0; p < PXT; p++) {
        const float fj = (float)(j0 + p);
        const float a0 = H[0] - H[6] * fj, a1 = H[1] - H[7] * fj, a2 = H[2] - H[8] * fj;
        const float den = a1 * b3 - a0 * b4;
        const float x = (a2 * b4 - a1 * b5) / den;
        const float y = (a0 * b5 - a2 * b3) / den;
        px[p][0] = px[p][1] = px[p][2] = 0;
        if (j0 + p < d_col && x < (float)s_col && x >= 0.f && y < (float)s_row && y >= 0.f) {
            bilinear(src, s_col, s_row, x, y, px[p]);
            valid |= 1u << p;
        }
    }
    if (valid) store_run(dst + (size_t)i * d_col * 3, j0, d_col, px, valid);
}

// warping, stitch.cpp:178-221, one image: cylindrical projection with focal
// length FL; every output pixel is written (black where the source position
// leaves the image); the source positions of the first and last rows go to
// upedge / downedge.
__global__ __launch_bounds__(256) void warp_cylinder_kernel(const uint8_t* __restrict__ src, int w, int h, float FL,
                                                            uint8_t* __restrict__ out, float2* __restrict__ upedge,
                                                            float2* __restrict__ downedge) {
    const int i0 = (blockIdx.x * 64 + threadIdx.x) * PXT;
    const int j = blockIdx.y * 4 + threadIdx.y;
    if (j >= h || i0 >= w) return;
    const float mid_x = (float)w / 2.f, mid_y = (float)h / 2.f;
    uint8_t px[PXT][3];
#pragma unroll
    for (int p = 0; p < PXT; p++) {
        const int i = i0 + p;
        const float dx = i - mid_x;
        const float k = sqrtf(FL * FL + dx * dx) / FL;
        const float x = dx * k + mid_x;
        const float y = (j - mid_y) * k + mid_y;
        px[p][0] = px[p][1] = px[p][2] = 0;
        if (i < w) {
            if (x >= 0.f && y >= 0.f && x < (float)w && y < (float)h) bilinear(src, w, h, x, y, px[p]);
            if (j == 0) {
                if (upedge) upedge[i] = make_float2(x, y);
            } else if (j == h - 1) {
                if (downedge) downedge[i] = make_float2(x, y);
            }
        }
    }
    store_run(out + (size_t)j * w * 3, i0, w, px, (1u << PXT) - 1u);
}

}  // namespace

void launch_warp_perspective(sift_hip_ctx* ctx, hipStream_t st, const uint8_t* d_src, int s_col, int s_row,
                             const float* H9, uint8_t* d_dst, int d_col, int d_row) {
    ProfScope ps(ctx, SIFT_HIP_K_WARP, st);
    Homog T;
    for (int k = 0; k < 9; k++) T.h[k] = H9[k];
    const dim3 grid((d_col + 64 * PXT - 1) / (64 * PXT), (d_row + 3) / 4);
    warp_perspective_kernel<<<grid, dim3(64, 4), 0, st>>>(d_src, s_col, s_row, T, d_dst, d_col, d_row);
}

void launch_warp_cylinder(sift_hip_ctx* ctx, hipStream_t st, const uint8_t* d_src, int w, int h, float focal,
                          uint8_t* d_out, float* d_upedge, float* d_downedge) {
    ProfScope ps(ctx, SIFT_HIP_K_WARP, st);
    const dim3 grid((w + 64 * PXT - 1) / (64 * PXT), (h + 3) / 4);
    warp_cylinder_kernel<<<grid, dim3(64, 4), 0, st>>>(d_src, w, h, focal, d_out, reinterpret_cast<float2*>(d_upedge),
                                                       reinterpret_cast<float2*>(d_downedge));
}

// ---- canvas assembly of blen2img (Blend.cpp:1010-1031) ---------------------------------
// The black canvas receives the first image top0 rows down from its top-left corner, then the
// non-black pixels of the second image at (left1, top1); pixels that fall outside the canvas
// are dropped.  One thread per canvas pixel.
__global__ __launch_bounds__(256) void compose_pair_kernel(const uint8_t* __restrict__ a, int ca, int ra,
                                                           const uint8_t* __restrict__ b, int cb, int rb, int top0,
                                                           int left1, int top1, uint8_t* __restrict__ canvas, int cols,
                                                           int rows) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)cols * rows) return;
    const int y = (int)(e / cols), x = (int)(e - (size_t)y * cols);
    uint8_t v0 = 0, v1 = 0, v2 = 0;
    const int ya = y - top0;
    if (ya >= 0 && ya < ra && x < ca) {
        const uint8_t* s = a + ((size_t)ya * ca + x) * 3;
        v0 = s[0];
        v1 = s[1];
        v2 = s[2];
    }
    const int yb = y - top1, xb = x - left1;
    if (yb >= 0 && yb < rb && xb >= 0 && xb < cb) {
        const uint8_t* s = b + ((size_t)yb * cb + xb) * 3;
        if (s[0] != 0 || s[1] != 0 || s[2] != 0) {
            v0 = s[0];
            v1 = s[1];
            v2 = s[2];
        }
    }
    uint8_t* d = canvas + e * 3;
    d[0] = v0;
    d[1] = v1;
    d[2] = v2;
}

void launch_compose_pair(sift_hip_ctx* ctx, hipStream_t st, const uint8_t* d_a, int ca, int ra, const uint8_t* d_b, int cb,
                         int rb, int top0, int left1, int top1, uint8_t* d_canvas, int cols, int rows) {
    ProfScope ps(ctx, SIFT_HIP_K_WARP, st);
    const size_t n = (size_t)cols * rows;
    compose_pair_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_a, ca, ra, d_b, cb, rb, top0, left1, top1,
                                                                               d_canvas, cols, rows);
}

// ---- ImgRaw's 8-bit copy (Imgraw.hpp:59-82) ---------------------------------------------
// (unsigned char)(v * 255.0) for images that hold [0,1] samples, (unsigned char)v otherwise: the product in double,
// truncated towards zero, its low byte kept (what the conversion compiles to on x86 for values an int holds).  Four
// samples per thread.
__global__ __launch_bounds__(256) void float_to_u8_kernel(const float* __restrict__ src, size_t n, int normalised,
                                                          uint8_t* __restrict__ dst) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const double scale = normalised ? 255.0 : 1.0;
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        uchar4 o;
        o.x = (uint8_t)(int)((double)v.x * scale);
        o.y = (uint8_t)(int)((double)v.y * scale);
        o.z = (uint8_t)(int)((double)v.z * scale);
        o.w = (uint8_t)(int)((double)v.w * scale);
        *reinterpret_cast<uchar4*>(dst + i) = o;
    } else {
        for (size_t k = i; k < n; ++k) dst[k] = (uint8_t)(int)((double)src[k] * scale);
    }
}

void launch_float_to_u8(sift_hip_ctx* ctx, hipStream_t st, const float* d_src, size_t n, bool normalised, uint8_t* d_dst) {
    if (n == 0) return;
    ProfScope ps(ctx, SIFT_HIP_K_WARP, st);
    const size_t threads = (n + 3) / 4;
    float_to_u8_kernel<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(d_src, n, normalised ? 1 : 0, d_dst);
}

}  // namespace sifthip
