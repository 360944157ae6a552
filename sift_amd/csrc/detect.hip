// Keypoint detection for one octave on gfx950:
//   detect_tile   3x3x3 extremum + contrast + edge test on LDS-staged DoG tiles
//                 (all layers); wavefront ballots become 64-pixel mask words
//   top_check     the top layer's comparison with the layer above it, whose
//                 Gaussian level is evaluated only at the pixels that need it
//   scan_rows     exclusive prefix sum of the per-row counts (one workgroup)
//   expand_rows   mask words -> ordered candidate list (wave prefix sums)
//   refine        optional sub-pixel stage (interp_extremum), one lane per candidate
//   orient        36-bin orientation histogram, one wavefront per candidate
//   emit          candidates x firing bins -> feature records, in reference order
//
// The candidate order (layer, row, column) and the feature order (candidate,
// bin) reproduce the append order of the reference's scan (Sift.cpp:173-200,
// :285-293), so the output array equals its linked list copied out (:207-224).
#include <algorithm>

#include "internal.h"

namespace sifthip {

// The DoG layers are never stored: dog[k] = gau[k] - gau[k+1] (Sift.cpp:161-168) is
// one float subtraction.  A workgroup stages the DoG values of a 64 x DET_TR pixel
// tile (plus a one-pixel halo) for all intvls+2 layers in LDS, reading each
// Gaussian level once, then tests its pixels from LDS.
constexpr int DET_TR = 16;             // tile rows
constexpr int DET_LW = 66;             // tile width incl. halo
constexpr int DET_LH = DET_TR + 2;     // tile rows incl. halo
constexpr int kMaxIntvls = 8;

// ---- findMaxMin (Sift.cpp:86-104) + harris_corner (Sift.cpp:105-129) -------------
// d = LDS DoG tile of layer li centred on the pixel (row stride DET_LW, layer
// stride DET_LH*DET_LW); v = its value, already past |v| > thr.
// upper = false leaves the layer above out of the extremum test (it is checked
// later, only for the pixels that pass everything else: see top_check_kernel).
__device__ __forceinline__ bool is_keypoint(const float* d, float v, float rad, bool upper) {
    constexpr int LS = DET_LH * DET_LW;
    // "no neighbour is larger" (v > 0) / "no neighbour is smaller" (v < 0), ties
    // allowed, as max/min over the 27 values (the centre itself changes nothing)
    float mx = v, mn = v;
#pragma unroll
    for (int k = -1; k <= 1; k++) {
        if (k == 1 && !upper) break;
#pragma unroll
        for (int j = -1; j <= 1; j++) {
            const float a = d[k * LS + j * DET_LW - 1], b = d[k * LS + j * DET_LW], c = d[k * LS + j * DET_LW + 1];
            mx = fmaxf(mx, fmaxf(a, fmaxf(b, c)));
            mn = fminf(mn, fminf(a, fminf(b, c)));
        }
    }
    if (v > 0.f ? (mx > v) : (mn < v)) return false;
    // edge test; the "r" of the reference call is the histogram radius
    const float thre = ((rad + 1.f) * (rad + 1.f)) / rad;
    const float Dxx = 2.f * v - d[-1] - d[1];
    const float Dyy = 2.f * v - d[-DET_LW] - d[DET_LW];
    float Dxy = d[DET_LW + 1] + d[-DET_LW - 1] - d[-DET_LW + 1] - d[DET_LW - 1];
    Dxy /= 4.f;
    const float Tr = Dxx + Dyy;
    const float Det = Dxx * Dyy - Dxy * Dxy;
    const float val = (Tr * Tr / Det);
    return val < thre;
}

// Neighbouring lanes' values (DPP wavefront shifts).
__device__ __forceinline__ float from_lane_below(float v) {  // lane i receives lane i-1 (lane 0: unspecified)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_lane_above(float v) {  // lane i receives lane i+1 (lane 63: unspecified)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// The tests of four consecutive tile rows (ry0 .. ry0+3) by one wave, three scales per octave.
// Reading the 27 neighbours of every pixel and layer from LDS is what bounded the kernel (its LDS
// pipe was busy longer than its loads take); here every staged value is read once: the wave walks
// down the six rows involved, lane = column, the left and right neighbours come from the adjacent
// lanes (lanes 0 and 63 read the halo column), and the 3 x 3 x 3 maximum is built from per-row
// 3-maxima kept in registers: per DoG layer max3(left, centre, right) of a row, max3 over the three
// rows, max3 over the three layers.  The values and neighbours of the tested layers stay in
// registers for the edge test.  Same comparisons and the same edge arithmetic as is_keypoint.
template <int NLAY>
__device__ __forceinline__ void detect_four_rows(const float* s_dog, int ry0, int lane, int x, int y0, int W, int H, int tx,
                                                 int WW, const LayerConsts* __restrict__ lc, float thr,
                                                 unsigned long long* __restrict__ masks, int* __restrict__ row_cnt) {
    constexpr int LS = DET_LH * DET_LW;
    constexpr bool lazy = NLAY == 4;
    int rad[3];
#pragma unroll
    for (int l = 0; l < 3; l++) rad[l] = lc[l].radius;
    // Most strips of a smooth image hold no pixel above the contrast bar: look at that first (one read and
    // one compare per pixel and layer) and leave with empty mask words if so.
    {
        unsigned long long any = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = y0 + ry0 + i;
            if (y < H) {
#pragma unroll
                for (int l = 0; l < 3; l++) {
                    const float v = s_dog[(l + 1) * LS + (ry0 + 1 + i) * DET_LW + lane + 1];
                    const int rd = rad[l];
                    const bool hit = (H >= 2 * rd + 3) && (W >= 2 * rd + 3) && y >= rd + 1 && y < H - rd - 1 && x >= rd + 1 &&
                                     x < W - rd - 1 && (fabsf(v) > thr);
                    any |= __ballot(hit);
                }
            }
        }
        if (any == 0) {
            if (lane < 12) {
                const int i = lane / 3, l = lane - 3 * i, y = y0 + ry0 + i;
                if (y < H) masks[((size_t)l * H + y) * WW + tx] = 0ull;
            }
            return;
        }
    }
    float C[3][3], L[3][3], R[3][3];        // tested layers: value, left, right of the last three rows
    float hmax[NLAY][3], hmin[NLAY][3];     // per layer and row: max / min of (left, centre, right)
    const int hcol = lane == 0 ? 0 : DET_LW - 1;
#pragma unroll
    for (int i = 0; i < 6; i++) {  // LDS row ry0 + i; slot = i mod 3
        const int s = i % 3;
        const float* row = s_dog + (ry0 + i) * DET_LW;
#pragma unroll
        for (int k = 0; k < NLAY; k++) {
            const float d = row[k * LS + lane + 1];
            const float hd = row[k * LS + hcol];
            float l = from_lane_below(d), r = from_lane_above(d);
            l = lane == 0 ? hd : l;
            r = lane == 63 ? hd : r;
            hmax[k][s] = fmaxf(fmaxf(l, d), r);
            hmin[k][s] = fminf(fminf(l, d), r);
            if (k >= 1 && k <= 3) {
                C[k - 1][s] = d;
                L[k - 1][s] = l;
                R[k - 1][s] = r;
            }
        }
        if (i < 2) continue;
        const int y = y0 + ry0 + i - 2;  // centre row: LDS row ry0 + i - 1
        if (y >= H) break;               // wave-uniform
        const int sc = (s + 2) % 3, su = (s + 1) % 3, sd = s;  // centre, row above, row below
        float vmax[NLAY], vmin[NLAY];
#pragma unroll
        for (int k = 0; k < NLAY; k++) {
            vmax[k] = fmaxf(fmaxf(hmax[k][0], hmax[k][1]), hmax[k][2]);
            vmin[k] = fminf(fminf(hmin[k][0], hmin[k][1]), hmin[k][2]);
        }
#pragma unroll
        for (int l = 0; l < 3; l++) {
            const float v = C[l][sc];
            const int rd = rad[l];
            // scan bounds of Sift.cpp:178-179: r+1 <= y < H-r-1, r+1 <= x < W-r-1
            bool hit = (H >= 2 * rd + 3) && (W >= 2 * rd + 3) && y >= rd + 1 && y < H - rd - 1 && x >= rd + 1 &&
                       x < W - rd - 1 && (fabsf(v) > thr);
            // layers l, l+1 (the tested one), l+2; the last is left out when it does not exist
            float mx = fmaxf(vmax[l], vmax[l + 1]), mn = fminf(vmin[l], vmin[l + 1]);
            if (!(lazy && l == 2)) {
                mx = fmaxf(mx, vmax[l + 2 < NLAY ? l + 2 : l]);
                mn = fminf(mn, vmin[l + 2 < NLAY ? l + 2 : l]);
            }
            if (hit) hit = !(v > 0.f ? (mx > v) : (mn < v));
            if (hit) {
                // harris_corner (Sift.cpp:105-129) with the reference's argument, the histogram radius
                const float rf = (float)rd;
                const float thre = ((rf + 1.f) * (rf + 1.f)) / rf;
                const float Dxx = 2.f * v - L[l][sc] - R[l][sc];
                const float Dyy = 2.f * v - C[l][su] - C[l][sd];
                float Dxy = R[l][sd] + L[l][su] - R[l][su] - L[l][sd];
                Dxy /= 4.f;
                const float Tr = Dxx + Dyy;
                const float Det = Dxx * Dyy - Dxy * Dxy;
                const float val = (Tr * Tr / Det);
                hit = val < thre;
            }
            const unsigned long long m = __ballot(hit);
            if (lane == 0) {
                masks[((size_t)l * H + y) * WW + tx] = m;
                if (m) atomicAdd(&row_cnt[l * H + y], __popcll(m));
            }
        }
    }
}

// grid = (ceil(W/64), ceil(H/DET_TR)).  Writes every mask word of its tile (zeros
// included) and adds the per-(layer,row) counts into row_cnt (zeroed beforehand).
// NLAY = number of DoG layers staged when known at compile time, else 0.  With
// lazy_top the last Gaussian level does not exist: intvls + 1 DoG layers are
// staged and the top layer's marks are tentative (its upper neighbours untested).
template <int NLAY>
__global__ __launch_bounds__(256) void detect_tile_kernel(const float* __restrict__ gau, size_t PX, int W, int H,
                                                          int intvls, const LayerConsts* __restrict__ lc, float thr,
                                                          int WW, unsigned long long* __restrict__ masks,
                                                          int* __restrict__ row_cnt, int lazy_top) {
    extern __shared__ float s_dog[];  // [nlay][DET_LH][DET_LW]
    constexpr int LS = DET_LH * DET_LW;
    // Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.
    // Remap so that one XCD owns a contiguous run of tiles: the halo columns and
    // rows shared with the neighbouring tiles are then L2 hits instead of fabric
    // reads (bijective for any tile count).
    const int ntile = WW * gridDim.y;
    const int bid = blockIdx.y * WW + blockIdx.x;
    const int q8 = ntile >> 3, r8 = ntile & 7, xcd = bid & 7;
    const int tid_lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tx = tid_lin % WW, ty = tid_lin / WW;
    const int x0 = tx * 64, y0 = ty * DET_TR;
    const int nlay = intvls + 2 - lazy_top;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // one wavefront per tile row: 64 coalesced columns + the two halo columns
        // on lanes 0 and 1; every level is read once
        const int gx = min(x0 + lane, W - 1);
        const int hx = lane == 0 ? max(x0 - 1, 0) : min(x0 + 64, W - 1);
        const int hcol = lane == 0 ? 0 : DET_LW - 1;
        if (NLAY > 0 && PX < ((size_t)1 << 30)) {
            // All loads of the tile in flight at once.  Buffer loads: one resource per level, the row as a
            // scalar byte offset, the column as a lane offset that is computed once, so a load costs no
            // vector instruction; the halo columns are fetched by every lane (lanes 2..63 repeat lane 1's
            // address and write to a spare word behind the tile) instead of under an execution mask.
            constexpr int NL = NLAY > 0 ? NLAY : 1;
            constexpr int NR = (DET_LH + 3) / 4;
            __amdgpu_buffer_rsrc_t lvl[NL + 1];
#pragma unroll
            for (int k = 0; k <= NL; k++)
                lvl[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gau + (size_t)k * PX), 0, (unsigned)(PX * 4), 0x00020000);
            const int vx = gx * 4;
            const int vh = (lane == 0 ? hx : min(x0 + 64, W - 1)) * 4;
            const int uwave = __builtin_amdgcn_readfirstlane(wave);
            float v[NR][NL + 1], hv[NR][NL + 1];
#pragma unroll
            for (int i = 0; i < NR; i++) {
                const int r = uwave + 4 * i;
                const int gy = min(max(y0 + r - 1, 0), H - 1);
                const int so = (int)((unsigned)gy * (unsigned)W * 4u);  // unsigned product: below 2^32, PX * 4 is
#pragma unroll
                for (int k = 0; k <= NL; k++) {
                    v[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lvl[k], vx, so, 0));
                    hv[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lvl[k], vh, so, 0));
                }
            }
            const int spare = nlay * LS;                                    // one word behind the staged layers
            const int hw = lane < 2 ? hcol : spare;                         // where this lane's halo value goes
#pragma unroll
            for (int i = 0; i < NR; i++) {
                const int r = uwave + 4 * i;
                if (r < DET_LH) {
                    float* dst = s_dog + r * DET_LW;
#pragma unroll
                    for (int k = 0; k < NL; k++) {
                        dst[k * LS + lane + 1] = v[i][k] - v[i][k + 1];
                        s_dog[lane < 2 ? k * LS + r * DET_LW + hw : hw] = hv[i][k] - hv[i][k + 1];
                    }
                }
            }
        } else if (NLAY > 0) {
            // (levels of 4 GB and more: plain 64-bit addressing)
            constexpr int NL = NLAY > 0 ? NLAY : 1;
            constexpr int NR = (DET_LH + 3) / 4;
            float v[NR][NL + 1], hv[NR][NL + 1];
#pragma unroll
            for (int i = 0; i < NR; i++) {
                const int r = wave + 4 * i;
                const int gy = min(max(y0 + r - 1, 0), H - 1);
                const float* row = gau + (size_t)gy * W;
#pragma unroll
                for (int k = 0; k <= NL; k++) {
                    v[i][k] = row[(size_t)k * PX + gx];
                    hv[i][k] = lane < 2 ? row[(size_t)k * PX + hx] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < NR; i++) {
                const int r = wave + 4 * i;
                if (r < DET_LH) {
                    float* dst = s_dog + r * DET_LW;
#pragma unroll
                    for (int k = 0; k < NL; k++) {
                        dst[k * LS + lane + 1] = v[i][k] - v[i][k + 1];
                        if (lane < 2) dst[k * LS + hcol] = hv[i][k] - hv[i][k + 1];
                    }
                }
            }
        } else {
            for (int r = wave; r < DET_LH; r += 4) {
                const int gy = min(max(y0 + r - 1, 0), H - 1);
                const float* row = gau + (size_t)gy * W;
                float* dst = s_dog + r * DET_LW;
                float a = row[gx];
                float ha = lane < 2 ? row[hx] : 0.f;
                for (int k = 0; k < nlay; k++) {
                    row += PX;
                    const float b = row[gx];
                    dst[k * LS + lane + 1] = a - b;
                    a = b;
                    if (lane < 2) {
                        const float hb = row[hx];
                        dst[k * LS + hcol] = ha - hb;
                        ha = hb;
                    }
                }
            }
        }
    }
    __syncthreads();
    const int x = x0 + lane;
    if (NLAY > 0 && intvls == 3) {
        static_assert(DET_TR == 16, "four waves of four rows");
        if (y0 + 4 * wave < H)
            detect_four_rows<(NLAY > 0 ? NLAY : 4)>(s_dog, 4 * wave, lane, x, y0, W, H, tx, WW, lc, thr, masks, row_cnt);
        return;
    }
    for (int ry = wave; ry < DET_TR; ry += 4) {
        const int y = y0 + ry;
        if (y >= H) break;  // wave-uniform
        for (int l = 0; l < intvls; l++) {
            const int rad = lc[l].radius;
            const float* d = s_dog + (l + 1) * LS + (ry + 1) * DET_LW + (lane + 1);
            const float v = *d;
            // scan bounds of Sift.cpp:178-179: r+1 <= y < H-r-1, r+1 <= x < W-r-1
            bool hit = (H >= 2 * rad + 3) && (W >= 2 * rad + 3) && y >= rad + 1 && y < H - rad - 1 && x >= rad + 1 &&
                       x < W - rad - 1 && (fabsf(v) > thr);
            if (hit) hit = is_keypoint(d, v, (float)rad, !(lazy_top && l == intvls - 1));
            const unsigned long long m = __ballot(hit);
            if (lane == 0) {
                masks[((size_t)l * H + y) * WW + tx] = m;
                if (m) atomicAdd(&row_cnt[l * H + y], __popcll(m));
            }
        }
    }
}

// Exclusive scan of n ints by one 1024-thread workgroup; total -> counters.
__device__ __forceinline__ int block_excl_scan_1024(int v, int* s_warp, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_warp[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int w = (lane < 16) ? s_warp[lane] : 0;
        int winc = w;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int t = __shfl_up(winc, o);
            if (lane >= o) winc += t;
        }
        if (lane < 16) s_warp[16 + lane] = winc - w;  // exclusive wave offsets
        if (lane == 15) s_warp[32] = winc;
    }
    __syncthreads();
    total = s_warp[32];
    const int res = s_warp[16 + wave] + inc - v;
    __syncthreads();
    return res;
}

__global__ __launch_bounds__(1024) void scan_rows_kernel(const int* __restrict__ row_cnt, int* __restrict__ row_off,
                                                         int nrows, int cand_cap, int* __restrict__ counters,
                                                         int* __restrict__ found_slot) {
    __shared__ int s_warp[40];
    // thread t owns the rows [t*C, (t+1)*C): one workgroup scan for any row count
    const int C = (nrows + 1023) / 1024;
    const int r0 = threadIdx.x * C, r1 = min(r0 + C, nrows);
    int mine = 0;
    for (int i = r0; i < r1; i++) mine += row_cnt[i];
    int total;
    int run = block_excl_scan_1024(mine, s_warp, total);
    for (int i = r0; i < r1; i++) {
        row_off[i] = run;
        run += row_cnt[i];
    }
    if (threadIdx.x == 0) {
        counters[CNT_NCAND_TOTAL] = total;
        counters[CNT_NCAND] = min(total, cand_cap);
        *found_slot = total;  // written on every frame; the host compares it with the table size
    }
}

// One wavefront per row: turn its mask words into candidates at row_off[row]...
__global__ __launch_bounds__(64) void expand_rows_kernel(const unsigned long long* __restrict__ masks,
                                                         const int* __restrict__ row_cnt,
                                                         const int* __restrict__ row_off, int H, int WW,
                                                         Cand* __restrict__ cand, int cand_cap) {
    const int row = blockIdx.x;
    if (row_cnt[row] == 0) return;
    const int lane = threadIdx.x;
    const int li = row / H;
    const int y = row - li * H;
    int running = row_off[row];
    for (int wb = 0; wb < WW; wb += 64) {
        const int word = wb + lane;
        unsigned long long m = (word < WW) ? masks[(size_t)row * WW + word] : 0ull;
        const int c = __popcll(m);
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        int pos = running + inc - c;
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (pos < cand_cap) cand[pos] = Cand{word * 64 + bit, y, li + 1};
            pos++;
        }
        running += __shfl(inc, 63);
    }
}

// ---- the last Gaussian level, only where it is looked at --------------------------
// gau[intvls+2] feeds nothing but dog[intvls+1], and that layer is only ever
// compared with the 3x3 neighbourhood of a top-layer pixel that has already passed
// the contrast, same-layer, lower-layer and edge tests: a few thousand pixels
// instead of every pixel of the most expensive blur (the longest kernel).  For each
// such pixel a workgroup recomputes the level at its 3x3 neighbourhood exactly as
// the two blur passes would (row pass values of the K+2 rows involved, rounded to
// float, then the column pass; same products, same order, same edge clamping),
// finishes the extremum test and clears the pixel's mask bit if it fails.
__global__ __launch_bounds__(256) void top_check_kernel(const float* __restrict__ g_lo, const float* __restrict__ g_hi,
                                                        int W, int H, const float* __restrict__ taps, int K,
                                                        const Cand* __restrict__ cand2,
                                                        const int* __restrict__ counters2, int WW,
                                                        unsigned long long* __restrict__ masks_top,
                                                        int* __restrict__ row_cnt_top) {
    extern __shared__ float s_tmp[];  // [3][K + 2] row-pass values, then the K taps
    __shared__ int s_bad;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int r5 = K >> 1;
    const int nt = K + 2;
    const int ncand = counters2[CNT_NCAND];
    if ((int)blockIdx.x >= ncand) return;
    float* s_taps = s_tmp + 3 * nt;
    for (int k = tid; k < K; k += nthr) s_taps[k] = taps[k];
    __syncthreads();
    for (int ci = blockIdx.x; ci < ncand; ci += gridDim.x) {
        const Cand cd = cand2[ci];
        const int x = cd.x, y = cd.y;
        if (tid == 0) s_bad = 0;
        // Row pass: thread t owns input row y-1-r5+t (clamped as the blur clamps) and
        // produces its three outputs at columns x-1, x, x+1 from one sweep over the
        // K+2 inputs they share.
        const int c0 = x - 1 - r5;  // column of input offset 0
        const bool inside = c0 >= 0 && c0 + K + 2 <= W;
        for (int t = tid; t < nt; t += nthr) {
            const int row = min(max(y - 1 - r5 + t, 0), H - 1);
            const float* src = g_hi + (size_t)row * W;
            double a0 = 0, a1 = 0, a2 = 0;
            if (K <= 62 && c0 >= 0 && c0 + 64 <= W) {
                // Short kernels (octave 0: 55 taps): the whole row of inputs in one burst of 16-byte loads
                // before the first use -- the sweep then costs one trip to memory, where the loop below pays one
                // per 16 taps and one per tap of its remainder (a candidate's time is nothing but such trips).
                typedef float top_f4u __attribute__((ext_vector_type(4), aligned(4)));
                const top_f4u* sp4 = reinterpret_cast<const top_f4u*>(src + c0);
                float v[64];
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const top_f4u q = sp4[j];
                    v[4 * j] = q.x;
                    v[4 * j + 1] = q.y;
                    v[4 * j + 2] = q.z;
                    v[4 * j + 3] = q.w;
                }
#pragma unroll
                for (int k = 0; k < 62; k++) {
                    if (k < K) {  // uniform
                        const float g = s_taps[k];
                        const float p0 = v[k] * g, p1 = v[k + 1] * g, p2 = v[k + 2] * g;
                        a0 += (double)p0;
                        a1 += (double)p1;
                        a2 += (double)p2;
                    }
                }
            } else if (inside) {
                const float* sp = src + c0;
                float v0 = sp[0], v1 = sp[1];
#pragma unroll 16
                for (int k = 0; k < K; k++) {
                    const float v2 = sp[k + 2];
                    const float g = s_taps[k];
                    const float p0 = v0 * g, p1 = v1 * g, p2 = v2 * g;
                    a0 += (double)p0;
                    a1 += (double)p1;
                    a2 += (double)p2;
                    v0 = v1;
                    v1 = v2;
                }
            } else {
                auto ld = [&](int j) { return src[min(max(c0 + j, 0), W - 1)]; };
                float v0 = ld(0), v1 = ld(1);
#pragma unroll 16
                for (int k = 0; k < K; k++) {
                    const float v2 = ld(k + 2);
                    const float g = s_taps[k];
                    const float p0 = v0 * g, p1 = v1 * g, p2 = v2 * g;
                    a0 += (double)p0;
                    a1 += (double)p1;
                    a2 += (double)p2;
                    v0 = v1;
                    v1 = v2;
                }
            }
            s_tmp[t] = (float)a0;
            s_tmp[nt + t] = (float)a1;
            s_tmp[2 * nt + t] = (float)a2;
        }
        __syncthreads();
        // Column pass at the nine positions, then the part of findMaxMin that was left out.
        if (tid < 9) {
            const int c = tid % 3, oyi = tid / 3;
            const float* col = s_tmp + c * nt + oyi;
            double acc = 0;
#pragma unroll 8
            for (int k = 0; k < K; k++) {
                const float prod = col[k] * s_taps[k];  // LDS reads run ahead of the ordered additions
                acc += (double)prod;
            }
            const size_t p = (size_t)(y - 1 + oyi) * W + (x - 1 + c);
            const float up = g_hi[p] - (float)acc;  // dog[top+1] at this neighbour
            const size_t pc = (size_t)y * W + x;
            const float v = g_lo[pc] - g_hi[pc];    // dog[top] at the pixel
            if ((v > 0.f && v < up) || (v < 0.f && v > up)) s_bad = 1;
        }
        __syncthreads();
        if (tid == 0 && s_bad) {
            atomicAnd(&masks_top[(size_t)y * WW + (x >> 6)], ~(1ull << (x & 63)));
            atomicSub(&row_cnt_top[y], 1);
        }
        __syncthreads();
    }
}

// ---- interp_extremum, matrix.hpp:140-196 (optional stage) -------------------------
// Lowe's sub-pixel refinement, one lane per candidate, expression by expression as
// the (never compiled) reference header states it: x = -H^-1 * dD from central
// differences of the DoG stack (deriv_3D :12-33, hessian_3D :34-67, the closed-form
// inverse :68-86), at most 5 steps, border 5, contrast test on the interpolated
// value.  The reference's only call (Sift.cpp:187) is commented out and mixes up
// its row/column arguments; here it runs as its signature intends.  A rejected
// candidate gets layer 0, an accepted one moves to the refined (layer, row,
// column) and must still leave room for that layer's orientation window (the
// scan bounds of Sift.cpp:178-179).  Offsets that are not finite or exceed 1e9
// (singular Hessian) reject the point; the reference's int conversion of such a
// value is undefined.
__global__ __launch_bounds__(256) void refine_kernel(const float* __restrict__ gau, size_t PX, int W, int H,
                                                     int intvls, const LayerConsts* __restrict__ lc,
                                                     float contr_thr, Cand* __restrict__ cand,
                                                     float4* __restrict__ cand_sub,
                                                     const int* __restrict__ counters) {
    const int ncand = counters[CNT_NCAND];
    for (int ci = blockIdx.x * 256 + threadIdx.x; ci < ncand; ci += gridDim.x * 256) {
        const Cand cd = cand[ci];
        int li = cd.layer, r = cd.y, c = cd.x;
        // dog[l] = gau[l] - gau[l+1]
        auto D = [&](int l, int rr, int cc) -> float {
            const size_t p = (size_t)l * PX + (size_t)rr * W + cc;
            return gau[p] - gau[p + PX];
        };
        float xi = 0.f, xr = 0.f, xc = 0.f;
        float dD0 = 0.f, dD1 = 0.f, dD2 = 0.f;
        bool ok = true;
        int i = 0;
        while (i < 5) {
            const float v = D(li, r, c);
            const float xp = D(li, r, c + 1), xm = D(li, r, c - 1);
            const float yp = D(li, r + 1, c), ym = D(li, r - 1, c);
            const float sp = D(li + 1, r, c), sm = D(li - 1, r, c);
            dD0 = (xp - xm) / 2.f;
            dD1 = (yp - ym) / 2.f;
            dD2 = (sp - sm) / 2.f;
            const float dxx = (xp + xm - 2.f * v);
            const float dyy = (yp + ym - 2.f * v);
            const float dss = (sp + sm - 2.f * v);
            const float dxy = (D(li, r + 1, c + 1) - D(li, r + 1, c - 1) - D(li, r - 1, c + 1) + D(li, r - 1, c - 1)) / 4.f;
            const float dxs = (D(li + 1, r, c + 1) - D(li + 1, r, c - 1) - D(li - 1, r, c + 1) + D(li - 1, r, c - 1)) / 4.f;
            const float dys = (D(li + 1, r + 1, c) - D(li + 1, r - 1, c) - D(li - 1, r + 1, c) + D(li - 1, r - 1, c)) / 4.f;
            // H = [dxx dxy dxs; dxy dyy dys; dxs dys dss], inverse by cofactors
            const float v0 = dxx, v1 = dxy, v2 = dxs, v3 = dxy, v4 = dyy, v5 = dys, v6 = dxs, v7 = dys, v8 = dss;
            float det = v0 * (v4 * v8 - v5 * v7);
            det -= v3 * (v1 * v8 - v2 * v7);
            det += v6 * (v1 * v5 - v2 * v4);
            const float m0 = (v4 * v8 - v5 * v7) / det;
            const float m1 = -(v1 * v8 - v2 * v7) / det;
            const float m2 = (v1 * v5 - v2 * v4) / det;
            const float m3 = -(v3 * v8 - v5 * v6) / det;
            const float m4 = (v0 * v8 - v2 * v6) / det;
            const float m5 = -(v0 * v5 - v2 * v3) / det;
            const float m6 = (v3 * v7 - v4 * v6) / det;
            const float m7 = -(v0 * v7 - v1 * v6) / det;
            const float m8 = (v0 * v4 - v1 * v3) / det;
            xc = (m0 * dD0 + m1 * dD1 + m2 * dD2) * -1.f;
            xr = (m3 * dD0 + m4 * dD1 + m5 * dD2) * -1.f;
            xi = (m6 * dD0 + m7 * dD1 + m8 * dD2) * -1.f;
            if (!(fabsf(xi) < 1e9f && fabsf(xr) < 1e9f && fabsf(xc) < 1e9f)) {
                ok = false;
                break;
            }
            if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
            c += (int)roundf(xc);
            r += (int)roundf(xr);
            li += (int)roundf(xi);
            if (li < 1 || li > intvls || c < 5 || c >= W - 5 || r < 5 || r >= H - 5) {
                ok = false;
                break;
            }
            i++;
        }
        if (ok && i >= 5) ok = false;
        float contr = 0.f;
        if (ok) {
            // interp_contr, matrix.hpp:113-124 (the derivative at the final point is the last one computed)
            const float t = dD0 * xc + dD1 * xr + dD2 * xi;
            contr = D(li, r, c) + t * 0.5f;
            if (fabsf(contr) < contr_thr) ok = false;
        }
        if (ok) {
            const int rad = lc[li - 1].radius;
            ok = (H >= 2 * rad + 3) && (W >= 2 * rad + 3) && r >= rad + 1 && r < H - rad - 1 && c >= rad + 1 &&
                 c < W - rad - 1;
        }
        Cand o;
        o.x = c;
        o.y = r;
        o.layer = ok ? li : 0;
        cand[ci] = o;
        cand_sub[ci] = make_float4(xc, xr, xi, contr);
    }
}

// ---- fastAtan2f, Imgraw.cpp:29-52 (degrees) ------------------------------------
__device__ __forceinline__ float fast_atan2f_deg(float dy, float dx) {
    const float k = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k;
    const float p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k;
    const float p7 = -0.04432655554792128f * k;
    const float eps = (float)2.2204460492503131e-016;
    const float ax = fabsf(dx), ay = fabsf(dy);
    // both branches of the reference evaluate the same polynomial in min/(max + eps);
    // selecting the operands first leaves one division for the whole wavefront
    const bool xmaj = ax >= ay;
    const float c = (xmaj ? ay : ax) / ((xmaj ? ax : ay) + eps);
    const float c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (!xmaj) a = 90.f - a;
    if (dx < 0) a = 180.f - a;
    if (dy < 0) a = 360.f - a;
    return a;
}

// ---- Sift::getHistogramMS, Sift.cpp:250-295 ---------------------------------------
// One wavefront per candidate.  The reference adds the weighted magnitudes into
// the 36 bins one window pixel after another (row-major); float addition is not
// associative and the ">= 0.8 * max" test decides how many features a keypoint
// becomes, so the order of the additions is kept.  The window is numbered row by row with
// its rows padded to a multiple of 4 slots (padding slots belong to no bin), and per pass of
// ORI_CH slots:
//   A  lane l evaluates magnitude and bin of the ORI_RUN consecutive slots RUN*l..RUN*l+RUN-1,
//      two groups of 4 consecutive pixels of one window row: the 14 values of each of the two
//      Gaussian levels that their 4 DoG gradients need arrive in four loads of 16 / 8 bytes
//      instead of 8 scalar loads per pixel, and the results stay in the lane's registers
//   B  a stable counting sort by bin: lane l counts its run per bin with LDS atomics, lane b
//      turns row b of the (bin x lane) count matrix into running offsets, and the pixels are
//      scattered to (bin, lane, rank) order -- window order within each bin
//   C  lane b adds the values of bin b front to back.
// Lane b touches only its own ~1/36 of the pixels instead of testing every pixel
// against its bin number.
constexpr int ORI_CU = 16;  // LDS reads in flight in the ordered sums (4: 110 us for octave 0 of the 4K frame, 8: 107, 16: 104)
constexpr int ORI_CH = 512;                 // window slots per pass (1024 and 2048 measured slower: fewer resident waves)
constexpr int ORI_RUN = ORI_CH / 64;       // consecutive slots a lane owns
constexpr int ORI_ROW = 33;  // dwords per count-matrix row: 32 lane pairs (two u16 each) + 1 pad
static_assert(ORI_RUN == 8, "a lane's run is two groups of 4 pixels");

// four / two floats at any 4-byte aligned address (one load: global memory takes unaligned accesses)
typedef float ori_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float ori_f2u __attribute__((ext_vector_type(2), aligned(4)));

// floor(th / 10) for th in [0, 360]: what (size_t)(sita / 10.0) yields in double
// arithmetic (th / 10.0 is never within rounding distance of an integer unless th
// is an exact multiple of 10), computed with exact float comparisons instead of a
// double division.
__device__ __forceinline__ int angle_bin(float th) {
    int b = (int)(th * 0.1f);
    if ((float)(10 * (b + 1)) <= th) b++;
    if ((float)(10 * b) > th) b--;
    return b;
}

__global__ __launch_bounds__(64) void orient_kernel(const float* __restrict__ gau, size_t PX, int W,
                                                    const LayerConsts* __restrict__ lc,
                                                    const float* __restrict__ table, const Cand* __restrict__ cand,
                                                    const int* __restrict__ counters,
                                                    unsigned long long* __restrict__ cmask, float* __restrict__ cmm) {
    __shared__ __attribute__((aligned(16))) unsigned s_cnt[37 * ORI_ROW + 59];  // 1280 dwords
    __shared__ float s_sorted[ORI_CH];
    __shared__ int s_base[36];
    const int lane = threadIdx.x;
    const int ncand = counters[CNT_NCAND];
    const unsigned half_shift = (lane & 1) * 16;
    for (int ci = blockIdx.x; ci < ncand; ci += gridDim.x) {
        const Cand cd = cand[ci];
        // layer 0 = rejected by the sub-pixel stage: an empty window and no firing bin
        // (an early `continue` here cost the whole loop 18 %)
        const bool live = cd.layer != 0;
        const LayerConsts L = lc[max(cd.layer, 1) - 1];
        const int r = L.radius;
        const int n = 2 * r + 1;
        const int n4 = (n + 3) & ~3;  // slots per window row
        const int total = live ? n * n4 : 0;
        const float* g1 = gau + (size_t)cd.layer * PX;  // dog[layer] = gau[layer] - gau[layer+1]
        const float* g2 = g1 + PX;
        const float* g = table + L.wts_off;
        // window position (row jj, slot ii) of this lane's first slot, advanced by ORI_CH slots per pass
        const int dj = ORI_CH / n4, di = ORI_CH - dj * n4;
        int jj = (ORI_RUN * lane) / n4;
        int ii = ORI_RUN * lane - jj * n4;
        float sum = 0.f;
        for (int e0 = 0; e0 < total; e0 += ORI_CH) {
            const int cnt = min(ORI_CH, total - e0);  // a multiple of 4
            const int run0 = lane * ORI_RUN;
            // ---- A: magnitude and bin of this lane's run (all 16 loads are issued before any is used)
            ori_f4u xa[2][2], up[2][2], dn[2][2], wi[2];
            ori_f2u xb[2][2];
            float wj[2];
            int gi[2];
            bool in[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                int gj = jj;
                gi[q] = ii + 4 * q;
                if (gi[q] >= n4) {
                    gi[q] -= n4;
                    gj++;
                }
                in[q] = run0 + 4 * q < cnt;
                if (in[q]) {
                    const unsigned p = (unsigned)((cd.y - r + gj) * W + (cd.x - r + gi[q]));
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const float* lv = k ? g2 : g1;
                        xa[q][k] = *reinterpret_cast<const ori_f4u*>(lv + p - 1u);
                        xb[q][k] = *reinterpret_cast<const ori_f2u*>(lv + p + 3u);
                        up[q][k] = *reinterpret_cast<const ori_f4u*>(lv + p - (unsigned)W);
                        dn[q][k] = *reinterpret_cast<const ori_f4u*>(lv + p + (unsigned)W);
                    }
                    wi[q] = *reinterpret_cast<const ori_f4u*>(g + gi[q]);
                    wj[q] = g[gj];
                }
            }
            float val[ORI_RUN];
            unsigned bin[ORI_RUN];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                // DoG values of the row: columns gi-1 .. gi+4
                const float x1[6] = {xa[q][0].x, xa[q][0].y, xa[q][0].z, xa[q][0].w, xb[q][0].x, xb[q][0].y};
                const float x2[6] = {xa[q][1].x, xa[q][1].y, xa[q][1].z, xa[q][1].w, xb[q][1].x, xb[q][1].y};
                const float u1[4] = {up[q][0].x, up[q][0].y, up[q][0].z, up[q][0].w};
                const float u2[4] = {up[q][1].x, up[q][1].y, up[q][1].z, up[q][1].w};
                const float d1[4] = {dn[q][0].x, dn[q][0].y, dn[q][0].z, dn[q][0].w};
                const float d2[4] = {dn[q][1].x, dn[q][1].y, dn[q][1].z, dn[q][1].w};
                const float wv[4] = {wi[q].x, wi[q].y, wi[q].z, wi[q].w};
                float dogx[6];
#pragma unroll
                for (int k = 0; k < 6; k++) dogx[k] = x1[k] - x2[k];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int t = 4 * q + e;
                    val[t] = 0.f;
                    bin[t] = 36u;  // padding slots and slots past the window: no bin
                    if (in[q] && gi[q] + e < n) {
                        const float dx = dogx[e + 2] - dogx[e];
                        const float dy = (d1[e] - d2[e]) - (u1[e] - u2[e]);
                        float mag = sqrtf(dx * dx + dy * dy);
                        const float th = fast_atan2f_deg(dy, dx);
                        mag = mag * wv[e];
                        mag = mag * wj[q];
                        val[t] = mag;
                        bin[t] = (unsigned)angle_bin(th);  // 36 == "one past the end": dropped
                    }
                }
            }
            ii += di;
            jj += dj;
            if (ii >= n4) {
                ii -= n4;
                jj++;
            }
            {
                uint4* z = reinterpret_cast<uint4*>(s_cnt);
#pragma unroll
                for (int i = 0; i < 5; i++) z[i * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
            // ---- B1: per-(bin, lane) counts of the lane's run
#pragma unroll
            for (int t = 0; t < ORI_RUN; t++)
                if (bin[t] < 36u) atomicAdd(&s_cnt[bin[t] * ORI_ROW + (lane >> 1)], 1u << half_shift);
            __syncthreads();
            // ---- B2: lane b turns row b into running offsets (lane order), then the
            // bins' totals into bin bases
            int my_total = 0;
            if (lane < 36) {
                unsigned* rowp = s_cnt + lane * ORI_ROW;
                unsigned run = 0;
#pragma unroll 8
                for (int i = 0; i < 32; i++) {
                    const unsigned wv = rowp[i];
                    const unsigned lo = wv & 0xffffu, hi = wv >> 16;
                    rowp[i] = run | ((run + lo) << 16);
                    run += lo + hi;
                }
                my_total = (int)run;
            }
            int inc = my_total;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            const int my_start = inc - my_total;
            if (lane < 36) s_base[lane] = my_start;
            __syncthreads();
            // ---- B3: scatter to (bin, lane, rank) order
#pragma unroll
            for (int t = 0; t < ORI_RUN; t++) {
                if (bin[t] < 36u) {
                    const unsigned old = atomicAdd(&s_cnt[bin[t] * ORI_ROW + (lane >> 1)], 1u << half_shift);
                    const int pos = s_base[bin[t]] + (int)((old >> half_shift) & 0xffffu);
                    s_sorted[pos] = val[t];
                }
            }
            __syncthreads();
            // ---- C: ordered sums, ORI_CU LDS reads in flight
            for (int it = 0; __any(it < my_total && lane < 36); it += ORI_CU) {
                float v[ORI_CU];
#pragma unroll
                for (int u = 0; u < ORI_CU; u++) v[u] = (lane < 36 && it + u < my_total) ? s_sorted[my_start + it + u] : 0.f;
#pragma unroll
                for (int u = 0; u < ORI_CU; u++) sum += v[u];
            }
            __syncthreads();
        }
        float mx = (lane < 36) ? sum : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const bool fire = live && (lane < 36) && ((double)sum >= (double)mx * 0.8);
        const unsigned long long m = __ballot(fire);
        if (lane == 0) cmask[ci] = m;
        if (lane < 36) cmm[(size_t)ci * 36 + lane] = sum;
    }
}

// ---- FeatAppend order (Sift.cpp:229-249, :285-293) ------------------------------
// One workgroup turns the per-candidate bin counts into output offsets within the
// octave's feature buffer; a grid then writes the feature records.
__global__ __launch_bounds__(1024) void emit_scan_kernel(const unsigned long long* __restrict__ cmask,
                                                         int* __restrict__ cand_off, int* __restrict__ counters,
                                                         int* __restrict__ gcounters, int octave) {
    __shared__ int s_warp[40];
    const int ncand = counters[CNT_NCAND];
    const int begin = 0;  // every octave fills its own feature buffer
    // thread t owns the candidates [t*C, (t+1)*C): one workgroup scan for any count
    const int C = (ncand + 1023) / 1024;
    const int c0 = threadIdx.x * C, c1 = min(c0 + C, ncand);
    int mine = 0;
    for (int i = c0; i < c1; i++) mine += __popcll(cmask[i]);
    int total;
    int run = begin + block_excl_scan_1024(mine, s_warp, total);
    for (int i = c0; i < c1; i++) {
        cand_off[i] = run;
        run += __popcll(cmask[i]);
    }
    const int running = begin + total;
    if (threadIdx.x == 0) {
        counters[CNT_OCT_BEGIN] = begin;
        counters[CNT_OCT_END] = running;
        gcounters[G_OCT_FEATS + octave] = running;
    }
}

__global__ __launch_bounds__(256) void emit_kernel(const Cand* __restrict__ cand,
                                                   const unsigned long long* __restrict__ cmask,
                                                   const float* __restrict__ cmm, const int* __restrict__ cand_off,
                                                   const LayerConsts* __restrict__ lc, float curr_size,
                                                   sift_hip_feat* __restrict__ feat, int feat_cap,
                                                   const int* __restrict__ counters,
                                                   const float4* __restrict__ cand_sub,
                                                   float* __restrict__ feat_sub, int textbook) {
    const int ncand = counters[CNT_NCAND];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ncand; i += gridDim.x * 256) {
        unsigned long long m = cmask[i];
        if (!m) continue;
        const Cand cd = cand[i];
        const LayerConsts L = lc[cd.layer - 1];
        int pos = cand_off[i];
        while (m) {
            const int bin = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (pos < feat_cap) {
                sift_hip_feat* f = feat + pos;
                f->size = curr_size;
                f->kai = cd.layer;
                f->sigmaOCT = L.sigma_oct;
                f->x = cd.x;
                f->y = cd.y;
                f->mm = cmm[(size_t)i * 36 + bin];
                int sita = bin * 10;  // the reference reports the lower edge of the 10-degree bin (Sift.cpp:287)
                if (textbook) {
                    // quality mode: the bin's centre moved by the vertex of the parabola through the peak and its neighbours
                    const float l = cmm[(size_t)i * 36 + (bin + 35) % 36], c0 = cmm[(size_t)i * 36 + bin],
                                r = cmm[(size_t)i * 36 + (bin + 1) % 36];
                    const float den = l - 2.f * c0 + r;
                    const float off = den < 0.f ? fminf(fmaxf(0.5f * (l - r) / den, -0.5f), 0.5f) : 0.f;
                    sita = ((int)lroundf(10.f * ((float)bin + 0.5f + off)) + 360) % 360;
                }
                f->sita = sita;
                if (feat_sub) {
                    const float4 o = cand_sub[i];
                    feat_sub[(size_t)pos * 3 + 0] = o.x;
                    feat_sub[(size_t)pos * 3 + 1] = o.y;
                    feat_sub[(size_t)pos * 3 + 2] = o.z;
                }
            }
            pos++;
        }
    }
}

void launch_detect(sift_hip_ctx* ctx, OctWs& ws, int octave, size_t PX, int W, int H, int intvls,
                   const LayerConsts* d_lc, float thr, int cand_cap, const float* d_top_taps, int top_K) {
    const int WW = (W + 63) / 64;
    const int nrows = intvls * H;
    const int lazy = d_top_taps != nullptr;
    {
        ProfScope ps(ctx, SIFT_HIP_K_DETECT, ws.stream);
        hipMemsetAsync(ws.row_cnt.p, 0, (size_t)nrows * sizeof(int), ws.stream);
        const int nlay = intvls + 2 - lazy;
        const size_t lds = ((size_t)nlay * DET_LH * DET_LW + 1) * sizeof(float);  // + a spare word (see the staging)
        auto kern = nlay == 5 ? detect_tile_kernel<5> : nlay == 4 ? detect_tile_kernel<4> : detect_tile_kernel<0>;
        hipLaunchKernelGGL(kern, dim3(WW, (H + DET_TR - 1) / DET_TR), dim3(256), lds, ws.stream,
                           (const float*)ws.gau_all.p, PX, W, H, intvls, d_lc, thr, WW,
                           (unsigned long long*)ws.masks.p, (int*)ws.row_cnt.p, lazy);
        if (lazy) {
            // tentative top-layer marks -> list -> exact check against the level that was not built
            const int top = intvls - 1;  // 0-based index of the top layer in masks / row_cnt
            unsigned long long* masks_top = (unsigned long long*)ws.masks.p + (size_t)top * H * WW;
            int* row_cnt_top = (int*)ws.row_cnt.p + (size_t)top * H;
            hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, ws.stream, (const int*)row_cnt_top,
                               (int*)ws.row_off2.p, H, cand_cap, (int*)ws.counters2.p,
                               (int*)ctx->gcounters.p + G_OCT_CAND_TOP + octave);
            hipLaunchKernelGGL(expand_rows_kernel, dim3(H), dim3(64), 0, ws.stream, (const unsigned long long*)masks_top,
                               (const int*)row_cnt_top, (const int*)ws.row_off2.p, H, WW, (Cand*)ws.cand2.p, cand_cap);
            const float* g_lo = (const float*)ws.gau_all.p + (size_t)intvls * PX;
            const size_t lds2 = (size_t)(3 * (top_K + 2) + top_K) * sizeof(float);
            const int nthr = std::min(256, (top_K + 2 + 63) / 64 * 64);  // one thread per input row
            hipLaunchKernelGGL(top_check_kernel, dim3(256 * 32), dim3(nthr), lds2, ws.stream, g_lo, g_lo + PX, W, H,
                               d_top_taps, top_K, (const Cand*)ws.cand2.p, (const int*)ws.counters2.p, WW, masks_top,
                               row_cnt_top);
        }
    }
    {
        ProfScope ps(ctx, SIFT_HIP_K_COMPACT, ws.stream);
        hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, ws.stream, (const int*)ws.row_cnt.p,
                           (int*)ws.row_off.p, nrows, cand_cap, (int*)ws.counters.p,
                           (int*)ctx->gcounters.p + G_OCT_CAND + octave);
        hipLaunchKernelGGL(expand_rows_kernel, dim3(nrows), dim3(64), 0, ws.stream,
                           (const unsigned long long*)ws.masks.p, (const int*)ws.row_cnt.p, (const int*)ws.row_off.p,
                           H, WW, (Cand*)ws.cand.p, cand_cap);
    }
}

void launch_refine(sift_hip_ctx* ctx, OctWs& ws, size_t PX, int W, int H, int intvls, const LayerConsts* d_lc,
                   float contr_thr) {
    ProfScope ps(ctx, SIFT_HIP_K_COMPACT, ws.stream);
    hipLaunchKernelGGL(refine_kernel, dim3(256), dim3(256), 0, ws.stream, (const float*)ws.gau_all.p, PX, W, H, intvls,
                       d_lc, contr_thr, (Cand*)ws.cand.p, (float4*)ws.cand_sub.p, (const int*)ws.counters.p);
}

void launch_orient(sift_hip_ctx* ctx, OctWs& ws, size_t PX, int W, const LayerConsts* d_lc, const float* d_table) {
    ProfScope ps(ctx, SIFT_HIP_K_ORIENT, ws.stream);
    // persistent one-wave workgroups, as many as are resident at once (LDS-limited)
    constexpr int lds_bytes = (37 * ORI_ROW + 59) * 4 + ORI_CH * 4 + 36 * 4;
    // (and register-limited: the kernel's 120-odd registers leave four waves per SIMD)
    constexpr int per_cu = (160 * 1024 / lds_bytes) < 16 ? (160 * 1024 / lds_bytes) : 16;
    hipLaunchKernelGGL(orient_kernel, dim3(256 * per_cu), dim3(64), 0, ws.stream, (const float*)ws.gau_all.p, PX, W, d_lc,
                       d_table, (const Cand*)ws.cand.p, (const int*)ws.counters.p, (unsigned long long*)ws.cand_mask.p,
                       (float*)ws.cand_mm.p);
}

void launch_emit(sift_hip_ctx* ctx, OctWs& ws, int octave, float curr_size, const LayerConsts* d_lc, int feat_cap,
                 bool with_sub, bool textbook) {
    ProfScope ps(ctx, SIFT_HIP_K_EMIT, ws.stream);
    hipLaunchKernelGGL(emit_scan_kernel, dim3(1), dim3(1024), 0, ws.stream,
                       (const unsigned long long*)ws.cand_mask.p, (int*)ws.cand_off.p, (int*)ws.counters.p,
                       (int*)ctx->gcounters.p, octave);
    hipLaunchKernelGGL(emit_kernel, dim3(128), dim3(256), 0, ws.stream, (const Cand*)ws.cand.p,
                       (const unsigned long long*)ws.cand_mask.p, (const float*)ws.cand_mm.p,
                       (const int*)ws.cand_off.p, d_lc, curr_size, (sift_hip_feat*)ws.feat.p, feat_cap,
                       (const int*)ws.counters.p, (const float4*)ws.cand_sub.p,
                       with_sub ? (float*)ws.feat_sub.p : nullptr, textbook ? 1 : 0);
}

}  // namespace sifthip
