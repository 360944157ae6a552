// 4x4x8 descriptor for gfx950: one workgroup (64...1024 threads, chosen per octave
// from the window size) per feature, the 128-bin histogram in LDS, the samples of
// the (2R+1)^2 window that can fall into the rotated 4x4 grid spread over its threads.
//
// Arithmetic follows Sift::descr_hist / interp_hist_entry / hist_to_descr
// (Sift.cpp:299-421) expression by expression, including that the orientation in
// DEGREES is used as a radian angle and that both normalize_descr calls are
// no-ops (feat->d is still 0 there), so the descriptor is the clamped raw
// histogram times 512.  atan2/exp/sqrt are hardware-rate approximations (2e-7 rad,
// 1 ulp), divisions by the bin width are multiplications by its reciprocal and
// the bin sums are fixed-point sums rather than the reference's sequential float
// additions, so the histogram matches to ~1e-6 relative, not bit for bit.
#include <algorithm>

#include "internal.h"

namespace sifthip {

// The bins are accumulated in LDS as 64-bit fixed point (2^-30 units) with integer
// atomics: on gfx950 a ds_add_f32 costs ~170 cycles per wave-instruction against
// ~6 for ds_add_u32 (measured, tools/microbench_lds.hip), and the integer sum is
// independent of the order in which lanes and waves arrive, so the result is
// deterministic.  The 128 bins exist NCOPY times (copy = thread mod NCOPY, odd
// row stride) because the samples a wave handles together are neighbours in the
// window and mostly hit the same few bins.
constexpr int NCOPY = 8;
constexpr int HSTRIDE = 129;
constexpr float kFixScale = 1073741824.f;  // 2^30 units per 1.0; one addend is < 4
constexpr int RCH = 512;                   // window rows handled per pass
constexpr int NCHUNK = 2048;               // 64-sample chunks of a pass whose first row is tabulated
constexpr int GU = 1;                      // groups of 4 consecutive samples in flight per thread (2 and 3 measured slower: registers)

// four floats at any 4-byte aligned address (one 16-byte load: global memory takes unaligned accesses)
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));

// v is already scaled by kFixScale
__device__ __forceinline__ unsigned long long to_fixed30(float v) { return (unsigned long long)(unsigned)v; }

// atan2 in (-pi, pi] from a degree-15 odd polynomial on [0,1] (max error 2e-7 rad).
__device__ __forceinline__ float atan2_poly(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
    const float s = a * a;
    float p = -0.004780722316354513f;
    p = fmaf(p, s, 0.024558117613196373f);
    p = fmaf(p, s, -0.059906188398599625f);
    p = fmaf(p, s, 0.0994287058711052f);
    p = fmaf(p, s, -0.1402946412563324f);
    p = fmaf(p, s, 0.1997138410806656f);
    p = fmaf(p, s, -0.3333209455013275f);
    p = fmaf(p, s, 0.9999999403953552f);
    float t = p * a;
    if (ay > ax) t = 1.57079632679489662f - t;
    if (x < 0.f) t = 3.14159265358979324f - t;
    return y < 0.f ? -t : t;
}

// Tightens the inclusive column range [jlo, jhi] of a window row to a superset of
// the columns with |j*p + q| < L.
__device__ __forceinline__ void clip_row(int& jlo, int& jhi, float p, float q, float L) {
    if (fabsf(p) > 1e-3f) {
        const float inv = 1.f / p;
        const float a = (-L - q) * inv, b = (L - q) * inv;
        jlo = max(jlo, (int)floorf(fminf(a, b)));
        jhi = min(jhi, (int)ceilf(fmaxf(a, b)));
    }
}

// cs: cos[36], sin[36] of the orientation (degrees read as radians), then the
// two-float wrap constant of each orientation (see build_plan).
template <bool textbook>
__global__ __launch_bounds__(1024) void descr_kernel(const float* __restrict__ gau, size_t PX, int W, int H,
                                                   const float* __restrict__ cs, sift_hip_feat* __restrict__ feat,
                                                   int feat_cap, const int* __restrict__ counters,
                                                   float* __restrict__ hist_out) {
    __shared__ unsigned long long hist[NCOPY * HSTRIDE];
    __shared__ int s_off[RCH + 1];  // first sample number of each window row of the pass
    __shared__ short s_lo[RCH];     // first column of each row
    // row of the first sample of each 64-sample chunk: a wave's 64 consecutive samples start
    // their search for "my row" there instead of walking every row of the window
    __shared__ unsigned short s_chunk[NCHUNK];
    // blockDim.x (a multiple of 64, chosen per octave from the window size) threads
    // share one feature; windows grow with the octave up to (2*382+1)^2 samples
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63;
    unsigned long long* myhist = hist + (tid & (NCOPY - 1)) * HSTRIDE;
    const int end = min(counters[CNT_OCT_END], feat_cap);
    const float PI2 = (float)(2.f * 3.14159265358979323846);
    const float bins_per_rad = 8 / PI2;
    const float exp_scale = -1.44269504088896341f / (4 * 4 * 0.5f);  // exp(-x/8) = 2^(x * exp_scale)
    // (A work queue on one global counter is no alternative: atomics on a single
    // address retire at ~13 ns each on this part, more than a feature costs.)
    for (int fi = counters[CNT_OCT_BEGIN] + blockIdx.x; fi < end; fi += gridDim.x) {
        const sift_hip_feat* f = feat + fi;
        const int kai = f->kai, r = f->y, c = f->x, sita = f->sita;
        // Quality mode: the frame angle in radians and the un-truncated sigma.  The orientation
        // histogram measures its angles with the row axis pointing down (Sift.cpp:262), the
        // descriptor's gradients with it pointing up (Sift.cpp:306): the same direction is -sita here.
        const float ori = textbook ? -(float)sita * (float)(3.14159265358979323846 / 180.0) : (float)sita;
        const float scl = textbook ? f->sigmaOCT : (float)(int)f->sigmaOCT;
        const float* img = gau + (size_t)kai * PX;
        for (int t = tid; t < NCOPY * HSTRIDE; t += nthr) hist[t] = 0ull;
        const int ob36 = sita / 10;
        const float cos_t = textbook ? cosf(ori) : cs[ob36];
        const float sin_t = textbook ? sinf(ori) : cs[36 + ob36];
        // The reference brings atan2 - ori (ori up to 350 "radians") into [0, 2*pi)
        // by adding 2*pi one float addition at a time (Sift.cpp:369-372); the sum of
        // those roundings for this orientation is folded into wrap_hi + wrap_lo.
        // (quality mode: atan2 - ori lies in (-pi, 3*pi): the two conditional steps below suffice)
        const float wrap_hi = textbook ? 0.f : cs[72 + ob36], wrap_lo = textbook ? 0.f : cs[108 + ob36];
        const float hist_width = 3.f * scl;
        const float inv_hw = 1.f / hist_width;
        const float L = 2.5f * hist_width;
        const int radius = (int)(hist_width * sqrtf(2.f) * (4 + 1.f) * 0.5f + 0.5f);
        const int n = 2 * radius + 1;
        for (int row0 = 0; row0 < n; row0 += RCH) {
            const int nr = min(RCH, n - row0);
            if (tid < 64) {
                // Row ranges: only columns inside the image interior (calc_grad_mag_ori,
                // Sift.cpp:299-311) and inside the rotated 4x4 grid can contribute.
                int running = 0;
                for (int b = 0; b < nr; b += 64) {
                    const int row = b + lane;
                    int jlo = 0, len = 0;
                    if (row < nr) {
                        const int i = row0 + row - radius;
                        const int rr = r + i;
                        if (rr > 0 && rr < H - 1) {
                            int jhi = min(radius, W - 2 - c);
                            jlo = max(-radius, 1 - c);
                            clip_row(jlo, jhi, sin_t, (float)i * cos_t, L);
                            clip_row(jlo, jhi, cos_t, -(float)i * sin_t, L);
                            // (padded to whole groups of 4: a thread takes 4 consecutive samples of ONE row;
                            // the samples past jhi fail the grid test below like any other sample outside it)
                            len = (max(0, jhi - jlo + 1) + 3) & ~3;
                        }
                    }
                    int inc = len;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int t = __shfl_up(inc, o);
                        if (lane >= o) inc += t;
                    }
                    if (row < nr) {
                        s_off[row] = running + inc - len;
                        s_lo[row] = (short)jlo;
                    }
                    running += __shfl(inc, 63);
                }
                if (lane == 0) s_off[nr] = running;
                // (same wave: the LDS writes above are in order before these reads)
                for (int b = 0; b < nr; b += 64) {
                    const int row = b + lane;
                    if (row < nr) {
                        const int lo = s_off[row], hi = s_off[row + 1];
                        for (int ch = (lo + 63) >> 6; ch <= ((hi - 1) >> 6) && ch < NCHUNK; ch++)
                            if (hi > lo) s_chunk[ch] = (unsigned short)row;
                    }
                }
            }
            __syncthreads();
            const int total = s_off[nr];
            int row = 0;
            // A thread takes groups of 4 consecutive samples of one window row: one row search and one address per
            // group, and the 14 pixels its 4 gradients need arrive in four loads (16 + 8 bytes of the row itself, 16 of
            // the row above, 16 of the row below) that are contiguous across the lanes of a wave.  GU groups per
            // thread and step are in flight before any is consumed.  Which thread adds a sample does not matter:
            // the bins are integer sums.
            const int ngroups = total >> 2;
            for (int g0 = tid; g0 < ngroups; g0 += GU * nthr) {
                float4u xa[GU], up[GU], dn[GU];
                float2u xb[GU];
                float rb_[GU][4], cb_[GU][4], ww[GU][4];
                bool ok[GU][4], any[GU];
#pragma unroll
                for (int u = 0; u < GU; u++) {
                    const int g = g0 + u * nthr;
                    any[u] = false;
#pragma unroll
                    for (int e = 0; e < 4; e++) ok[u][e] = false;
                    if (g < ngroups) {
                        const int idx = g << 2;
                        if ((idx >> 6) < NCHUNK) row = s_chunk[idx >> 6];
                        while (idx >= s_off[row + 1]) row++;
                        const int i = row0 + row - radius;
                        const int j0 = s_lo[row] + idx - s_off[row];
                        const float fi_ = (float)i;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float fj = (float)(j0 + e);
                            const float c_rot = (fj * cos_t - fi_ * sin_t) * inv_hw;
                            const float r_rot = (fj * sin_t + fi_ * cos_t) * inv_hw;
                            rb_[u][e] = r_rot + 4 / 2.f - 0.5f;
                            cb_[u][e] = c_rot + 4 / 2.f - 0.5f;
                            ww[u][e] = c_rot * c_rot + r_rot * r_rot;
                            // (j0 + e <= jhi: the padding of a row lies outside the image interior at the right edge)
                            ok[u][e] = rb_[u][e] > -1.f && rb_[u][e] < 4.f && cb_[u][e] > -1.f && cb_[u][e] < 4.f &&
                                       c + j0 + e <= W - 2;
                            any[u] = any[u] || ok[u][e];
                        }
                        if (any[u]) {
                            const float* p = img + (unsigned)((r + i) * W + (c + j0));  // one level holds < 2^31 floats
                            xa[u] = *reinterpret_cast<const float4u*>(p - 1);
                            xb[u] = *reinterpret_cast<const float2u*>(p + 3);
                            up[u] = *reinterpret_cast<const float4u*>(p - W);
                            dn[u] = *reinterpret_cast<const float4u*>(p + W);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < GU; u++) {
                    if (!any[u]) continue;
                    const float xs[6] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w, xb[u].x, xb[u].y};  // columns j0-1 .. j0+4
                    const float us[4] = {up[u].x, up[u].y, up[u].z, up[u].w};
                    const float ds[4] = {dn[u].x, dn[u].y, dn[u].z, dn[u].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (!ok[u][e]) continue;
                        const float rbin = rb_[u][e], cbin = cb_[u][e];
                        const float dx = xs[e + 2] - xs[e];
                        const float dy = us[e] - ds[e];
                        const float mag = __builtin_amdgcn_sqrtf(dx * dx + dy * dy);
                        float go = atan2_poly(dy, dx) - ori;
                        go = (go + wrap_hi) + wrap_lo;
                        if (go < 0.f) go += PI2;
                        if (go >= PI2) go -= PI2;
                        const float obin = go * bins_per_rad;
                        const float w = __builtin_amdgcn_exp2f(ww[u][e] * exp_scale);
                        const float m = mag * w * kFixScale;  // exact scaling by 2^30
                        // interp_hist_entry (Sift.cpp:313-343)
                        const float fr0 = floorf(rbin), fc0 = floorf(cbin), fo0 = floorf(obin);
                        const int r0 = (int)fr0, c0 = (int)fc0, o0 = (int)fo0;
                        const float d_r = rbin - fr0, d_c = cbin - fc0, d_o = obin - fo0;
                        unsigned long long* hb = myhist + (r0 * 4 + c0) * 8;
                        const int ob0 = o0 & 7, ob1 = (o0 + 1) & 7;
#pragma unroll
                        for (int a = 0; a <= 1; a++) {
                            const int rb = r0 + a;
                            if (rb >= 0 && rb < 4) {
                                const float v_r = m * ((a == 0) ? 1.f - d_r : d_r);
#pragma unroll
                                for (int b = 0; b <= 1; b++) {
                                    const int cb = c0 + b;
                                    if (cb >= 0 && cb < 4) {
                                        const float v_c = v_r * ((b == 0) ? 1.f - d_c : d_c);
                                        atomicAdd(&hb[(a * 4 + b) * 8 + ob0], to_fixed30(v_c * (1.f - d_o)));
                                        atomicAdd(&hb[(a * 4 + b) * 8 + ob1], to_fixed30(v_c * d_o));
                                    }
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        // hist_to_descr: clamp at 0.2 (double compare), x512 in double, truncate, min 255
        if (tid < 64) {
            unsigned char q[2];
            if constexpr (!textbook) {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int b = tid * 2 + t;
                    unsigned long long acc = 0ull;
#pragma unroll
                    for (int k = 0; k < NCOPY; k++) acc += hist[k * HSTRIDE + b];
                    float hv = (float)((double)acc * (1.0 / 1073741824.0));
                    if ((double)hv > 0.2) hv = (float)0.2;
                    if (hist_out) hist_out[(size_t)fi * 128 + b] = hv;
                    const int iv = (int)(512.0 * (double)hv);
                    q[t] = (unsigned char)min(255, iv);
                }
            } else {
                // quality mode: unit length, clamp at 0.2, unit length again (what normalize_descr was
                // meant to do, Sift.cpp:389-414); sums over the wave in a fixed order
                float hv[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    unsigned long long acc = 0ull;
#pragma unroll
                    for (int k = 0; k < NCOPY; k++) acc += hist[k * HSTRIDE + tid * 2 + t];
                    hv[t] = (float)((double)acc * (1.0 / 1073741824.0));
                }
                for (int pass = 0; pass < 2; pass++) {
                    float ss = hv[0] * hv[0] + hv[1] * hv[1];
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
                    const float inv = ss > 0.f ? 1.f / sqrtf(ss) : 0.f;
                    hv[0] *= inv;
                    hv[1] *= inv;
                    if (pass == 0) {
                        hv[0] = fminf(hv[0], 0.2f);
                        hv[1] = fminf(hv[1], 0.2f);
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    if (hist_out) hist_out[(size_t)fi * 128 + tid * 2 + t] = hv[t];
                    q[t] = (unsigned char)min(255, (int)(512.0 * (double)hv[t]));
                }
            }
            unsigned char* d = feat[fi].descr + tid * 2;
            d[0] = q[0];
            d[1] = q[1];
        }
        __syncthreads();
    }
}

void launch_descr(sift_hip_ctx* ctx, OctWs& ws, size_t PX, int W, int H, const float* d_cs, int feat_cap,
                  float* d_hist, int max_window, bool textbook) {
    ProfScope ps(ctx, SIFT_HIP_K_DESCR, ws.stream);
    // threads per feature: about 48 window samples per thread for the octave's largest window
    int nthr = ((max_window / 48) + 63) / 64 * 64;
    nthr = std::max(64, std::min(1024, nthr));
    const int blocks = 256 * std::max(2, 2048 / nthr);
    // (the quality mode is its own instance: its in-kernel sin/cos would cost the parity path registers)
    auto kern = textbook ? descr_kernel<true> : descr_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(nthr), 0, ws.stream, (const float*)ws.gau_all.p, PX, W, H,
                       d_cs, (sift_hip_feat*)ws.feat.p, feat_cap, (const int*)ws.counters.p, d_hist);
}

}  // namespace sifthip
