// 4x4x8 descriptor for gfx950: one wavefront per feature, the 128-bin histogram
// in LDS, samples of the (2R+1)^2 window spread over the 64 lanes.
//
// Arithmetic follows Sift::descr_hist / interp_hist_entry / hist_to_descr
// (Sift.cpp:299-421) expression by expression, including that the orientation in
// DEGREES is used as a radian angle and that both normalize_descr calls are
// no-ops (feat->d is still 0 there), so the descriptor is the clamped raw
// histogram times 512.  atan2f/expf come from the device math library and the
// bin sums are exact fixed-point sums rather than the reference's sequential
// float additions, so the histogram matches to rounding (~1e-6 relative), not
// bit for bit.
#include <algorithm>

#include "internal.h"

namespace sifthip {

// The bins are accumulated in LDS as 64-bit fixed point (2^-40 units) with integer
// atomics: on gfx950 a ds_add_f32 costs ~170 cycles per wave-instruction against
// ~6 for ds_add_u32 (measured, tools/microbench_lds.hip), and the integer sum is
// independent of the order in which lanes and waves arrive, so the result is
// deterministic and within 2^-41 per addend of the exact sum.  The 128 bins
// exist NCOPY times (copy = thread mod NCOPY, odd row stride) because the
// samples a wave handles together are neighbours in the window and mostly hit
// the same few bins.
constexpr int NCOPY = 8;
constexpr int HSTRIDE = 129;
constexpr float kFixHi = 256.f;          // 2^8
constexpr float kFixLo = 4294967296.f;   // 2^32  (2^8 * 2^32 = 2^40 units per 1.0)

__device__ __forceinline__ unsigned long long to_fixed40(float v) {
    const float t = v * kFixHi;              // exact
    const unsigned hi = (unsigned)t;         // v < 2^16 here, so (float)hi below is exact
    const float fr = t - (float)hi;          // exact, in [0,1)
    const unsigned lo = (unsigned)(fr * kFixLo);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(1024) void descr_kernel(const float* __restrict__ gau, size_t PX, int W, int H,
                                                   const float* __restrict__ cs, sift_hip_feat* __restrict__ feat,
                                                   int feat_cap, const int* __restrict__ counters,
                                                   float* __restrict__ hist_out) {
    __shared__ unsigned long long hist[NCOPY * HSTRIDE];
    // blockDim.x (a multiple of 64, chosen per octave from the window size) threads
    // share one feature; windows grow with the octave up to (2*382+1)^2 samples
    const int tid = threadIdx.x, nthr = blockDim.x;
    unsigned long long* myhist = hist + (tid & (NCOPY - 1)) * HSTRIDE;
    const int begin = counters[CNT_OCT_BEGIN];
    const int end = min(counters[CNT_OCT_END], feat_cap);
    const float PI2 = (float)(2.f * 3.14159265358979323846);
    const float bins_per_rad = 8 / PI2;
    const float exp_denom = 4 * 4 * 0.5f;
    for (int fi = begin + blockIdx.x; fi < end; fi += gridDim.x) {
        const sift_hip_feat* f = feat + fi;
        const int kai = f->kai, r = f->y, c = f->x, sita = f->sita;
        const float ori = (float)sita;
        const float scl = (float)(int)f->sigmaOCT;
        const float* img = gau + (size_t)kai * PX;
        for (int t = tid; t < NCOPY * HSTRIDE; t += nthr) hist[t] = 0ull;
        __syncthreads();
        const float cos_t = cs[sita / 10];
        const float sin_t = cs[36 + sita / 10];
        const float hist_width = 3.f * scl;
        const int radius = (int)(hist_width * sqrtf(2.f) * (4 + 1.f) * 0.5f + 0.5f);
        // every sample needs at least `pre` additions of 2*pi to leave the negatives
        int pre = (int)floorf((ori - 3.2f) / 6.2831855f) - 1;
        if (pre < 0) pre = 0;
        // the (2*radius+1)^2 window, flattened row-major over the 64 lanes
        const int n = 2 * radius + 1;
        const int total = n * n;
        const int di = nthr / n, dj = nthr - di * n;
        int wi = tid / n;
        int wj = tid - wi * n;
        for (int idx = tid; idx < total; idx += nthr) {
            const int i = wi - radius, j = wj - radius;
            wj += dj;
            wi += di;
            if (wj >= n) {
                wj -= n;
                wi++;
            }
            const int rr = r + i, cc = c + j;
            if (rr <= 0 || rr >= H - 1 || cc <= 0 || cc >= W - 1) continue;  // calc_grad_mag_ori rejects it
            const float c_rot = ((float)j * cos_t - (float)i * sin_t) / hist_width;
            const float r_rot = ((float)j * sin_t + (float)i * cos_t) / hist_width;
            const float rbin = r_rot + 4 / 2.f - 0.5f;
            const float cbin = c_rot + 4 / 2.f - 0.5f;
            if (!(rbin > -1.f && rbin < 4.f && cbin > -1.f && cbin < 4.f)) continue;
            const size_t p = (size_t)rr * W + cc;
            const float dx = img[p + 1] - img[p - 1];
            const float dy = img[p - W] - img[p + W];
            const float mag = sqrtf(dx * dx + dy * dy);
            float go = atan2f(dy, dx);
            go -= ori;
            for (int t = 0; t < pre; t++) go += PI2;
            while (go < 0.f) go += PI2;
            while (go >= PI2) go -= PI2;
            const float obin = go * bins_per_rad;
            const float w = expf(-(c_rot * c_rot + r_rot * r_rot) / exp_denom);
            const float m = mag * w;
            // interp_hist_entry
            const float fr0 = floorf(rbin), fc0 = floorf(cbin), fo0 = floorf(obin);
            const int r0 = (int)fr0, c0 = (int)fc0, o0 = (int)fo0;
            const float d_r = rbin - fr0, d_c = cbin - fc0, d_o = obin - fo0;
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
#pragma unroll
                            for (int o = 0; o <= 1; o++) {
                                const int ob = (o0 + o) & 7;
                                const float v_o = v_c * ((o == 0) ? 1.f - d_o : d_o);
                                atomicAdd(&myhist[(rb * 4 + cb) * 8 + ob], to_fixed40(v_o));
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // hist_to_descr: clamp at 0.2 (double compare), x512 in double, truncate, min 255
        if (tid < 64) {
        unsigned char q[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int b = tid * 2 + t;
            unsigned long long acc = 0ull;
#pragma unroll
            for (int k = 0; k < NCOPY; k++) acc += hist[k * HSTRIDE + b];
            float hv = (float)((double)acc * (1.0 / 1099511627776.0));
            if ((double)hv > 0.2) hv = (float)0.2;
            if (hist_out) hist_out[(size_t)fi * 128 + b] = hv;
            const int iv = (int)(512.0 * (double)hv);
            q[t] = (unsigned char)min(255, iv);
        }
        unsigned char* d = feat[fi].descr + tid * 2;
        d[0] = q[0];
        d[1] = q[1];
        }
        __syncthreads();
    }
}

void launch_descr(sift_hip_ctx* ctx, OctWs& ws, size_t PX, int W, int H, const float* d_cs, int feat_cap,
                  float* d_hist, int max_window) {
    ProfScope ps(ctx, SIFT_HIP_K_DESCR, ws.stream);
    // threads per feature: about 48 window samples per thread for the octave's largest window
    int nthr = ((max_window / 48) + 63) / 64 * 64;
    nthr = std::max(64, std::min(1024, nthr));
    const int blocks = 256 * std::max(2, 2048 / nthr);
    hipLaunchKernelGGL(descr_kernel, dim3(blocks), dim3(nthr), 0, ws.stream, (const float*)ws.gau_all.p, PX, W, H,
                       d_cs, (sift_hip_feat*)ws.feat.p, feat_cap, (const int*)ws.counters.p, d_hist);
}

}  // namespace sifthip
