// 4x4x8 descriptor for gfx950: one wavefront per feature, the 128-bin histogram
// in LDS, samples of the (2R+1)^2 window spread over the 64 lanes.
//
// Arithmetic follows Sift::descr_hist / interp_hist_entry / hist_to_descr
// (Sift.cpp:299-421) expression by expression, including that the orientation in
// DEGREES is used as a radian angle and that both normalize_descr calls are
// no-ops (feat->d is still 0 there), so the descriptor is the clamped raw
// histogram times 512.  atan2f/expf come from the device math library and the
// bin sums are accumulated in a different order than the sequential reference,
// so the histogram matches to rounding (~1e-6 relative), not bit for bit.
#include "internal.h"

namespace sifthip {

__global__ __launch_bounds__(64) void descr_kernel(const float* __restrict__ gau, size_t PX, int W, int H,
                                                   const float* __restrict__ cs, sift_hip_feat* __restrict__ feat,
                                                   int feat_cap, const int* __restrict__ counters,
                                                   float* __restrict__ hist_out) {
    __shared__ float hist[128];
    const int lane = threadIdx.x;
    const int begin = counters[CNT_OCT_BEGIN];
    const int end = min(counters[CNT_FEAT_TOTAL], feat_cap);
    const float PI2 = (float)(2.f * 3.14159265358979323846);
    const float bins_per_rad = 8 / PI2;
    const float exp_denom = 4 * 4 * 0.5f;
    for (int fi = begin + blockIdx.x; fi < end; fi += gridDim.x) {
        const sift_hip_feat* f = feat + fi;
        const int kai = f->kai, r = f->y, c = f->x, sita = f->sita;
        const float ori = (float)sita;
        const float scl = (float)(int)f->sigmaOCT;
        const float* img = gau + (size_t)kai * PX;
        hist[lane] = 0.f;
        hist[lane + 64] = 0.f;
        __syncthreads();
        const float cos_t = cs[sita / 10];
        const float sin_t = cs[36 + sita / 10];
        const float hist_width = 3.f * scl;
        const int radius = (int)(hist_width * sqrtf(2.f) * (4 + 1.f) * 0.5f + 0.5f);
        // every sample needs at least `pre` additions of 2*pi to leave the negatives
        int pre = (int)floorf((ori - 3.2f) / 6.2831855f) - 1;
        if (pre < 0) pre = 0;
        for (int i = -radius; i <= radius; i++) {
            const int rr = r + i;
            if (rr <= 0 || rr >= H - 1) continue;  // calc_grad_mag_ori rejects the row
            const float isin = (float)i * sin_t, icos = (float)i * cos_t;
            for (int j0 = -radius; j0 <= radius; j0 += 64) {
                const int j = j0 + lane;
                const int cc = c + j;
                if (j > radius || cc <= 0 || cc >= W - 1) continue;
                const float c_rot = ((float)j * cos_t - isin) / hist_width;
                const float r_rot = ((float)j * sin_t + icos) / hist_width;
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
                                    atomicAdd(&hist[(rb * 4 + cb) * 8 + ob], v_o);
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // hist_to_descr: clamp at 0.2 (double compare), x512 in double, truncate, min 255
        unsigned char q[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            float hv = hist[lane * 2 + t];
            if ((double)hv > 0.2) hv = (float)0.2;
            if (hist_out) hist_out[(size_t)fi * 128 + lane * 2 + t] = hv;
            const int iv = (int)(512.0 * (double)hv);
            q[t] = (unsigned char)min(255, iv);
        }
        unsigned char* d = feat[fi].descr + lane * 2;
        d[0] = q[0];
        d[1] = q[1];
        __syncthreads();
    }
}

void launch_descr(sift_hip_ctx* ctx, const float* gau, size_t PX, int W, int H, const float* d_cs, int feat_cap,
                  float* d_hist) {
    ProfScope ps(ctx, SIFT_HIP_K_DESCR);
    hipLaunchKernelGGL(descr_kernel, dim3(256 * 16), dim3(64), 0, ctx->stream, gau, PX, W, H, d_cs,
                       (sift_hip_feat*)ctx->feat.p, feat_cap, (const int*)ctx->counters.p, d_hist);
}

}  // namespace sifthip
