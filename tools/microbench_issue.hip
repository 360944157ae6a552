// How fast ONE wave (and 2, 3, 4 ... per SIMD) issues the exact blur's instruction mix.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/microbench_issue.hip -o tools/microbench_issue
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITERS = 1024;
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: 16 independent v_add_f64; 1: 16 independent v_cvt_f64_f32 (+ consumed by xor trick);
// 2: the stream-kernel step: 5 pk_mul + 10 cvt + 21 add on 21 accumulators
template <int MODE>
__global__ __launch_bounds__(64) void k(const float* in, float* out, float g0) {
    double acc[21];
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = in[threadIdx.x + 64 * i];
    for (int i = 0; i < 21; i++) acc[i] = a[i % 16];
    float g = g0;
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] += 1.000001;
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) { acc[i] = (double)a[i]; asm volatile("" : "+v"(acc[i])); }
        }
        if (MODE == 2) {
            double q[11];
#pragma unroll
            for (int m = 0; m < 10; m += 2) {
                f2 gp = {g, g + 1.f};
                f2 pr = gp * a[m];
                q[m] = (double)pr.x;
                q[m + 1] = (double)pr.y;
            }
            q[10] = (double)(a[10] * g);
#pragma unroll
            for (int d = -10; d <= 10; d++) acc[d + 10] += q[10 - (d < 0 ? -d : d)];
#pragma unroll
            for (int j = 0; j < 21; j++) asm volatile("" : "+v"(acc[j]));
            g += 1e-9f;
        }
        if (MODE == 3) {  // the same step 21 times in a row with the accumulator ring rotating: ~800 instructions of straight-line code
#pragma unroll
            for (int t = 0; t < 21; t++) {
                double q[11];
#pragma unroll
                for (int m = 0; m < 10; m += 2) {
                    f2 gp = {g, g + 1.f};
                    f2 pr = gp * a[(m + t) % 16];
                    q[m] = (double)pr.x;
                    q[m + 1] = (double)pr.y;
                }
                q[10] = (double)(a[(10 + t) % 16] * g);
#pragma unroll
                for (int d = -10; d <= 10; d++) acc[(t + d + 21) % 21] += q[10 - (d < 0 ? -d : d)];
#pragma unroll
                for (int j = 0; j < 21; j++) asm volatile("" : "+v"(acc[j]));
            }
            g += 1e-9f;
        }
    }
    double s = 0;
    for (int i = 0; i < 21; i++) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = (float)s;
}

template <int MODE>
void run(const char* name, int instrs, int blocks) {
    float *in, *out;
    hipMalloc(&in, 64 * 16 * 4);
    hipMalloc(&out, blocks * 64 * 4);
    hipMemset(in, 0, 64 * 16 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<MODE><<<blocks, 64>>>(in, out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; r++) k<MODE><<<blocks, 64>>>(in, out, 1.0f);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    const double per_simd = (double)blocks / 1024.0 * ITERS * instrs;  // wave-instructions per SIMD
    printf("%-26s waves/SIMD %.1f: %7.3f ms, %.2f ns per wave-instruction per SIMD (%.1f cyc @2.4GHz)\n", name,
           blocks / 1024.0, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(in);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 8; w++) run<0>("16 x add_f64", 16, 1024 * w);
    for (int w = 1; w <= 8; w *= 2) run<1>("16 x cvt_f64_f32", 16, 1024 * w);
    for (int w = 1; w <= 6; w++) run<2>("step: 5.5 pk + 11 cvt + 21 add", 37, 1024 * w);
    for (int w = 1; w <= 4; w++) run<3>("21 steps straight-line", 37 * 21, 1024 * w);
    return 0;
}
