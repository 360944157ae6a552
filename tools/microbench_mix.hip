// Issue rate of the exact blur's instruction kinds on one SIMD: independent instructions (no chain),
// 1 to 8 waves per SIMD.  Prints SIMD cycles per wave-level instruction at the measured clock
// (4.0 = one 64-lane instruction every four cycles, the full rate of a 16-lane SIMD).
//   hipcc -O3 --offload-arch=gfx950 -o tools/microbench_mix tools/microbench_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));

// MODE 0: v_fma_f32   1: v_add_f64   2: v_cvt_f64_f32   3: v_pk_mul_f32
//      4: the streaming column pass's mix per 8 outputs: 1 v_pk_mul_f32, 2 v_cvt_f64_f32... (10 pk_mul + 20 cvt + 39 add)
//      5: the row pass's mix (4 v_pk_mul_f32 + 8 v_cvt_f64_f32 + 8 v_add_f64 per tap of 8 outputs)
template <int MODE>
__global__ __launch_bounds__(64) void k(const float* in, double* out, int n) {
    double acc[40];
    float2v f[16];
    double q[20];
#pragma unroll
    for (int j = 0; j < 40; j++) acc[j] = in[threadIdx.x + j];
#pragma unroll
    for (int j = 0; j < 16; j++) f[j] = float2v{in[threadIdx.x + j], in[threadIdx.x + 64 + j]};
#pragma unroll
    for (int j = 0; j < 20; j++) q[j] = in[threadIdx.x + 2 * j];
    const float2v g = {in[200], in[201]};
    for (int i = 0; i < n; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 32; j++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[j & 15].x) : "v"(g.x), "v"(g.y));
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 32; j++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(q[j % 20]));
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 32; j++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(q[j % 20]) : "v"(f[j & 15].x));
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 32; j++) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(f[j & 15]) : "v"(g), "v"(f[(j + 1) & 15]));
        } else if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 20; j++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(q[j]) : "v"((j & 1) ? f[j >> 1].y : f[j >> 1].x));
#pragma unroll
            for (int j = 0; j < 10; j++) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(f[j]) : "v"(g), "v"(f[15]));
#pragma unroll
            for (int j = 0; j < 39; j++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(q[j < 20 ? j : 38 - j]));
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) {
#pragma unroll
                for (int j = 0; j < 4; j++) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(f[j]) : "v"(g), "v"(f[8 + j]));
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(q[j]) : "v"((j & 1) ? f[j >> 1].y : f[j >> 1].x));
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(q[j]));
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 40; j++) s += acc[j];
#pragma unroll
    for (int j = 0; j < 16; j++) s += f[j].x + f[j].y;
#pragma unroll
    for (int j = 0; j < 20; j++) s += q[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
static float run(const float* in, double* out, int blocks, int n) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, in, out, n);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, in, out, n);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* in;
    double* out;
    hipMalloc(&in, 4096);
    hipMalloc(&out, 8 << 20);
    hipMemset(in, 0, 4096);
    const int n = 20000;
    const char* names[6] = {"v_fma_f32", "v_add_f64", "v_cvt_f64_f32", "v_pk_mul_f32", "column-stream mix (10 pk_mul + 20 cvt + 39 add)",
                            "row-pass mix (4 pk_mul + 8 cvt + 8 add)"};
    const int per_iter[6] = {32, 32, 32, 32, 69, 80};
    int clock_khz = 0;
    hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0);
    printf("device clock attribute %d kHz (cycles below use 2.3 GHz, the sustained clock under vector load)\n", clock_khz);
    for (int w : {1, 2, 3, 4, 8}) {
        for (int mode = 0; mode < 6; mode++) {
            float ms = 0;
            const int blocks = 1024 * w;
            switch (mode) {
                case 0: ms = run<0>(in, out, blocks, n); break;
                case 1: ms = run<1>(in, out, blocks, n); break;
                case 2: ms = run<2>(in, out, blocks, n); break;
                case 3: ms = run<3>(in, out, blocks, n); break;
                case 4: ms = run<4>(in, out, blocks, n); break;
                default: ms = run<5>(in, out, blocks, n); break;
            }
            const double ns_per_instr = ms * 1e6 / ((double)n * per_iter[mode] * w);
            printf("waves/SIMD %d  %-50s %.2f ns per instruction = %.2f cycles\n", w, names[mode], ns_per_instr, ns_per_instr * 2.3);
        }
    }
    return 0;
}
