// Instruction-throughput microbenchmark for the ops the exact blur is made of.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/microbench_valu.hip -o /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, float g0) {
    float a[8];
    double acc[8];
    for (int i = 0; i < 8; i++) { a[i] = in[threadIdx.x + 256 * i]; acc[i] = a[i]; }
    float g = g0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) { acc[i] += (double)(a[i] * g); }                  // mul + cvt + add_f64
            if (MODE == 1) { acc[i] += 1.000001; }                            // add_f64 only
            if (MODE == 2) { acc[i] = (double)(float)(a[i]); a[i] = a[i] * g; }  // mul + cvt
            if (MODE == 3) { a[i] = a[i] * g; }                               // mul only
            if (MODE == 4) { acc[i] = __builtin_fma(acc[i], 1.0000001, 0.5); }  // fma_f64
            if (MODE == 5) { a[i] = __builtin_fmaf(a[i], g, 0.5f); }          // fma_f32
            if (MODE == 6) { float p = a[i] * g; float t = (float)acc[i] + p; acc[i] = t; } // f32 add chain (in double var)
        }
        if (MODE == 7 || MODE == 8) {
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f2 v = {a[2 * i], a[2 * i + 1]};
                v = v * g;                                                    // one v_pk_mul_f32 for two statements
                if (MODE == 7) { acc[2 * i] += (double)v.x; acc[2 * i + 1] += (double)v.y; }
                if (MODE == 8) { a[2 * i] = v.x; a[2 * i + 1] = v.y; }
            }
        }
        if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 7 || MODE == 8) g += 1e-9f;
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += (float)acc[i] + a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int ops_per_inner) {
    float *in, *out;
    const int blocks = 256 * 8;
    CHECK(hipMalloc(&in, 256 * 8 * 4));
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    std::vector<float> h(256 * 8, 0.5f);
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(in, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k<MODE><<<blocks, 256>>>(in, out, 1.0f);
    hipEventRecord(b);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    // wave-level inner statements executed per SIMD
    double waves = blocks * 4.0;
    double stmts = waves * ITERS * 8.0;
    double per_simd = stmts / 1024.0;
    double ns_per_stmt = ms * 1e6 / per_simd;
    printf("%-28s %8.3f ms  %.2f ns per wave-statement per SIMD (= %.1f cycles @2.4GHz, %.1f @2.0GHz)\n", name, ms, ns_per_stmt, ns_per_stmt * 2.4, ns_per_stmt * 2.0);
    hipFree(in); hipFree(out);
    return 0;
}

int main() {
    run<0>("mul_f32+cvt_f64+add_f64", 3);
    run<1>("add_f64", 1);
    run<2>("mul_f32+cvt_f64_f32", 2);
    run<3>("mul_f32", 1);
    run<4>("fma_f64", 1);
    run<5>("fma_f32", 1);
    run<6>("mul+cvt32<-64+add32+cvt64", 4);
    run<7>("pk_mul_f32/2+cvt_f64+add_f64", 3);
    run<8>("pk_mul_f32/2", 1);
    return 0;
}
