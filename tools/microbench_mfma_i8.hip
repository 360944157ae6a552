// Practical ceiling of v_mfma_i32_32x32x32_i8 on this chip: register-only MFMA loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int ITERS = 4096;
__global__ __launch_bounds__(256) void k(int* out, int seed) {
    v4i a = {seed + (int)threadIdx.x, seed * 3, seed * 5 + (int)threadIdx.x, 7};
    v4i b = {seed * 11, (int)threadIdx.x * 13, 17, seed};
    if (seed < 0) {  // random bytes in every operand register (the toggle rate of real descriptors)
        unsigned x = (unsigned)(threadIdx.x + 256 * blockIdx.x) * 2654435761u + (unsigned)seed;
        for (int e = 0; e < 4; e++) {
            x = x * 1664525u + 1013904223u; a[e] = (int)(x ^ (x >> 13));
            x = x * 1664525u + 1013904223u; b[e] = (int)(x ^ (x >> 11));
        }
    }
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < ITERS; i++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, b, c3, 0, 0, 0);
    }
    int s = 0;
    for (int e = 0; e < 16; e++) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    int* out;
    hipMalloc(&out, 2048 * 256 * 4);
    for (int waves_per_simd = 1; waves_per_simd <= 2; waves_per_simd++) {
        const int blocks = 256 * waves_per_simd;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        k<<<blocks, 256>>>(out, 1);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            for (int r = 0; r < 20; r++) k<<<blocks, 256>>>(out, r);
            hipEventRecord(b);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
            double ops = (double)blocks * 4 * ITERS * 4 * 65536.0;
            printf("waves/SIMD=%d  %.3f ms  %.1f TOP/s\n", waves_per_simd, ms, ops / ms / 1e9);
        }
    }
    // sustained: ~100 ms of back-to-back launches with random operands (the clock settles under load)
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        for (int r = 0; r < 250; r++) k<<<512, 256>>>(out, -1 - r);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        double ops = 250.0 * 512 * 4 * ITERS * 4 * 65536.0;
        printf("sustained random operands, 2 waves/SIMD: %.1f ms  %.1f TOP/s\n", ms, ops / ms / 1e9);
    }
    return 0;
}
