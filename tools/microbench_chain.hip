// Latency of a dependent chain: one wave, N dependent v_add_f64 (and v_fma_f32 for comparison),
// with and without the product/convert work of the exact blur beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(const float* in, double* out, int n) {
    double acc = 0.0;
    float facc = 0.f;
    float v = in[threadIdx.x];
    const float g = in[64 + (threadIdx.x & 3)];
#pragma unroll 32
    for (int i = 0; i < n; i++) {
        if (MODE == 0) { acc += (double)v; }                                  // add only (operand loop-invariant)
        if (MODE == 1) { const float p = v * g; acc += (double)p; v += 1.f; }  // product + convert + add
        if (MODE == 2) { facc = fmaf(v, g, facc); }                           // float chain
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + facc;
}
int main() {
    float* in; double* out;
    hipMalloc(&in, 1024); hipMalloc(&out, 1 << 20);
    hipMemset(in, 0, 1024);
    const int n = 1 << 16;
    for (int blocks : {1, 1024, 8192}) {
        for (int mode = 0; mode < 3; mode++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            auto run = [&] { if (mode == 0) k<0><<<blocks, 64>>>(in, out, n); else if (mode == 1) k<1><<<blocks, 64>>>(in, out, n); else k<2><<<blocks, 64>>>(in, out, n); };
            run(); hipDeviceSynchronize();
            hipEventRecord(a); run(); hipEventRecord(b); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("waves %5d mode %d (%s): %.1f ns per iteration\n", blocks, mode, mode == 0 ? "v_add_f64 chain" : mode == 1 ? "mul + cvt + add chain" : "v_fma_f32 chain", ms * 1e6 / n);
        }
    }
    return 0;
}
