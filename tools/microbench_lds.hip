// LDS atomic throughput: ds_add_f32 vs ds_add_u32 vs plain read-add-write.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int salt) {
    __shared__ float h[64 * 129];
    for (int i = threadIdx.x; i < 64 * 129; i += 256) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + salt;
    float* mine = h + lane * 129;
    for (int it = 0; it < ITERS; it++) {
        s = s * 1664525u + 1013904223u;
        const int bin = (s >> 20) & 127;
        const float v = (float)(s & 255) * 0.001f;
        if (MODE == 0) atomicAdd(&mine[bin], v);
        if (MODE == 1) atomicAdd((unsigned*)&mine[bin], (unsigned)(s & 255));
        if (MODE == 2) mine[bin] += v;
        if (MODE == 3) asm volatile("" ::"v"(bin), "v"(v));
    }
    __syncthreads();
    float acc = 0;
    for (int i = threadIdx.x; i < 64 * 129; i += 256) acc += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 256 * 4 * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<1024, 256>>>(out, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; r++) k<MODE><<<1024, 256>>>(out, r);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    // per CU: 4 blocks x 4 waves x ITERS wave-instructions
    double per_cu = 4.0 * 4 * ITERS;
    printf("%-24s %8.3f ms   %.1f ns per wave-instruction per CU (%.0f cycles @2.1GHz)\n", name, ms, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.1);
    hipFree(out);
}
int main() {
    run<3>("no LDS op (loop only)");
    run<0>("ds_add_f32");
    run<1>("ds_add_u32");
    run<2>("read-add-write f32");
    return 0;
}
