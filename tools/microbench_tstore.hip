// Transposed stores: what run length per row makes them as cheap as coalesced ones?  A wave
// copies a 64-column x ROWS-row block of a W x H float image into its transpose (outT[x * H + y]):
// rows are read coalesced, staged in LDS, and leave as runs of RUN bytes per column
// (RUN / 16 lanes x 16 bytes).  Prints the time and the bytes moved per second for each form,
// next to a plain copy.  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench_tstore.hip -o tools/microbench_tstore
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void copy_kernel(const float* __restrict__ in, float* __restrict__ out, int W, int H,
                                                  int rows) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y0 = blockIdx.y * rows;
    for (int r = 0; r < rows; r++) out[(size_t)(y0 + r) * W + x] = in[(size_t)(y0 + r) * W + x];
}

// LPR = lanes per run (RUN = 16 * LPR bytes), ROWS = 4 * LPR rows staged per flush
template <int LPR>
__global__ __launch_bounds__(64) void tstore_kernel(const float* __restrict__ in, float* __restrict__ outT, int W, int H,
                                                    int rows) {
    constexpr int ROWS = 4 * LPR, PITCH = 68;
    __shared__ float stage[ROWS * PITCH];
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 64;
    const int y0 = blockIdx.y * rows;
    const int c = lane / LPR, r4 = (lane % LPR) * 4;
    constexpr int CPI = 64 / LPR;  // columns per store instruction
    for (int yb = y0; yb < y0 + rows; yb += ROWS) {
        float v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) v[r] = in[(size_t)(yb + r) * W + x0 + lane];
#pragma unroll
        for (int r = 0; r < ROWS; r++) stage[r * PITCH + lane] = v[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int i = 0; i < LPR; i++) {
            const float* sp = stage + r4 * PITCH + c + i * CPI;
            float4 o = make_float4(sp[0], sp[PITCH], sp[2 * PITCH], sp[3 * PITCH]);
            *reinterpret_cast<float4*>(outT + (size_t)(x0 + c + i * CPI) * H + yb + r4) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// one 4-byte store per lane and row: 64 rows of outT per instruction
__global__ __launch_bounds__(64) void tstore4_kernel(const float* __restrict__ in, float* __restrict__ outT, int W, int H,
                                                     int rows) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y0 = blockIdx.y * rows;
    for (int r = 0; r < rows; r++) outT[(size_t)x * H + y0 + r] = in[(size_t)(y0 + r) * W + x];
}

template <typename F>
static void timeit(const char* name, size_t bytes, F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-44s %7.1f us  %6.2f TB/s (read + write)\n", name, ms / 5 * 1e3, 2.0 * bytes / (ms / 5 * 1e-3) / 1e12);
}

int main() {
    for (int pass = 0; pass < 2; pass++) {
        const int W = pass ? 3840 : 7680, H = pass ? 2160 : 4320, rows = pass ? 64 : 128;  // rows: per wave (multiple of 64)
        float *in, *out;
        hipMalloc(&in, (size_t)W * H * 4);
        hipMalloc(&out, (size_t)W * H * 4);
        hipMemset(in, 0, (size_t)W * H * 4);
        const size_t bytes = (size_t)W * H * 4;
        dim3 grid(W / 64, H / rows);
        printf("%d x %d, %d rows per wave (%d waves)\n", W, H, rows, grid.x * grid.y);
        timeit("plain copy", bytes, [&] { copy_kernel<<<grid, 64>>>(in, out, W, H, rows); });
        timeit("transposed, 4 B per lane (64 rows / instr)", bytes, [&] { tstore4_kernel<<<grid, 64>>>(in, out, W, H, rows); });
        timeit("transposed, runs of 16 B", bytes, [&] { tstore_kernel<1><<<grid, 64>>>(in, out, W, H, rows); });
        timeit("transposed, runs of 32 B", bytes, [&] { tstore_kernel<2><<<grid, 64>>>(in, out, W, H, rows); });
        timeit("transposed, runs of 64 B", bytes, [&] { tstore_kernel<4><<<grid, 64>>>(in, out, W, H, rows); });
        timeit("transposed, runs of 128 B", bytes, [&] { tstore_kernel<8><<<grid, 64>>>(in, out, W, H, rows); });
        timeit("transposed, runs of 256 B", bytes, [&] { tstore_kernel<16><<<grid, 64>>>(in, out, W, H, rows); });
        hipFree(in);
        hipFree(out);
    }
    return 0;
}
