// Times the streaming column pass of the blur on its own (events around 5 launches),
// for several segment lengths.  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -pragma-unroll-threshold=200000 \
//         tools/microbench_stream.hip -o tools/microbench_stream
#include "../sift_amd/csrc/pyramid.hip"

#include <cstdio>
#include <vector>

namespace sifthip {  // the launchers refer to these; unused here
bool hip_ok(sift_hip_ctx*, hipError_t e, const char*) { return e == hipSuccess; }
void prof_begin(sift_hip_ctx*, int, hipStream_t) {}
void prof_end(sift_hip_ctx*, hipStream_t) {}
}  // namespace sifthip

template <int K>
void run(int W, int H, const float* in, float* out, const float* taps) {
    for (int seg : {64, 128, 173, 256, 540, 1080}) {
        dim3 grid((W + 63) / 64, (H + seg - 1) / seg);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        sifthip::blur_col_stream_kernel<K><<<grid, 64>>>(in, out, W, H, taps, seg);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) sifthip::blur_col_stream_kernel<K><<<grid, 64>>>(in, out, W, H, taps, seg);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("K=%d seg %4d (%5d waves): %.1f us\n", K, seg, grid.x * grid.y, ms / 5 * 1e3);
    }
}

int main() {
    const int W = 7680, H = 4320;
    float *in, *out, *taps;
    hipMalloc(&in, (size_t)W * H * 4);
    hipMalloc(&out, (size_t)W * H * 4);
    hipMalloc(&taps, 64 * 4);
    std::vector<float> h((size_t)W * H);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> g(64, 0.02f);
    hipMemcpy(taps, g.data(), 64 * 4, hipMemcpyHostToDevice);
    run<11>(W, H, in, out, taps);
    run<21>(W, H, in, out, taps);
    run<39>(W, H, in, out, taps);
    return 0;
}
