// Lifetimes and placement of the workgroups of blur_sweep_kernel<39, 1> on an octave-0-sized image:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DSWEEP_TRACE -Iinclude -Isift_amd/csrc \
//         tools/microbench_sweep.hip -o tools/microbench_sweep && tools/microbench_sweep [seg]
// Prints, per workgroup, start and end relative to the first start (ns, from the 100 MHz clock), its length in shader
// clock cycles, XCC, SE, CU.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sift_amd/csrc/pyramid.hip"

namespace sifthip {  // the profiling hooks of the library, not needed here
void prof_begin(sift_hip_ctx*, int, hipStream_t) {}
void prof_end(sift_hip_ctx*, hipStream_t) {}
}  // namespace sifthip

int main(int argc, char** argv) {
    const int W = 7680, H = 4320, K = 39;
    const int seg = argc > 1 ? atoi(argv[1]) : 176;
    using G = sifthip::SweepGeom<39, 1>;
    float *in, *out, *taps;
    hipMalloc(&in, (size_t)W * H * 4);
    hipMalloc(&out, (size_t)W * H * 4);
    hipMalloc(&taps, K * 4);
    std::vector<float> h((size_t)W * H), t(K, 1.f / K);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 977) / 977.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(taps, t.data(), K * 4, hipMemcpyHostToDevice);
    const dim3 grid((W + G::TW - 1) / G::TW, (H + seg - 1) / seg);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0);
        sifthip::blur_sweep_kernel<39, 1><<<grid, 256, G::lds_bytes>>>(in, out, W, H, taps, seg);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.1f us, %u workgroups\n", it, ms * 1e3, grid.x * grid.y);
    }
    const unsigned nwg = grid.x * grid.y;
    std::vector<unsigned long long> tr(6 * 8192);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(sifthip::g_sweep_trace), tr.size() * 8);
    unsigned long long t0 = ~0ull;
    for (unsigned i = 0; i < nwg; i++) t0 = tr[6 * i + 4] < t0 ? tr[6 * i + 4] : t0;
    for (unsigned i = 0; i < nwg; i++) {
        const unsigned hw = (unsigned)tr[6 * i + 2], xcc = (unsigned)tr[6 * i + 3] & 0xf;
        printf("wg %4u start_ns %7llu end_ns %7llu cycles %7llu xcc %u se %u cu %2u\n", i, (tr[6 * i + 4] - t0) * 10,
               (tr[6 * i + 5] - t0) * 10, tr[6 * i + 1] - tr[6 * i], xcc, (hw >> 13) & 7, (hw >> 8) & 15);
    }
    return 0;
}
