// Lifetimes of the workgroups of some blur kernels (the sweeping one-launch form on an octave-0-sized image, the
// passes of the deepest tiled octave):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DWG_TRACE -Iinclude -Isift_amd/csrc \
//         tools/microbench_sweep.hip -o tools/microbench_sweep && tools/microbench_sweep [seg]
// Prints, per workgroup, start and end relative to the first start (ns, from the 100 MHz clock), its length in shader
// clock cycles, XCC, SE, CU.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sift_amd/csrc/pyramid.hip"

namespace sifthip {  // the profiling hooks of the library, not needed here
void prof_begin(sift_hip_ctx*, int, hipStream_t) {}
void prof_end(sift_hip_ctx*, hipStream_t) {}
}  // namespace sifthip

static void report(const char* what, unsigned nwg) {
    std::vector<unsigned long long> tr(6 * 32768);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(sifthip::g_wg_trace), tr.size() * 8);
    nwg = nwg < 32768 ? nwg : 32768;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (unsigned i = 0; i < nwg; i++) {
        t0 = tr[6 * i + 4] < t0 ? tr[6 * i + 4] : t0;
        t1 = tr[6 * i + 5] > t1 ? tr[6 * i + 5] : t1;
    }
    std::vector<double> st, life;
    for (unsigned i = 0; i < nwg; i++) {
        st.push_back((double)(tr[6 * i + 4] - t0) * 0.01);
        life.push_back((double)(tr[6 * i + 5] - tr[6 * i + 4]) * 0.01);
    }
    std::sort(st.begin(), st.end());
    std::sort(life.begin(), life.end());
    printf("%s: %u workgroups, first start to last end %.1f us; starts p10 %.1f p50 %.1f p90 %.1f max %.1f us; "
           "lifetimes min %.1f p50 %.1f p90 %.1f max %.1f us\n", what, nwg, (double)(t1 - t0) * 0.01, st[nwg / 10], st[nwg / 2],
           st[nwg * 9 / 10], st[nwg - 1], life[0], life[nwg / 2], life[nwg * 9 / 10], life[nwg - 1]);
    if (getenv("WG_TRACE_DUMP"))
        for (unsigned i = 0; i < nwg; i++) {
            const unsigned hw = (unsigned)tr[6 * i + 2], xcc = (unsigned)tr[6 * i + 3] & 0xf;
            printf("wg %5u start_us %7.2f end_us %7.2f cycles %7llu xcc %u se %u cu %2u\n", i, (double)(tr[6 * i + 4] - t0) * 0.01,
                   (double)(tr[6 * i + 5] - t0) * 0.01, tr[6 * i + 1] - tr[6 * i], xcc, (hw >> 13) & 7, (hw >> 8) & 15);
        }
}

template <typename F>
static void timed(const char* what, unsigned nwg, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    printf("%s: %.1f us by events\n", what, best * 1e3);
    report(what, nwg);
}

int main(int argc, char** argv) {
    const int seg = argc > 1 ? atoi(argv[1]) : 176;
    {
        const int W = 7680, H = 4320, K = 39;
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
        timed("sweep<39> 7680x4320", grid.x * grid.y,
              [&] { sifthip::blur_sweep_kernel<39, 1><<<grid, 256, G::lds_bytes>>>(in, out, W, H, taps, seg); });
        hipFree(in);
        hipFree(out);
        hipFree(taps);
    }
    {
        // the deepest octave with tiled passes: 960 x 540 under 309 and 79 taps
        const int W = 960, H = 540;
        float *in, *out, *taps;
        hipMalloc(&in, (size_t)W * H * 4);
        hipMalloc(&out, (size_t)W * H * 4);
        hipMalloc(&taps, 512 * 4);
        std::vector<float> h((size_t)W * H), t(512, 1.f / 309);
        for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 977) / 977.f;
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(taps, t.data(), 512 * 4, hipMemcpyHostToDevice);
        for (int K : {309, 79}) {
            char name[96];
            {
                constexpr int P = 4;
                const int n_in = 64 * P + K - 1 + P;
                const size_t lds = (size_t)(n_in + n_in / P + 1) * sizeof(float);
                const dim3 grid((W + 64 * P - 1) / (64 * P), H);
                snprintf(name, sizeof name, "row<float,4> 960x540 K=%d", K);
                timed(name, grid.x * grid.y, [&] { sifthip::blur_row_kernel<float, 4><<<grid, 64, lds>>>(in, out, W, H, taps, K); });
            }
            {
                const size_t lds = (size_t)(sifthip::padx<8>(64 * 8 + K - 1 + 8) + 1 + 3 * 64 * 9) * sizeof(float);
                const dim3 grid((W + 511) / 512, H);
                snprintf(name, sizeof name, "row_split<4> 960x540 K=%d", K);
                timed(name, grid.x * grid.y, [&] { sifthip::blur_row_split_kernel<4><<<grid, 256, lds>>>(in, out, W, H, taps, K); });
                snprintf(name, sizeof name, "row_split<1> 960x540 K=%d", K);
                timed(name, grid.x * grid.y, [&] { sifthip::blur_row_split_kernel<1><<<grid, 64, lds>>>(in, out, W, H, taps, K); });
            }
            {
                const dim3 grid((W + 63) / 64, (H + 7) / 8);
                snprintf(name, sizeof name, "col_split<4> 960x540 K=%d", K);
                timed(name, grid.x * grid.y, [&] { sifthip::blur_col_split_kernel<4><<<grid, 256>>>(in, out, W, H, taps, K); });
                snprintf(name, sizeof name, "col_split<1> 960x540 K=%d", K);
                timed(name, grid.x * grid.y, [&] { sifthip::blur_col_split_kernel<1><<<grid, 64>>>(in, out, W, H, taps, K); });
            }
        }
    }
    return 0;
}
