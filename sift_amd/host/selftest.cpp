// Exercises the drop-in C++ classes the way the reference's StitchImg does
// (Sift_main.cpp:31-55): two images -> Sift::pyramid() each -> Stitching::Check.
//   selftest <in.bin> <out.bin> [matchTh]
// in.bin : int32 w, h, channels; then two images of w*h*channels float32
// out.bin: int32 n1, n2; n1 + n2 records {float size; int kai; float sigmaOCT;
//          int x, y; float mm; int sita; uint8 descr[128]}; then n2 int32 match
//          indices into image 1's features (-1 = none)
// stdout: one JSON line with counts, the RANSAC homography (image 2 -> image 1)
//          and the indices of its consensus features in image 2
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "Blend.hpp"
#include "Sift.hpp"
#include "stitch.hpp"

#pragma pack(push, 1)
struct Rec {
    float size;
    int32_t kai;
    float sigmaOCT;
    int32_t x, y;
    float mm;
    int32_t sita;
    uint8_t descr[128];
};
#pragma pack(pop)

static void dump(FILE* f, const Sift& s) {
    for (size_t i = 0; i < s.feaNum; i++) {
        const Feature& a = s.FeatStart[i];
        Rec r;
        r.size = a.size; r.kai = a.kai; r.sigmaOCT = a.sigmaOCT; r.x = a.x; r.y = a.y; r.mm = a.mm; r.sita = a.sita;
        for (int k = 0; k < 128; k++) r.descr[k] = (uint8_t)a.descr[k];
        fwrite(&r, sizeof r, 1, f);
    }
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: selftest in.bin out.bin [matchTh]\n"); return 2; }
    const float th = argc > 3 ? (float)atof(argv[3]) : 0.5f;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open input"); return 2; }
    int32_t hdr[3];
    if (fread(hdr, 4, 3, f) != 3) return 2;
    const int w = hdr[0], h = hdr[1], ch = hdr[2];
    const size_t n = (size_t)w * h * ch;
    std::vector<float> a(n), b(n);
    if (fread(a.data(), 4, n, f) != n || fread(b.data(), 4, n, f) != n) return 2;
    fclose(f);
    try {
        ImgRaw img1(a, w, h, (uint16_t)(8 * ch)), img2(b, w, h, (uint16_t)(8 * ch));
        {   // warm-up (context creation, workspace and transfer-buffer growth), so that the times below are steady-state
            Sift warm1(img1), warm2(img2);
            warm1.pyramid();
            warm2.pyramid();
            Stitching warm(warm1, warm2);
            warm.Check(th);
        }
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point t0, std::chrono::steady_clock::time_point t1) {
            return std::chrono::duration<double, std::milli>(t1 - t0).count();
        };
        const auto t0 = now();
        Sift fea1(img1);
        fea1.pyramid();
        Sift fea2(img2);
        fea2.pyramid();
        const auto t1 = now();
        Stitching match(fea1, fea2);
        sift_amd::ransac_seed() = 1234;  // reproducible sample stream for the checker
        const auto t2 = now();
        match.Check(th);
        const auto t3 = now();
        const double ms_pyramids = ms(t0, t1), ms_check = ms(t2, t3);
        // Check() ran RANSAC on the GPU; the CPU implementation must agree bit for bit
        Feature** cpu_in = nullptr;
        int cpu_n = 0;
        const std::vector<float> cpu_H = ransac_xform(fea2.FeatStart, (int)fea2.feaNum, 4, 0.005f, 3.f, &cpu_in, cpu_n);
        bool ransac_equal = cpu_H == match.HomogMat && cpu_n == match.RANSAC_num;
        for (int i = 0; ransac_equal && i < cpu_n; i++) ransac_equal = cpu_in[i] == match.RANSAC_feat[i];
        delete[] cpu_in;
        FILE* o = fopen(argv[2], "wb");
        if (!o) { perror("open output"); return 2; }
        const int32_t cnt[2] = {(int32_t)fea1.feaNum, (int32_t)fea2.feaNum};
        fwrite(cnt, 4, 2, o);
        dump(o, fea1);
        dump(o, fea2);
        size_t matched = 0;
        for (size_t i = 0; i < fea2.feaNum; i++) {
            const Feature* m = fea2.FeatStart[i].fwd_match;
            const int32_t idx = m ? (int32_t)(m - fea1.FeatStart) : -1;
            matched += m != nullptr;
            fwrite(&idx, 4, 1, o);
        }
        fclose(o);
        // the stitcher's warps as blen2img calls them (Blend.cpp:907-922, :974), for RGB input:
        // <out.bin>.warp = int32 cols, rows + bytes of the perspective canvas, then for each of the
        // two cylinder outputs int32 cols, rows + bytes, then int32 n + n (x, y) floats of upedge, downedge
        if (ch == 3 && match.HomogMat.size() == 9) {
            auto to_raw = [](const ImgRaw& src) {
                Raw dst(src.width, src.height);
                dst.RGB = src;
                return dst;
            };
            std::vector<Raw> in = {to_raw(img1), to_raw(img2)};
            Raw warpImg((size_t)(in[1].getCol() * 1.6), (size_t)(in[1].getRow() * 1.1));
            _WarpPerspective(in[1], warpImg, match.HomogMat);
            std::vector<Raw> cyl;
            std::vector<fpoint> up, down;
            warping(in, 700.f, cyl, up, down);
            FILE* wf = fopen((std::string(argv[2]) + ".warp").c_str(), "wb");
            if (!wf) { perror("open warp output"); return 2; }
            auto put = [&](const Raw& r) {
                const int32_t d[2] = {r.getCol(), r.getRow()};
                fwrite(d, 4, 2, wf);
                fwrite(r.RGB.data(), 1, r.RGB.size(), wf);
            };
            put(warpImg);
            put(cyl[0]);
            put(cyl[1]);
            for (const std::vector<fpoint>* e : {&up, &down}) {
                const int32_t n = (int32_t)e->size();
                fwrite(&n, 4, 1, wf);
                for (const fpoint& p : *e) { fwrite(&p.x, 4, 1, wf); fwrite(&p.y, 4, 1, wf); }
            }
            fclose(wf);
        }
        // Check() matched the descriptor bytes pyramid() kept on the device; packing Feature[] and uploading it
        // again must link every feature to the same partner, and so must a second pyramid() of the same object
        bool match_paths_equal = fea1.dev_descr && fea2.dev_descr;
        {
            std::vector<const Feature*> kept(fea2.feaNum);
            for (size_t i = 0; i < fea2.feaNum; i++) kept[i] = fea2.FeatStart[i].fwd_match;
            sift_amd::options().keep_descriptors = 0;
            const size_t acc_packed = match.Match(th);
            sift_amd::options().keep_descriptors = 1;
            match_paths_equal = match_paths_equal && acc_packed == matched;
            for (size_t i = 0; match_paths_equal && i < fea2.feaNum; i++) match_paths_equal = kept[i] == fea2.FeatStart[i].fwd_match;
            // a Feature[] the caller edited: declared by clearing dev_descr_of, matched from the edited floats
            if (fea2.feaNum > 0 && fea1.feaNum > 0) {
                float saved[128];
                std::memcpy(saved, fea2.FeatStart[0].descr, sizeof saved);
                std::memcpy(fea2.FeatStart[0].descr, fea1.FeatStart[fea1.feaNum - 1].descr, sizeof saved);
                const Feature* of = fea2.dev_descr_of;
                fea2.dev_descr_of = nullptr;
                match.Match(th);
                std::vector<const Feature*> declared(fea2.feaNum);
                for (size_t i = 0; i < fea2.feaNum; i++) declared[i] = fea2.FeatStart[i].fwd_match;
                fea2.dev_descr_of = of;
                sift_amd::options().keep_descriptors = 0;
                match.Match(th);
                sift_amd::options().keep_descriptors = 1;
                for (size_t i = 0; match_paths_equal && i < fea2.feaNum; i++) match_paths_equal = declared[i] == fea2.FeatStart[i].fwd_match;
                std::memcpy(fea2.FeatStart[0].descr, saved, sizeof saved);
                fea2.dev_descr_of = of;
                match.Match(th);
                for (size_t i = 0; match_paths_equal && i < fea2.feaNum; i++) match_paths_equal = kept[i] == fea2.FeatStart[i].fwd_match;
            }
        }
        // the image-op mirrors: blur into the same vector must throw file_same
        bool threw = false;
        try { Gaus::GauBlur(img1.raw_img, img1.raw_img, w, h, 1.6f); } catch (const file_same&) { threw = true; }
        // optional sub-pixel stage: img_pt carries the refined source-image position
        sift_amd::options().subpixel = 1;
        Sift fea3(img1);
        fea3.pyramid();
        sift_amd::options().subpixel = 0;
        double sx = 0.0, sy = 0.0;
        for (size_t i = 0; i < fea3.feaNum; i++) {
            sx += fea3.FeatStart[i].img_pt.x;
            sy += fea3.FeatStart[i].img_pt.y;
        }
        // two Sift objects driven from two threads at once share the process-wide context under its lock
        bool threads_equal = true;
        {
            Sift ta(img1), tb(img2);
            std::thread t1([&] { for (int i = 0; i < 3; i++) ta.pyramid(); });
            std::thread t2([&] { for (int i = 0; i < 3; i++) tb.pyramid(); });
            t1.join();
            t2.join();
            threads_equal = ta.feaNum == fea1.feaNum && tb.feaNum == fea2.feaNum;
            for (size_t i = 0; threads_equal && i < ta.feaNum; i++)
                threads_equal = ta.FeatStart[i].x == fea1.FeatStart[i].x && ta.FeatStart[i].y == fea1.FeatStart[i].y &&
                                !std::memcmp(ta.FeatStart[i].descr, fea1.FeatStart[i].descr, sizeof ta.FeatStart[i].descr);
            for (size_t i = 0; threads_equal && i < tb.feaNum; i++)
                threads_equal = !std::memcmp(tb.FeatStart[i].descr, fea2.FeatStart[i].descr, sizeof tb.FeatStart[i].descr);
        }
        // Check() went on to blen2img for colour pairs: <out.bin>.pano = int32 cols, rows + the float canvas
        if (match.Panorama.size() > 0) {
            FILE* pf = fopen((std::string(argv[2]) + ".pano").c_str(), "wb");
            if (!pf) { perror("open panorama output"); return 2; }
            const int32_t dim[2] = {(int32_t)match.Panorama.width, (int32_t)match.Panorama.height};
            fwrite(dim, 4, 2, pf);
            fwrite(match.Panorama.raw_img.data(), 4, match.Panorama.raw_img.size(), pf);
            fclose(pf);
        }
        printf("{\"ms_two_pyramids\": %.3f, \"ms_check\": %.3f, ", ms_pyramids, ms_check);
        printf("\"threads_equal\": %s, \"pano\": [%u, %u], \"focal\": %.9g, ", threads_equal ? "true" : "false",
               match.Panorama.width, match.Panorama.height,
               match.HomogMat.size() == 9 ? getFocal(match.HomogMat, img1.size(), img2.size()) : 0.f);
        printf("\"n1\": %zu, \"n2\": %zu, \"matched\": %zu, \"file_same\": %s, \"n_sub\": %zu, \"sub_sum_x\": %.6f, "
               "\"sub_sum_y\": %.6f, \"ransac_equal\": %s, \"match_paths_equal\": %s, \"ransac_num\": %d, \"homog\": [",
               fea1.feaNum, fea2.feaNum, matched, threw ? "true" : "false", fea3.feaNum, sx, sy, ransac_equal ? "true" : "false",
               match_paths_equal ? "true" : "false", match.RANSAC_num);
        for (size_t i = 0; i < match.HomogMat.size(); i++) printf("%s%.9g", i ? ", " : "", match.HomogMat[i]);
        printf("], \"ransac_feat\": [");
        for (int i = 0; i < match.RANSAC_num; i++) printf("%s%d", i ? ", " : "", (int)(match.RANSAC_feat[i] - fea2.FeatStart));
        printf("]}\n");
    } catch (const std::exception& e) {
        fprintf(stderr, "selftest failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
