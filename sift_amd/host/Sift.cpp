// Host side of the drop-in: the reference's classes over the C ABI of
// include/sift_hip.h.  Nothing here computes pixels or distances on the CPU.
#include "Sift.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

#include "../../include/sift_hip.h"
#include "Blend.hpp"

namespace sift_amd {

static void check(sift_hip_ctx* ctx, int rc, const char* what) {
    if (rc == SIFT_HIP_OK) return;
    std::string msg = std::string(what) + ": " + sift_hip_strerror(rc);
    if (ctx && sift_hip_last_error(ctx)[0]) msg += std::string(" (") + sift_hip_last_error(ctx) + ")";
    throw std::runtime_error(msg);
}

sift_hip_ctx* default_context() {
    static sift_hip_ctx* ctx = nullptr;
    static std::once_flag once;
    static int rc = SIFT_HIP_OK;
    std::call_once(once, [] {
        const char* e = std::getenv("SIFT_HIP_DEVICE");
        rc = sift_hip_create(&ctx, e ? std::atoi(e) : 0);
    });
    check(nullptr, rc, "sift_hip_create");
    return ctx;
}

static std::recursive_mutex& context_mutex() {
    static std::recursive_mutex m;
    return m;
}

ContextUse::ContextUse() : ctx(nullptr) {
    context_mutex().lock();
    try {
        ctx = default_context();
    } catch (...) {
        context_mutex().unlock();
        throw;
    }
}

ContextUse::~ContextUse() { context_mutex().unlock(); }

Options& options() {
    static Options o;
    return o;
}

// SIFT_HOST_TIMING=1: the steps of pyramid() / Check() print their wall time to stderr
StepTimer::StepTimer(const char* what) : what_(what), t0_(std::chrono::steady_clock::now()) {
    static const bool on = [] { const char* e = std::getenv("SIFT_HOST_TIMING"); return e && e[0] == '1'; }();
    on_ = on;
}
void StepTimer::step(const char* name) {
    if (!on_) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[%s] %-28s %8.3f ms\n", what_, name, std::chrono::duration<double, std::milli>(t - t0_).count());
    t0_ = t;
}

}  // namespace sift_amd

using sift_amd::check;
using sift_amd::default_context;

// ---- imglib mirrors --------------------------------------------------------------
void Gaus::GauBlur(std::vector<float>& img_gau, const std::vector<float>& img_ori, size_t width, size_t height, float p,
                   size_t mat_len) {
    if (&img_gau == &img_ori) throw file_same("## Erroe! in and out is same.");  // imglib.cpp:76-78
    if (img_ori.size() < width * height) throw Size_error("GauBlur: image smaller than width*height");
    img_gau.resize(height * width);
    sift_amd::ContextUse use;
    sift_hip_ctx* ctx = use.ctx;
    check(ctx, sift_hip_gau_blur(ctx, img_ori.data(), img_gau.data(), (int)width, (int)height, p, (int)mat_len,
                                 sift_amd::options().blur_mode), "sift_hip_gau_blur");
}

std::vector<float> Gaus::gau_matrix(float p, size_t mat_len) {
    std::vector<float> g(mat_len ? mat_len + 2 : (size_t)(p * 6 + 8));
    int n = sift_hip_gau_matrix(p, (int)mat_len, g.data(), (int)g.size());
    if (n < 0 && n != SIFT_HIP_ERR_ARG) {
        g.resize((size_t)-n);
        n = sift_hip_gau_matrix(p, (int)mat_len, g.data(), (int)g.size());
    }
    if (n < 0) throw std::runtime_error("sift_hip_gau_matrix: invalid argument");
    g.resize((size_t)n);
    return g;
}

void Scaling::first(std::vector<float>& img, std::vector<float>& img_ori, size_t width, size_t height, float Ratio) {
    const int newH = static_cast<int>(std::floor(height * Ratio));
    const int newW = static_cast<int>(std::floor(width * Ratio));
    if (newH < 1 || newW < 1) throw Size_error("Scaling::first: empty result");
    img.resize((size_t)newH * newW);
    sift_amd::ContextUse use;
    sift_hip_ctx* ctx = use.ctx;
    check(ctx, sift_hip_first(ctx, img_ori.data(), (int)width, (int)height, Ratio, img.data()), "sift_hip_first");
}

float Scaling::bilinear(const std::vector<float>& img, size_t width, float y, float x) {
    const size_t x0 = (size_t)std::floor(x), x1 = (size_t)std::ceil(x);
    const size_t y0 = (size_t)std::floor(y), y1 = (size_t)std::ceil(y);
    const float dx1 = x - x0, dx2 = 1 - dx1;
    const float dy1 = y - y0, dy2 = 1 - dy1;
    const float A = img[y0 * width + x0], B = img[y0 * width + x1];
    const float C = img[y1 * width + x0], D = img[y1 * width + x1];
    const float AB = A * dx2 + B * dx1;
    const float CD = C * dx2 + D * dx1;
    return AB * dy2 + CD * dy1;
}

// ---- ImgRaw ------------------------------------------------------------------------
float fastAtan2f(float dy, float dx) {
    static const float k = (float)(180 / 3.14159265358979323846);
    static const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    static const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    static const float eps = (float)2.2204460492503131e-016;
    const float ax = std::fabs(dx), ay = std::fabs(dy);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (dx < 0) a = 180.f - a;
    if (dy < 0) a = 360.f - a;
    return a;
}

const float ImgRaw::atBilinear(float y, float x) const {
    if (x < 0 || y < 0 || x >= width - 1 || y >= height - 1) throw Out_of_imgRange("atBilinear(x, y) out of range");
    return Scaling::bilinear(raw_img, width, y, x);
}

ImgRaw ImgRaw::ConverGray() const {
    if (bitCount == 24) {
        ImgRaw gray(width, height, 8);
        sift_amd::ContextUse use;
    sift_hip_ctx* ctx = use.ctx;
        check(ctx, sift_hip_convert_gray(ctx, raw_img.data(), (int)(width * height), gray.raw_img.data()),
              "sift_hip_convert_gray");
        return gray;
    }
    if (bitCount == 8) return *this;
    throw Size_error("ConverGray: bitCount must be 8 or 24");
}

void ImgRaw::first(ImgRaw& tar, ImgRaw& sou, float z) {
    Scaling::first(tar.raw_img, sou.raw_img, sou.width, sou.height, z);
    tar.width = (uint32_t)(size_t)(sou.width * z);
    tar.height = (uint32_t)(size_t)(sou.height * z);
    tar.bitCount = 8;
}

void ImgRaw::gauBlur(ImgRaw& tar, ImgRaw& sou, float p) {
    Gaus::GauBlur(tar.raw_img, sou.raw_img, sou.width, sou.height, p);
    tar.width = sou.width;
    tar.height = sou.height;
    tar.bitCount = sou.bitCount;
}

// ---- Sift ----------------------------------------------------------------------------
Sift::Sift(ImgRaw img, size_t intvls) : raw_img(std::move(img)), pyWidth(intvls + SIFT_INTVLS_DIFF) {
    FeatStart = new Feature[1];  // the reference starts with one empty head node (Sift.cpp:52)
    FeatEnd = FeatStart;
}

Sift::~Sift() {
    if (dev_descr) {
        sift_amd::ContextUse use;
        sift_hip_release_descriptors(use.ctx, (sift_hip_descr_set*)dev_descr);
    }
    delete[] FeatStart;
}

void Sift::pyramid() {
    if (raw_img.bitCount != 8 && raw_img.bitCount != 24) throw Size_error("Sift::pyramid: bitCount must be 8 or 24");
    if (raw_img.raw_img.size() < (size_t)raw_img.width * raw_img.height * (raw_img.bitCount / 8))
        throw Size_error("Sift::pyramid: image buffer smaller than width*height");
    sift_amd::StepTimer tm("pyramid");
    sift_amd::ContextUse use;
    sift_hip_ctx* ctx = use.ctx;
    sift_hip_params prm;
    sift_hip_default_params(&prm);
    prm.intvls = (int)pyWidth - SIFT_INTVLS_DIFF;
    prm.octaves = sift_amd::options().octaves;
    prm.subpixel = sift_amd::options().subpixel;
    prm.blur_mode = sift_amd::options().blur_mode;
    prm.textbook = sift_amd::options().textbook;
    // The records come back into a page-locked buffer kept beside the shared context (direct
    // DMA); like the context it is only touched while `use` holds the context lock.
    static sift_hip_feat* buf = nullptr;
    static int buf_cap = 0;
    int cap = buf_cap > 0 ? buf_cap : 1 << 15, n = 0;
    for (;;) {
        if (cap > buf_cap) {
            sift_hip_host_free(ctx, buf);
            buf = nullptr;
            buf_cap = 0;
            void* p = nullptr;
            check(ctx, sift_hip_host_alloc(ctx, &p, (size_t)cap * sizeof(sift_hip_feat)), "sift_hip_host_alloc");
            buf = (sift_hip_feat*)p;
            buf_cap = cap;
        }
        const int rc = sift_hip_extract(ctx, raw_img.raw_img.data(), (int)raw_img.width, (int)raw_img.height,
                                        raw_img.bitCount / 8, &prm, buf, buf_cap, &n, nullptr);
        if (rc == SIFT_HIP_ERR_NOSPC) {
            cap = n;
            continue;
        }
        check(ctx, rc, "sift_hip_extract");
        break;
    }
    // the descriptor bytes stay on the device for Stitching::Match (the records are rewritten by the next extract)
    if (dev_descr) {
        sift_hip_release_descriptors(ctx, (sift_hip_descr_set*)dev_descr);
        dev_descr = nullptr;
        dev_descr_of = nullptr;
        dev_descr_n = 0;
    }
    sift_hip_descr_set* kept = nullptr;
    if (sift_amd::options().keep_descriptors && n > 0) {
        int kn = 0;
        check(ctx, sift_hip_keep_descriptors(ctx, &kept, &kn), "sift_hip_keep_descriptors");
        if (kn != n) {  // cannot happen after a successful extract; never match stale data
            sift_hip_release_descriptors(ctx, kept);
            kept = nullptr;
        }
    }
    tm.step("sift_hip_extract");
    // refined positions (matrix.hpp:188-189); without refinement img_pt stays zero as in the reference
    std::vector<float> sub;
    if (prm.subpixel && n > 0) {
        sub.resize((size_t)n * 3);
        const int got = sift_hip_last_offsets(ctx, sub.data(), n);
        if (got != n) check(ctx, got < 0 ? got : SIFT_HIP_ERR_HIP, "sift_hip_last_offsets");
    }
    // the reference copies its linked list into new Feature[feaNum] (Sift.cpp:207-224)
    Feature* arr = new Feature[n > 0 ? n : 1];
    const float* subp = sub.empty() ? nullptr : sub.data();
    const sift_hip_feat* recs = buf;
    // (each record is 128 byte-to-float conversions: a thread per 2048 features)
    sift_amd::parallel_for((size_t)n, [=](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            const sift_hip_feat& s = recs[i];
            Feature& f = arr[i];
            f.size = s.size;
            f.kai = s.kai;
            f.sigmaOCT = s.sigmaOCT;
            f.x = s.x;
            f.y = s.y;
            f.mm = s.mm;
            f.sita = s.sita;
            for (int k = 0; k < 128; k++) f.descr[k] = (float)s.descr[k];
            f.d = 128;
            if (subp) {
                f.img_pt.x = ((float)s.x + subp[i * 3 + 0]) / s.size;
                f.img_pt.y = ((float)s.y + subp[i * 3 + 1]) / s.size;
            }
        }
    }, 2048);
    delete[] FeatStart;
    FeatStart = arr;
    FeatEnd = arr + (n > 0 ? n - 1 : 0);
    feaNum = (size_t)n;
    if (kept) {
        dev_descr = kept;
        dev_descr_of = arr;
        dev_descr_n = (size_t)n;
    }
    tm.step("Feature[] fill");
}

// ---- Stitching -------------------------------------------------------------------------
Stitching::Stitching(const Sift& desc1, const Sift& desc2)
    : feat1(desc1.FeatStart), feat2(desc2.FeatStart), sift1(&desc1), sift2(&desc2), feat1_Count(desc1.feaNum),
      feat2_Count(desc2.feaNum) {
    img1 = desc1.raw_img;
    img2 = desc2.raw_img;
    Width = img1.width + img2.width;
    Height = img1.height;
    // side-by-side canvas the reference keeps for its debug drawings (Sift.cpp:502-517)
    if (img1.bitCount == 24 && img2.bitCount == 24 && img1.height == img2.height) {
        stackImg.resize(img1.width + img2.width, img1.height, 24);
        for (size_t j = 0; j < img1.height; j++) {
            std::memcpy(&stackImg[(j * stackImg.width) * 3], &img1[(j * img1.width) * 3], (size_t)img1.width * 3 * 4);
            std::memcpy(&stackImg[(j * stackImg.width + img1.width) * 3], &img2[(j * img2.width) * 3],
                        (size_t)img2.width * 3 * 4);
        }
    }
}

size_t Stitching::Match(float matchTh) {
    // Both feature arrays as pyramid() made them, their descriptor bytes still on the device: match those.
    auto kept_of = [](const Sift* sf, const Feature* f, size_t n) -> sift_hip_descr_set* {
        const bool ok = sift_amd::options().keep_descriptors && sf->dev_descr && sf->FeatStart == f && sf->feaNum == n &&
                        sf->dev_descr_of == f && sf->dev_descr_n == n;
        return ok ? (sift_hip_descr_set*)sf->dev_descr : nullptr;
    };
    sift_hip_descr_set* kq = kept_of(sift2, feat2, feat2_Count);
    sift_hip_descr_set* kr = kept_of(sift1, feat1, feat1_Count);
    if (kq && kr) {
        std::vector<int32_t> best(feat2_Count);
        sift_amd::ContextUse use;
        check(use.ctx, sift_hip_match_kept(use.ctx, kq, kr, matchTh, sift_amd::options().ratio_form, best.data(), nullptr, nullptr),
              "sift_hip_match_kept");
        return link_matches(best);
    }
    std::vector<uint8_t> q(feat2_Count * 128), r(feat1_Count * 128);
    auto pack = [](const Feature* f, size_t n, std::vector<uint8_t>& out) {
        uint8_t* dst = out.data();
        sift_amd::parallel_for(n, [=](size_t b, size_t e) {
            for (size_t i = b; i < e; i++)
                for (int k = 0; k < 128; k++) {
                    const float v = f[i].descr[k];  // integer-valued, 0..255 (Sift.cpp:416-420)
                    dst[i * 128 + k] = (uint8_t)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
                }
        }, 2048);
    };
    pack(feat2, feat2_Count, q);
    pack(feat1, feat1_Count, r);
    std::vector<int32_t> best(feat2_Count);
    sift_amd::ContextUse use;
    sift_hip_ctx* ctx = use.ctx;
    check(ctx, sift_hip_match_u8(ctx, q.data(), (int)feat2_Count, r.data(), (int)feat1_Count, matchTh,
                                 sift_amd::options().ratio_form, best.data(), nullptr, nullptr), "sift_hip_match_u8");
    return link_matches(best);
}

size_t Stitching::link_matches(const std::vector<int32_t>& best) {
    size_t accepted = 0;
    for (size_t i = 0; i < feat2_Count; i++) {
        if (best[i] >= 0) {
            feat2[i].fwd_match = &feat1[best[i]];  // Sift.cpp:577
            accepted++;
        } else if (sift_amd::options().ratio_form == 0) {
            feat2[i].fwd_match = nullptr;
        }
    }
    return accepted;
}

Stitching::~Stitching() { delete[] RANSAC_feat; }

void Stitching::Check(float matchTh) {
    sift_amd::StepTimer tm("Check");
    Match(matchTh);
    tm.step("Match");
    // Sift.cpp:681-684
    delete[] RANSAC_feat;
    RANSAC_feat = nullptr;
    RANSAC_num = 0;
    if (!sift_amd::options().ransac_on_gpu) {
        HomogMat = ransac_xform(feat2, (int)feat2_Count, 4, 0.005f, 3.f, &RANSAC_feat, RANSAC_num);
    } else {
        // the same loop with the sample fits and consensus counts on the GPU; equal results.
        // Pairs as ransac_xform forms them: matched features in index order, positions through rX()/rY().
        std::vector<float> xyuv;
        std::vector<Feature*> owner;
        for (size_t i = 0; i < feat2_Count; i++)
            if (const Feature* mt = feat2[i].fwd_match) {
                xyuv.insert(xyuv.end(), {feat2[i].rX(), feat2[i].rY(), mt->rX(), mt->rY()});
                owner.push_back(feat2 + i);
            }
        std::vector<int32_t> idx(owner.size() + 1);
        float H9[9];
        int32_t n_in = -1;
        sift_amd::ContextUse use;
        sift_hip_ctx* ctx = use.ctx;
        check(ctx, sift_hip_ransac_homog_pairs(ctx, xyuv.data(), (int)owner.size(), 4, 0.005f, 3.f,
                                               sift_amd::ransac_seed(), H9, idx.data(), &n_in, nullptr),
              "sift_hip_ransac_homog_pairs");
        HomogMat.clear();
        if (n_in >= 0) {
            HomogMat.assign(H9, H9 + 9);
            RANSAC_feat = new Feature*[owner.size() + 1];
            for (int32_t i = 0; i < n_in; i++) RANSAC_feat[i] = owner[(size_t)idx[(size_t)i]];
            RANSAC_num = n_in;
        }
    }
    tm.step("RANSAC");
    // blen2img (Sift.cpp:697) for colour pairs with a model; the reference then writes the panorama to
    // a BMP file, which is left to the caller here
    Panorama = ImgRaw();
    if (HomogMat.size() == 9 && RANSAC_num >= 2 && img1.bitCount == 24 && img2.bitCount == 24)
        blen2img(img1, img2, Panorama, HomogMat, RANSAC_feat, RANSAC_num);
    tm.step("blen2img");
}
