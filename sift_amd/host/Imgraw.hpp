// Float image container with the public surface of the reference's
// Sift/Imgraw.hpp:22-128 that the SIFT path and its consumers use.  BMP file
// I/O (the string constructors and bmp()) belongs to the reference's Raw2Img
// module and is not part of this drop-in.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "imglib.hpp"

namespace sift_amd {
// fn(begin, end) over [0, n) in contiguous chunks on a few threads (SIFT_HOST_THREADS, default 8,
// never more than the machine reports): the per-pixel format conversions of whole 4K RGB images
// that the reference's data model asks for (float <-> 8 bit) are memory-bound loops over
// tens of millions of elements, far longer than the GPU work between them when run on one core.
// `grain` = the fewest elements worth a thread of their own.
template <typename F>
inline void parallel_for(size_t n, F fn, size_t grain = (size_t)1 << 18) {
    static const unsigned want = [] {
        const char* e = std::getenv("SIFT_HOST_THREADS");
        const unsigned t = e ? (unsigned)std::atoi(e) : 8u;
        const unsigned hw = std::thread::hardware_concurrency();
        return std::max(1u, std::min(t, hw ? hw : 1u));
    }();
    const size_t nt = std::min<size_t>(want, (n + grain - 1) / std::max<size_t>(grain, 1));
    if (nt <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    const size_t per = (n + nt - 1) / nt;
    for (size_t t = 1; t < nt; t++) th.emplace_back([=] { fn(std::min(n, t * per), std::min(n, (t + 1) * per)); });
    fn((size_t)0, std::min(n, per));
    for (auto& x : th) x.join();
}

// Room for n elements in an empty vector, with the kernel asked to back it by 2 MiB pages where it can: a whole-image
// vector (tens to hundreds of MB) is fresh memory every time, and taking its page faults 4 KiB at a time costs more
// than filling it.  A hint only: when the kernel declines, nothing changes.
template <typename T>
inline void reserve_large(std::vector<T>& v, size_t n) {
    v.reserve(n);
    constexpr uintptr_t huge = (uintptr_t)1 << 21;
    const uintptr_t lo = ((uintptr_t)v.data() + huge - 1) & ~(huge - 1);
    const uintptr_t hi = ((uintptr_t)v.data() + v.capacity() * sizeof(T)) & ~(huge - 1);
    if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
}
}  // namespace sift_amd

float fastAtan2f(float dy, float dx);  // Imgraw.cpp:29-52, degrees in [0,360]

#define ImgRawPixMax 255.0

class ImgRaw {
private:
    using types = float;

public:
    ImgRaw() = default;
    // Copies of whole images (Sift's and Stitching's constructors take theirs by value, as the reference's do) go into
    // memory the kernel is asked to back by large pages, in one pass.
    ImgRaw(const ImgRaw& o) : width(o.width), height(o.height), bitCount(o.bitCount), nomal(o.nomal) {
        sift_amd::reserve_large(raw_img, o.raw_img.size());
        raw_img.assign(o.raw_img.begin(), o.raw_img.end());
    }
    ImgRaw& operator=(const ImgRaw& o) {
        if (this != &o) {
            if (raw_img.capacity() < o.raw_img.size()) {
                std::vector<types>().swap(raw_img);
                sift_amd::reserve_large(raw_img, o.raw_img.size());
            }
            raw_img.assign(o.raw_img.begin(), o.raw_img.end());
            width = o.width, height = o.height, bitCount = o.bitCount, nomal = o.nomal;
        }
        return *this;
    }
    ImgRaw(ImgRaw&&) = default;
    ImgRaw& operator=(ImgRaw&&) = default;
    ImgRaw(std::vector<types> img, uint32_t width, uint32_t height, uint16_t bits)
        : raw_img(std::move(img)), width(width), height(height), bitCount(bits) {}
    ImgRaw(const std::vector<unsigned char>& img, uint32_t width, uint32_t height, uint16_t bits)
        : ImgRaw(img.data(), img.size(), width, height, bits) {}
    // the same from n bytes anywhere (e.g. a page-locked transfer buffer)
    ImgRaw(const unsigned char* src, size_t n, uint32_t width, uint32_t height, uint16_t bits)
        : width(width), height(height), bitCount(bits) {
        sift_amd::reserve_large(raw_img, n);
        raw_img.resize(n);
        // (the 256 possible results of the reference's expression, Imgraw.hpp:30-41, tabulated)
        float lut[256];
        for (int v = 0; v < 256; ++v) lut[v] = nomal ? (float)(v / ImgRawPixMax) : (float)v;
        float* dst = raw_img.data();
        sift_amd::parallel_for(n, [=](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) dst[i] = lut[src[i]];
        });
    }
    ImgRaw(uint32_t width, uint32_t height, uint16_t bits)
        : raw_img((size_t)width * height * (bits / 8)), width(width), height(height), bitCount(bits) {}

    operator std::vector<types>&() { return raw_img; }
    operator const std::vector<types>&() const { return raw_img; }
    // 8-bit copy (Imgraw.hpp:59-82): what the stitcher's Raw images are filled from
    operator std::vector<unsigned char>() const {
        std::vector<unsigned char> img;
        sift_amd::reserve_large(img, raw_img.size());
        img.resize(raw_img.size());
        to_bytes(img.data());
        return img;
    }
    // the same conversion into a buffer of size() bytes the caller provides
    void to_bytes(unsigned char* dst) const {
        const float* src = raw_img.data();
        const bool scaled = nomal;
        sift_amd::parallel_for(raw_img.size(), [=](size_t b, size_t e) {
            if (scaled)
                for (size_t i = b; i < e; ++i) dst[i] = (unsigned char)(src[i] * ImgRawPixMax);
            else
                for (size_t i = b; i < e; ++i) dst[i] = (unsigned char)src[i];
        });
    }
    types& operator[](size_t idx) { return raw_img[idx]; }
    const types& operator[](size_t idx) const { return raw_img[idx]; }
    types& at2d(size_t y, size_t x) { return raw_img[y * width + x]; }
    const types& at2d(size_t y, size_t x) const { return raw_img[y * width + x]; }
    const types atBilinear(float y, float x) const;
    size_t size() const { return raw_img.size(); }
    void resize(uint32_t w, uint32_t h, uint16_t bits) {
        raw_img.resize((size_t)w * h * bits / 8);
        width = w;
        height = h;
        bitCount = bits;
    }
    // Imgraw.cpp:257-269; 24-bit -> gray on the GPU, 8-bit -> copy
    ImgRaw ConverGray() const;
    // Imgraw.cpp:228-242
    static void first(ImgRaw& tar, ImgRaw& sou, float z);
    static void gauBlur(ImgRaw& tar, ImgRaw& sou, float p);

public:
    std::vector<types> raw_img;
    uint32_t width = 0;
    uint32_t height = 0;
    uint16_t bitCount = 0;
    bool nomal = 1;
};
inline bool operator==(const ImgRaw& lhs, const ImgRaw& rhs) { return lhs.width == rhs.width && lhs.height == rhs.height; }
inline bool operator!=(const ImgRaw& lhs, const ImgRaw& rhs) { return !(lhs == rhs); }
