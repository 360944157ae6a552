// Float image container with the public surface of the reference's
// Sift/Imgraw.hpp:22-128 that the SIFT path and its consumers use.  BMP file
// I/O (the string constructors and bmp()) belongs to the reference's Raw2Img
// module and is not part of this drop-in.
#pragma once
#include <cstdint>
#include <vector>

#include "imglib.hpp"

float fastAtan2f(float dy, float dx);  // Imgraw.cpp:29-52, degrees in [0,360]

#define ImgRawPixMax 255.0

class ImgRaw {
private:
    using types = float;

public:
    ImgRaw() = default;
    ImgRaw(std::vector<types> img, uint32_t width, uint32_t height, uint16_t bits)
        : raw_img(std::move(img)), width(width), height(height), bitCount(bits) {}
    ImgRaw(const std::vector<unsigned char>& img, uint32_t width, uint32_t height, uint16_t bits)
        : width(width), height(height), bitCount(bits) {
        raw_img.resize(img.size());
        for (size_t i = 0; i < img.size(); ++i) raw_img[i] = nomal ? (float)(img[i] / ImgRawPixMax) : (float)img[i];
    }
    ImgRaw(uint32_t width, uint32_t height, uint16_t bits)
        : raw_img((size_t)width * height * (bits / 8)), width(width), height(height), bitCount(bits) {}

    operator std::vector<types>&() { return raw_img; }
    operator const std::vector<types>&() const { return raw_img; }
    // 8-bit copy (Imgraw.hpp:59-82): what the stitcher's Raw images are filled from
    operator std::vector<unsigned char>() const {
        std::vector<unsigned char> img(raw_img.size());
        for (size_t i = 0; i < raw_img.size(); ++i)
            img[i] = nomal ? (unsigned char)(raw_img[i] * ImgRawPixMax) : (unsigned char)raw_img[i];
        return img;
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
