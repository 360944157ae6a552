// Host-side mirror of the reference's Sift/imglib/imglib.hpp:13-57 for the
// functions on the SIFT path.  Same class and function names, argument meaning
// and error behaviour (file_same when in == out); the pixels are processed by
// the HIP library, never on the CPU.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

class file_same : public std::runtime_error {
public:
    explicit file_same(const std::string& str) : std::runtime_error(str) {}
};

class Gaus {
public:
    // imglib.cpp:72-118; mat_len = 0 selects round(6p+1)|1 taps
    static void GauBlur(std::vector<float>& img_gau, const std::vector<float>& img_ori, size_t width, size_t height,
                        float p, size_t mat_len = 0);
    // imglib.cpp:127-155
    static std::vector<float> gau_matrix(float p, size_t mat_len = 0);
};

class Scaling {
public:
    // imglib.cpp:227-252
    static void first(std::vector<float>& img, std::vector<float>& img_ori, size_t width, size_t height, float Ratio);
    // imglib.cpp:203-226 (one sample; plain host arithmetic)
    static float bilinear(const std::vector<float>& img, size_t width, float y, float x);
};
