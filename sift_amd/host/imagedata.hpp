// Keypoint record shared with the RANSAC / blend consumers.
// Mirrors the public layout of the reference's Sift/imagedata.hpp:9-86 (fpoint,
// Raw, Feature) so that code written against it compiles unchanged: same member
// names, types and order, same rX()/rY() helpers.
#pragma once
#include <cstddef>
#include <vector>

struct fpoint {
    fpoint() : x(0.f), y(0.f) {}
    fpoint(float px, float py) : x(px), y(py) {}
    float x;
    float y;
};

// Interleaved 8-bit RGB image of the stitcher (imagedata.hpp:17-33).
class Raw {
public:
    Raw() = default;
    Raw(size_t w, size_t h) : RGB(w * h * 3), col(w), row(h) {}
    void resize(size_t w, size_t h) {
        RGB.resize(w * h * 3);
        col = w, row = h;
    }
    int getCol() const { return (int)col; }
    int getRow() const { return (int)row; }

public:
    std::vector<unsigned char> RGB;

protected:
    size_t col = 0;
    size_t row = 0;
};

struct Feature {
    float size;                    // octave scale: 2, 1, 0.5, ...
    int kai;                       // DoG layer
    float sigmaOCT;                // integer-truncated sigma of the layer
    int x, y;                      // coordinates in the octave image
    float mm;                      // orientation-histogram bin value
    int sita;                      // orientation in degrees (multiple of 10)
    Feature* nextptr = nullptr;    // unused once features are in an array
    float descr[128] = {};         // integer-valued descriptor
    int d = 0;                     // descriptor length (128 after describe)
    Feature* fwd_match = nullptr;  // matched feature in the other image (not owned)
    fpoint img_pt;
    fpoint mdl_pt;
    void* feature_data = nullptr;  // scratch slot used by the kd-tree / RANSAC consumers
    float d_l;

    float rX() const { return x / size; }
    float rY() const { return y / size; }
};
