// The compositing half of the stitcher over the C ABI (reference: Sift/stitch/Blend.cpp).
#include "Blend.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../include/sift_hip.h"
#include "Sift.hpp"
#include "stitch.hpp"

namespace {

void check(sift_hip_ctx* ctx, int rc, const char* what) {
    if (rc == SIFT_HIP_OK) return;
    std::string msg = std::string(what) + ": " + sift_hip_strerror(rc);
    const char* detail = sift_hip_last_error(ctx);
    if (detail && *detail) msg += std::string(" (") + detail + ")";
    throw std::runtime_error(msg);
}

// fastAtanf_rad, Imgraw.cpp:102-124: atan(t) in radians from the 7th-order odd polynomial
float atan_poly_rad(float t) {
    const float p1 = 0.9997878412794807f, p3 = -0.3258083974640975f, p5 = 0.1555786518463281f, p7 = -0.04432655554792128f;
    const float eps = (float)2.2204460492503131e-016;
    const float mag = std::abs(t);
    float angle;
    if (mag <= 1.0f) {
        const float c = mag / (1.0f + eps), c2 = c * c;
        angle = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        const float c = 1.0f / (mag + eps), c2 = c * c;
        angle = (float)(3.14159265358979323846 * 0.5 - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c);
    }
    return t < 0 ? -angle : angle;
}

// One focal-length estimate from a pair of candidate squares (focalsFromHomography, Blend.cpp:827-865):
// the larger candidate wins unless both are positive, in which case the better-conditioned one does.
bool focal_from(float d1, float d2, float v1, float v2, float* f) {
    if (v1 < v2) std::swap(v1, v2);
    if (v1 > 0 && v2 > 0) {
        *f = std::sqrt(std::abs(d1) > std::abs(d2) ? v1 : v2);
        return true;
    }
    if (v1 > 0) {
        *f = std::sqrt(v1);
        return true;
    }
    return false;
}

// Cylinder coordinates (column, row) of image position (px, py) in an image of the given centre,
// rounded to pixels (alignMatch, Blend.cpp:772-789); x_shift is added before rounding.
void cylinder_pixel(float px, float py, float mid_x, float mid_y, float fl, float fl_sq, float x_shift, int* cx, int* cy) {
    const float theta = atan_poly_rad((px - mid_x) / fl);
    float hgt = py - mid_y;
    hgt /= std::sqrt(std::pow((px - mid_x), 2) + fl_sq);  // double, as pow(float, int) is
    *cx = (int)std::round(fl * theta + mid_x + x_shift);
    *cy = (int)std::round(fl * hgt + mid_y);
}

// alignMatch (Blend.cpp:745-825): mean offset of the second image against the first over the
// consensus matches (all but the last, as the reference loops), then its parity fix-ups.
// Columns and rows of an image, which is all alignMatch reads of its two Raw arguments.
struct Extent {
    int col, row;
    int getCol() const { return col; }
    int getRow() const { return row; }
};

void align_offsets(const Extent& img1, const Extent& img2, const Feature* const* good, int n, float fl, int* dx, int* dy) {
    const float mid_x1 = (float)img1.getCol() / 2.f, mid_x2 = (float)img2.getCol() / 2.f;
    const float mid_y1 = (float)img1.getRow() / 2.f, mid_y2 = (float)img2.getRow() / 2.f;
    const float fl_sq = std::pow(fl, 2);
    int sum_dx = 0, sum_dy = 0;
    for (int i = 0; i < n - 1; i++) {
        const Feature* in2 = good[i];
        const Feature* in1 = in2->fwd_match;
        int x1, y1, x2, y2;
        cylinder_pixel(in1->rX(), in1->rY(), mid_x1, mid_y1, fl, fl_sq, 0.f, &x1, &y1);
        cylinder_pixel(in2->rX(), in2->rY(), mid_x2, mid_y2, fl, fl_sq, (float)img1.getCol(), &x2, &y2);
        sum_dx += x2 - x1;
        sum_dy += img1.getRow() - y1 + y2;
    }
    int ox = (int)((float)sum_dx / (float)(n - 1));
    int oy = (int)((float)sum_dy / (float)(n - 1));
    if (ox % 2 == 0) ox += (ox + 1 <= img1.getCol() && ox + 1 <= img2.getCol()) ? 1 : -1;
    if (oy % 2 == 0) oy += (oy + 1 <= img1.getRow() && oy + 1 <= img2.getRow()) ? 1 : -1;
    else if (oy % 2 == 1) oy += 1;
    *dx = ox;
    *dy = oy;
}

}  // namespace

void multiBandBlend(Raw& limg, Raw& rimg, int dx, int dy) {
    sift_amd::ContextUse use;
    check(use.ctx, sift_hip_multiband_blend(use.ctx, limg.RGB.data(), limg.getCol(), limg.getRow(), rimg.RGB.data(),
                                            rimg.getCol(), rimg.getRow(), dx, dy),
          "sift_hip_multiband_blend");
}

float getFocal(const std::vector<float>& H, size_t img1Size, size_t img2Size) {
    if (H.size() >= 9) {
        float f0 = 0.f, f1 = 0.f;
        // f1 from the bottom row, f0 from the translation column (Blend.cpp:832-864)
        const bool ok1 = focal_from(H[6] * H[7], (H[7] - H[6]) * (H[7] + H[6]), -(H[0] * H[1] + H[3] * H[4]) / (H[6] * H[7]),
                                    (H[0] * H[0] + H[3] * H[3] - H[1] * H[1] - H[4] * H[4]) / ((H[7] - H[6]) * (H[7] + H[6])),
                                    &f1);
        const float e1 = H[0] * H[3] + H[1] * H[4], e2 = H[0] * H[0] + H[1] * H[1] - H[3] * H[3] - H[4] * H[4];
        const bool ok0 = focal_from(e1, e2, -H[2] * H[5] / e1, (H[5] * H[5] - H[2] * H[2]) / e2, &f0);
        if (ok0 && ok1) return sqrtf(f0 * f1);  // one pair of images: the median of one estimate
    }
    float sizes = 0.f;
    sizes += img1Size + img2Size;
    return sizes / 2.f;
}

void blen2img(const ImgRaw& img1, const ImgRaw& img2, ImgRaw& dst, const std::vector<float>& HomogMat,
              const Feature* const* RANSAC_feat, int RANSAC_num) {
    if (img1.bitCount != 24 || img2.bitCount != 24) throw std::runtime_error("blen2img: both images must be 24-bit RGB");
    if (!RANSAC_feat || RANSAC_num < 2) throw std::runtime_error("blen2img: needs at least two consensus matches");
    sift_amd::StepTimer tm("blen2img");
    // focal length and alignment need the sizes of the images only
    const Extent input[2] = {{(int)img1.width, (int)img1.height}, {(int)img2.width, (int)img2.height}};
    const float ft = getFocal(HomogMat, img1.size(), img2.size());
    int dx = 0, dy = 0;
    align_offsets(input[0], input[1], RANSAC_feat, RANSAC_num, ft, &dx, &dy);
    // the cylindrical projections keep the size of their images (stitch.cpp:129-131)
    const int r0 = input[0].getRow(), c0 = input[0].getCol(), r1 = input[1].getRow(), c1 = input[1].getCol();
    // an offset beyond the image height means the second image lies above the first (Blend.cpp:999-1003)
    const bool above = dy > r0;
    const int blend_dy = above ? -(r0 - std::abs(r0 - dy)) : dy;
    // canvas: the first image a tenth of its height down, the second dx columns into its right
    // edge and shifted by the row offset; 150 spare rows (Blend.cpp:1010-1031)
    const int cols = (c0 - dx) + c1;
    const int rows2 = (above ? 0 : r0 - dy) + r1 + 150;
    if (cols <= 0 || rows2 <= 0) throw std::runtime_error("blen2img: the alignment leaves no canvas");
    const int top0 = r0 / 10;
    const int left1 = c0 - dx;
    const int top1 = above ? top0 - std::abs(r0 - dy) : top0 + std::abs(r0 - dy);
    // warping (Blend.cpp:974), multiBandBlend (:994-996) and the canvas loops (:1010-1031; rows and
    // columns that leave the canvas are dropped, where the reference's buffer, tall enough for
    // images of up to 1500 rows only, has no guard) in one device call: two images up, the canvas down
    // The 8-bit copies of both images (Blend.cpp:907-916, ImgRaw's conversion) are made by a few threads straight into
    // a page-locked buffer kept beside the shared context, and the canvas comes back into the same buffer: both
    // transfers are direct DMA (from the float vectors the upload alone would be four times the bytes through the
    // runtime's staging copies).
    const size_t n1 = (size_t)c0 * r0 * 3, n2 = (size_t)c1 * r1 * 3, nc = (size_t)cols * rows2 * 3;
    const size_t off2 = (n1 + 255) / 256 * 256, offc = off2 + (n2 + 255) / 256 * 256;
    static unsigned char* pin = nullptr;  // only touched while `use` holds the context lock
    static size_t pin_cap = 0;
    {
        sift_amd::ContextUse use;
        if (offc + nc > pin_cap) {
            sift_hip_host_free(use.ctx, pin);
            pin = nullptr;
            pin_cap = 0;
            void* p = nullptr;
            check(use.ctx, sift_hip_host_alloc(use.ctx, &p, offc + nc), "sift_hip_host_alloc");
            pin = (unsigned char*)p;
            pin_cap = offc + nc;
        }
        img1.to_bytes(pin);
        img2.to_bytes(pin + off2);
        tm.step("8-bit copies");
        check(use.ctx, sift_hip_panorama_pair(use.ctx, pin, c0, r0, pin + off2, c1, r1, ft, dx, blend_dy, top0, left1, top1,
                                              pin + offc, cols, rows2),
              "sift_hip_panorama_pair");
        tm.step("sift_hip_panorama_pair");
        dst = ImgRaw(pin + offc, nc, (uint32_t)cols, (uint32_t)rows2, 24);
    }
    tm.step("canvas to float");
}
