// TEST INFRASTRUCTURE ONLY -- CPU oracle for the two geometric warps between
// RANSAC and the blender: _WarpPerspective and warping (Sift/stitch/stitch.cpp).
// Only tests may load this; the product never links or calls it.
//
// PARITY UNPINNED: stitch.cpp includes <timer.hpp>, which does not exist in this
// environment, so it cannot be compiled here, and the reference ships no
// vectors for it.  The functions below restate it operation by operation (all
// float, same operand order); the build keeps contraction off.
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace {

// stitch.cpp:19-63.  A sample whose 2x2 neighbourhood leaves the image is black.
inline void bilinear(const uint8_t* rgb, int cols, int rows, float _x, float _y, uint8_t out[3]) {
    const int x = (int)_x, y = (int)_y;
    const float l_x = _x - (float)x, r_x = 1.f - l_x;
    const float t_y = _y - (float)y, b_y = 1.f - t_y;
    float c[3] = {0.f, 0.f, 0.f};
    if (x >= 0 && x < cols - 1 && y >= 0 && y < rows - 1) {
        for (int k = 0; k < 3; k++) {
            c[k] = (float)rgb[((size_t)y * cols + x) * 3 + k] * (r_x * b_y);
            c[k] += (float)rgb[((size_t)y * cols + x + 1) * 3 + k] * (l_x * b_y);
            c[k] += (float)rgb[((size_t)(y + 1) * cols + x) * 3 + k] * (r_x * t_y);
            c[k] += (float)rgb[((size_t)(y + 1) * cols + x + 1) * 3 + k] * (l_x * t_y);
        }
    }
    for (int k = 0; k < 3; k++) out[k] = (unsigned char)(c[k] > 255.0 ? 255 : c[k] < 0.0 ? 0 : c[k]);
}

}  // namespace

extern "C" {

// stitch.cpp:106-141.  dst is in/out: pixels whose source position falls outside src keep their value.
void orc_warp_perspective(const uint8_t* src, int s_col, int s_row, const float* H, uint8_t* dst, int d_col, int d_row) {
    for (int i = 0; i < d_row; ++i) {
        uint8_t* dst_rgb = dst + (size_t)i * d_col * 3;
        for (int j = 0; j < d_col; ++j) {
            float x = (H[2] - H[8] * (float)j) * (H[4] - H[7] * (float)i) - (H[1] - H[7] * (float)j) * (H[5] - H[8] * (float)i);
            x /= (H[1] - H[7] * (float)j) * (H[3] - H[6] * (float)i) - (H[0] - H[6] * (float)j) * (H[4] - H[7] * (float)i);
            float y = (H[0] - H[6] * (float)j) * (H[5] - H[8] * (float)i) - (H[2] - H[8] * (float)j) * (H[3] - H[6] * (float)i);
            y /= (H[1] - H[7] * (float)j) * (H[3] - H[6] * (float)i) - (H[0] - H[6] * (float)j) * (H[4] - H[7] * (float)i);
            if (x < (float)s_col && x >= 0.0 && y < (float)s_row && y >= 0.0) bilinear(src, s_col, s_row, x, y, dst_rgb + j * 3);
        }
    }
}

// stitch.cpp:178-221 for one image.  out is w*h*3, zero-filled by the caller (the reference resizes a fresh Raw);
// upedge/downedge receive w (x, y) pairs: the source positions of the first and last output rows.
void orc_warp_cylinder(const uint8_t* src, int w, int h, float FL, uint8_t* out, float* upedge, float* downedge) {
    const float mid_x = (float)w / 2.f, mid_y = (float)h / 2.f;
    for (int j = 0; j < h; j++) {
        for (int i = 0; i < w; i++) {
            const float k = sqrtf(FL * FL + (i - mid_x) * (i - mid_x)) / FL;
            const float x = (i - mid_x) * k + mid_x;
            const float y = (j - mid_y) * k + mid_y;
            if (x >= 0 && y >= 0 && x < (float)w && y < (float)h) bilinear(src, w, h, x, y, out + ((size_t)j * w + i) * 3);
            if (j == 0) {
                upedge[2 * i] = x;
                upedge[2 * i + 1] = y;
            } else if (j == h - 1) {
                downedge[2 * i] = x;
                downedge[2 * i + 1] = y;
            }
        }
    }
}

}  // extern "C"
