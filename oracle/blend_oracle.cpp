// TEST INFRASTRUCTURE ONLY -- CPU oracle for the compositing half of the stitcher:
// multiBandBlend, alignMatch, getFocal and the canvas assembly of blen2img
// (Sift/stitch/Blend.cpp).  Only tests may load this; the product never links or
// calls it.
//
// PARITY UNPINNED: Blend.cpp includes <timer.hpp> and back-slashed header paths that
// do not exist in this environment, so it cannot be compiled here, and the reference
// ships no vectors for it.  The functions below restate it operation by operation
// (operand types and order kept: float products, double sums where the reference's
// expressions promote, pow() on (float, int) evaluated in double as C++11 does).
//
// Where the reference reads memory it never wrote, this restatement fixes a value and
// says so:
//   * `new float[]` buffers it reads before writing (sub() outside the smaller
//     operand, the overlap copy outside the image) count as 0.0f, which is what a
//     fresh allocation of that size holds in practice;
//   * GauBlur3d's column pass reads channel index +3 of the last pixel, one float past
//     the array: taken as 0.0f;
//   * the "extend the black border" loops read row ur+1 / column ur+1, which can be one
//     past the overlap when the first non-black sample is the last one: clamped.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int orc_gau_matrix(float p, int mat_len, float* out, int cap);  // sift_oracle.cpp (pinned to imglib.cpp)
extern "C" void orc_warp_cylinder(const uint8_t* src, int w, int h, float FL, uint8_t* out, float* upedge,
                                  float* downedge);  // stitch_oracle.cpp

namespace {

constexpr int PYR_OCTAVE = 5;  // Blend.cpp:30
constexpr int PYR_R = 2;       // Blend.cpp:33
constexpr int LEFT = 0, RIGHT = 1;

struct Img {  // Blend_Image (Blend.hpp:4-9) with value semantics; RGB zero-initialised
    int width = 0, height = 0;
    std::vector<float> RGB;
    void alloc(int w, int h) {
        width = w;
        height = h;
        RGB.assign((size_t)std::max(w, 0) * (size_t)std::max(h, 0) * 3, 0.f);
    }
};

// Blend.cpp:188-197.  pow(float, int) promotes to double; the result is rounded to float on return.
float BiCubicPoly(float x) {
    const float abs_x = std::abs(x);
    const float a = -0.5f;
    if (abs_x <= 1.f) {
        return (float)((a + 2.f) * std::pow((double)abs_x, 3) - (a + 3.f) * std::pow((double)abs_x, 2) + 1.f);
    } else if (abs_x < 2.0) {
        return (float)(a * std::pow((double)abs_x, 3) - 5.f * a * std::pow((double)abs_x, 2) + 8.f * a * abs_x - 4.f * a);
    }
    return 0.f;
}

// Blend.cpp:199-367 for the two diagonal matrices it is called with (s = 0.5 or 2).
Img Bicubic(const Img& src, float s) {
    Img dst;
    float x = src.width * 1.0f, y = 0.0f;
    const float u1 = x * s + y * 0.f, v1 = x * 0.f + y * s;
    x = src.width * 1.0f;
    y = src.height * 1.0f;
    const float u2 = x * s + y * 0.f, v2 = x * 0.f + y * s;
    x = 0.0f;
    y = src.width * 1.0f;  // (sic: the width, Blend.cpp:219)
    const float u3 = x * s + y * 0.f, v3 = x * 0.f + y * s;
    const float left = std::min(std::min(std::min(0.0f, u1), u2), u3);
    const float right = std::max(std::max(std::max(0.0f, u1), u2), u3);
    const float top = std::min(std::min(std::min(0.0f, v1), v2), v3);
    const float down = std::max(std::max(std::max(0.0f, v1), v2), v3);
    dst.alloc(int(std::abs(right - left)), int(std::abs(down - top)));
    for (int i = 0; i < dst.height; ++i) {
        float* p = &dst.RGB[(size_t)i * dst.width * 3];
        for (int j = 0; j < dst.width; ++j) {
            x = (j + left) / s;
            y = (i + top) / s;
            const int x0 = int(x) - 1, y0 = int(y) - 1, x1 = int(x), y1 = int(y);
            const int x2 = int(x) + 1, y2 = int(y) + 1, x3 = int(x) + 2, y3 = int(y) + 2;
            if ((x0 >= 0) && (x3 < src.width) && (y0 >= 0) && (y3 < src.height)) {
                const float* q[4] = {&src.RGB[(size_t)y0 * src.width * 3], &src.RGB[(size_t)y1 * src.width * 3],
                                     &src.RGB[(size_t)y2 * src.width * 3], &src.RGB[(size_t)y3 * src.width * 3]};
                const int xs[4] = {x0, x1, x2, x3};
                const float dx[4] = {BiCubicPoly(x - x0), BiCubicPoly(x - x1), BiCubicPoly(x - x2), BiCubicPoly(x - x3)};
                const float dy[4] = {BiCubicPoly(y - y0), BiCubicPoly(y - y1), BiCubicPoly(y - y2), BiCubicPoly(y - y3)};
                float wgt[4][4];  // [x][y]
                for (int a = 0; a < 4; a++)
                    for (int b = 0; b < 4; b++) wgt[a][b] = dx[a] * dy[b];
                for (int c = 0; c < 3; c++) {
                    // the reference's sum runs x-major, y inside, left to right
                    float acc = q[0][3 * xs[0] + c] * wgt[0][0];
                    for (int a = 0; a < 4; a++)
                        for (int b = 0; b < 4; b++)
                            if (a || b) acc = acc + q[b][3 * xs[a] + c] * wgt[a][b];
                    p[3 * j + c] = acc;
                }
                const float thre = 198.f;
                if ((std::abs(p[3 * j] - q[1][3 * x1]) > thre) || (std::abs(p[3 * j + 1] - q[1][3 * x1 + 1]) > thre) ||
                    (std::abs(p[3 * j + 2] - q[1][3 * x1 + 2]) > thre)) {
                    p[3 * j] = q[1][3 * x1];
                    p[3 * j + 1] = q[1][3 * x1 + 1];
                    p[3 * j + 2] = q[1][3 * x1 + 2];
                }
            } else {
                p[j * 3 + 0] = p[j * 3 + 1] = p[j * 3 + 2] = 0.f;
            }
        }
    }
    return dst;
}

// Blend.cpp:370-429 + BlurImage :430-451.  Row pass per channel; the column pass writes
// G from channel 2 and B from channel 3 (= R of the next pixel) of the row-pass image, as written.
Img BlurImage(const Img& img, float sigma, int r) {
    const size_t width = (size_t)img.width, height = (size_t)img.height;
    std::vector<float> g(64);
    const int K = orc_gau_matrix(sigma, r * 2 + 1, g.data(), (int)g.size());
    g.resize((size_t)K);
    std::vector<float> gx(img.RGB.size());
    const long rr = (long)(g.size() / 2);
    for (size_t j = 0; j < height; ++j)
        for (size_t i = 0; i < width; ++i) {
            double s[3] = {0, 0, 0};
            for (size_t k = 0; k < g.size(); ++k) {
                long idx = (long)i - rr + (long)k;
                idx = idx < 0 ? 0 : (idx > (long)width - 1 ? (long)width - 1 : idx);
                for (int c = 0; c < 3; c++) s[c] += img.RGB[(j * width + (size_t)idx) * 3 + c] * g[k];
            }
            for (int c = 0; c < 3; c++) gx[(j * width + i) * 3 + c] = (float)s[c];
        }
    Img dst;
    dst.alloc(img.width, img.height);
    auto at = [&](size_t e) -> float { return e < gx.size() ? gx[e] : 0.f; };  // one float past the end: 0
    for (size_t j = 0; j < height; ++j)
        for (size_t i = 0; i < width; ++i) {
            double sR = 0, sG = 0, sB = 0;
            for (size_t k = 0; k < g.size(); ++k) {
                long idx = (long)j - rr + (long)k;
                idx = idx < 0 ? 0 : (idx > (long)height - 1 ? (long)height - 1 : idx);
                const size_t e = ((size_t)idx * width + i) * 3;
                sR += at(e + 0) * g[k];
                sG += at(e + 2) * g[k];
                sB += at(e + 3) * g[k];
            }
            dst.RGB[(j * width + i) * 3 + 0] = (float)sR;
            dst.RGB[(j * width + i) * 3 + 1] = (float)sG;
            dst.RGB[(j * width + i) * 3 + 2] = (float)sB;
        }
    return dst;
}

// Blend.cpp:70-90; outside n's extent the reference leaves the new buffer unwritten: 0 here.
Img sub(const Img& m, const Img& n) {
    Img out;
    out.alloc(m.width, m.height);
    for (int i = 0; i < out.height; i++)
        for (int j = 0; j < out.width; j++)
            if (i < n.height && j < n.width)
                for (int c = 0; c < 3; c++)
                    out.RGB[((size_t)i * out.width + j) * 3 + c] =
                        m.RGB[((size_t)i * m.width + j) * 3 + c] - n.RGB[((size_t)i * n.width + j) * 3 + c];
    return out;
}

// Blend.cpp:166-185 (the form multiBandBlend calls: dx = x, dy = 0).
std::vector<float> getGaussianKernel_rr(int x, int y, int dx) {
    std::vector<float> kernel((size_t)std::max(x, 0) * (size_t)std::max(y, 0), 0.f);
    const float half = (dx - 1) / 2.f;
    const float sigma = std::sqrt((-1.f) * std::pow((float)x - 1.f - half, 2.f) / (2.f * std::log(0.5f)));
    for (int i = (x - dx); i < x; i++) {
        float g;
        if (i <= (x - half)) {
            g = std::exp((-1) * i * i / (2 * sigma * sigma));  // int product converted to float, float exp
        } else {
            g = 1.f - std::exp((-1.f) * std::pow(x - i - 1.f, 2.f) / (2.f * sigma * sigma));
        }
        for (int j = 0; j < y; j++) kernel[(size_t)j * x + i] = g;
    }
    return kernel;
}

// Blend.cpp:492-636.
std::vector<unsigned char> buildLaplacianMap(const uint8_t* rgb, int imgW, int imgH, std::vector<Img>& out, int dx, int dy,
                                             int chkLR) {
    std::vector<unsigned char> flags;
    Img tmp;
    tmp.alloc(dx, std::abs(dy));
    int disx = (chkLR == RIGHT) ? 0 : (imgW - dx);
    const int disy = (dy >= 0) ? ((chkLR == RIGHT) ? 0 : (imgH - dy)) : ((chkLR == RIGHT) ? (imgH + dy) : 0);
    if (disx < 0) disx = 0;
    for (int j = 0; j < tmp.height; j++)
        for (int i = 0; i < tmp.width; i++) {
            bool flag = false;
            // (a negative disy indexes before the image in the reference; such rows stay empty here)
            if ((j + disy) < imgH && (i + disx) < imgW && (j + disy) >= 0)
                for (int c = 0; c < 3; c++) {
                    const uint8_t v = rgb[((size_t)(j + disy) * imgW + (i + disx)) * 3 + c];
                    tmp.RGB[((size_t)j * tmp.width + i) * 3 + c] = (float)v;
                    if (v != 0) flag = true;
                }
            flags.push_back(flag);
        }
    // extend the image into its black border: columns first, then rows; ur / dr carry over from one
    // column (row) to the next when a column (row) holds nothing but black, and from the columns to the rows
    const int W = tmp.width, H = tmp.height;
    auto px = [&](int i, int j) -> float* { return &tmp.RGB[((size_t)i * W + j) * 3]; };
    auto lit = [&](int i, int j) { const float* p = px(i, j); return p[0] != 0.f || p[1] != 0.f || p[2] != 0.f; };
    auto put = [&](int i, int j, int si, int sj) {  // pixel (i, j) = pixel (si, sj); the source is never inside the run being written
        const float v[3] = {px(si, sj)[0], px(si, sj)[1], px(si, sj)[2]};
        std::memcpy(px(i, j), v, sizeof v);
    };
    const int set_r = 1;
    int ur = 0, dr = 0;
    for (int j = 0; j < W; j++) {
        for (int i = 0; i < H; i++)
            if (lit(i, j)) { ur = i; break; }
        for (int i = ur; i >= 0; i--) put(i, j, std::min(ur + set_r, H - 1), j);
        for (int i = H - 1; i >= 0; i--)
            if (lit(i, j)) { dr = i; break; }
        dr = dr < 1 ? 1 : dr;
        for (int i = dr; i < H; i++) put(i, j, dr - set_r, j);
    }
    for (int i = 0; i < H; i++) {
        for (int j = 0; j < W; j++)
            if (lit(i, j)) { ur = j; break; }
        for (int j = std::min(ur, W - 1); j >= 0; j--) put(i, j, i, std::min(ur + set_r, W - 1));
        for (int j = W - 1; j >= 0; j--)
            if (lit(i, j)) { dr = j; break; }
        dr = dr < 1 ? 1 : dr;
        for (int j = dr; j < W; j++) put(i, j, i, dr - set_r);
    }
    out.clear();
    out.resize(PYR_OCTAVE);
    out[0] = tmp;
    const float sigma = std::sqrt((-1) * std::pow(2.f, 2.f) / (2.f * std::log(0.5f)));
    for (int i = 1; i < PYR_OCTAVE; i++) out[i] = Bicubic(BlurImage(out[i - 1], sigma, PYR_R), 0.5f);
    for (int a = 0; a < PYR_OCTAVE - 1; a++) out[a] = sub(out[a], Bicubic(out[a + 1], 2.0f));
    return flags;
}

// Blend.cpp:454-490.
void blendImg(uint8_t* rgb, int imgW, int imgH, const Img& ov, int dx, int dy, int lr, const std::vector<unsigned char>& bol,
              const std::vector<unsigned char>& bor) {
    int disx = (lr == RIGHT) ? 0 : (imgW - dx);
    int disy = dy >= 0 ? ((lr == RIGHT) ? 0 : (imgH - dy)) : ((lr == RIGHT) ? (imgH + dy) : 0);
    if (disy < 0) disy = 0;
    if (disx < 0) disx = 0;
    const int currW = ov.width;
    for (int j = 0; j < ov.height; j++)
        for (int i = 0; i < currW - 1; i++)
            if (bol[(size_t)j * currW + i] && bor[(size_t)j * currW + i] && bol[(size_t)j * currW + i + 1] &&
                bor[(size_t)j * currW + i + 1])
                if (j + disy < imgH && i + disx < imgW)
                    for (int c = 0; c < 3; c++) {
                        const float v = ov.RGB[((size_t)j * currW + i) * 3 + c];
                        rgb[((size_t)(j + disy) * imgW + (i + disx)) * 3 + c] = (unsigned char)((v < 0.f) ? 0 : (v > 255.f) ? 255 : v);
                    }
}

// Imgraw.cpp:102-124
float fastAtanf_rad(float dy) {
    static const float p1 = 0.9997878412794807f, p3 = -0.3258083974640975f, p5 = 0.1555786518463281f,
                       p7 = -0.04432655554792128f;
    static const float eps = 2.2204460492503131e-016;
    const float ax = 1.0, ay = std::abs(dy);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + static_cast<float>(eps));
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + static_cast<float>(eps));
        c2 = c * c;
        a = 3.14159265358979323846 * 0.5 - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (dy < 0) a = -a;
    return a;
}

// Blend.cpp:827-865
void focalsFromHomography(const float* Hmg, float& f0, float& f1, bool& f0_ok, bool& f1_ok) {
    float d1, d2, v1, v2;
    f1_ok = true;
    d1 = Hmg[6] * Hmg[7];
    d2 = (Hmg[7] - Hmg[6]) * (Hmg[7] + Hmg[6]);
    v1 = -(Hmg[0] * Hmg[1] + Hmg[3] * Hmg[4]) / d1;
    v2 = (Hmg[0] * Hmg[0] + Hmg[3] * Hmg[3] - Hmg[1] * Hmg[1] - Hmg[4] * Hmg[4]) / d2;
    if (v1 < v2) std::swap(v1, v2);
    if (v1 > 0 && v2 > 0) f1 = std::sqrt(std::abs(d1) > std::abs(d2) ? v1 : v2);
    else if (v1 > 0) f1 = std::sqrt(v1);
    else f1_ok = false;
    f0_ok = true;
    d1 = Hmg[0] * Hmg[3] + Hmg[1] * Hmg[4];
    d2 = Hmg[0] * Hmg[0] + Hmg[1] * Hmg[1] - Hmg[3] * Hmg[3] - Hmg[4] * Hmg[4];
    v1 = -Hmg[2] * Hmg[5] / d1;
    v2 = (Hmg[5] * Hmg[5] - Hmg[2] * Hmg[2]) / d2;
    if (v1 < v2) std::swap(v1, v2);
    if (v1 > 0 && v2 > 0) f0 = std::sqrt(std::abs(d1) > std::abs(d2) ? v1 : v2);
    else if (v1 > 0) f0 = std::sqrt(v1);
    else f0_ok = false;
}

}  // namespace

extern "C" {

// Blend.cpp:638-740.  Both images (8-bit RGB) are updated in place inside their overlap.
void orc_multiband_blend(uint8_t* limg, int lw, int lh, uint8_t* rimg, int rw, int rh, int dx, int dy) {
    std::vector<Img> llpyr, rlpyr;
    const std::vector<unsigned char> bol = buildLaplacianMap(limg, lw, lh, llpyr, dx, dy, LEFT);
    const std::vector<unsigned char> bor = buildLaplacianMap(rimg, rw, rh, rlpyr, dx, dy, RIGHT);
    std::vector<Img> LS(PYR_OCTAVE);
    const Img& top = llpyr[llpyr.size() - 1];
    const std::vector<float> k = getGaussianKernel_rr(top.width, top.height, top.width);
    for (size_t a = 0; a < llpyr.size(); a++) {
        LS[a].alloc(llpyr[a].width, llpyr[a].height);
        const int center = (int)(llpyr[a].width / 2);
        const int w = LS[a].width;
        for (int j = 0; j < LS[a].height; j++)
            for (int i = 0; i < w; i++)
                for (int c = 0; c < 3; c++) {
                    const size_t e = ((size_t)j * w + i) * 3 + c;
                    if (a == llpyr.size() - 1) {
                        LS[a].RGB[e] = llpyr[a].RGB[e] * k[(size_t)j * w + i];
                        LS[a].RGB[e] += rlpyr[a].RGB[e] * (1.f - k[(size_t)j * w + i]);
                    } else if (i == center) {
                        LS[a].RGB[e] = (llpyr[a].RGB[e] + rlpyr[a].RGB[e]) / 2.f;
                    } else if (i > center) {
                        LS[a].RGB[e] = rlpyr[a].RGB[e];
                    } else {
                        LS[a].RGB[e] = llpyr[a].RGB[e];
                    }
                }
    }
    for (int a = PYR_OCTAVE - 1; a > 0; a--) {
        const Img result = Bicubic(LS[a], 2.0f);
        Img& lo = LS[a - 1];
        for (int j = 0; j < lo.height; j++)
            for (int i = 0; i < lo.width; i++)
                for (int c = 0; c < 3; c++)
                    if (j < result.height && i < result.width) {
                        float& v = lo.RGB[((size_t)j * lo.width + i) * 3 + c];
                        v = v + result.RGB[((size_t)j * result.width + i) * 3 + c];
                        v = v < 0.f ? 0.f : v > 255.f ? 255.f : v;
                    }
    }
    blendImg(limg, lw, lh, LS[0], dx, dy, LEFT, bol, bor);
    blendImg(rimg, rw, rh, LS[0], dx, dy, RIGHT, bol, bor);
}

// Blend.cpp:745-825.  a = positions of the consensus features (image 2), b = of their matches (image 1),
// n pairs (the reference loops over n - 1 of them).  out[0] = dx, out[1] = dy.
void orc_align_match(int w1, int h1, int w2, int h2, const float* ax, const float* ay, const float* bx, const float* by,
                     int n, float FL, int* out) {
    int cal_dx = 0, cal_dy = 0;
    const float mid_x1 = (float)w1 / 2.f, mid_x2 = (float)w2 / 2.f, mid_y1 = (float)h1 / 2.f, mid_y2 = (float)h2 / 2.f;
    const float fL1 = FL, fL2 = FL;
    const float fL1_pow = std::pow(fL1, 2), fL2_pow = std::pow(fL2, 2);  // pow(float, int) -> double -> float
    for (int i = 0; i < n - 1; i++) {
        const float imgX1 = ax[i], imgY1 = ay[i], imgX2 = bx[i], imgY2 = by[i];
        const float theta1 = fastAtanf_rad((imgX2 - mid_x1) / fL1);
        float h1_ = imgY2 - mid_y1;
        h1_ /= std::sqrt(std::pow((imgX2 - mid_x1), 2) + fL1_pow);
        const int x1 = (int)std::round(fL1 * theta1 + mid_x1);
        const int y1 = (int)std::round(fL1 * h1_ + mid_y1);
        const float theta2 = fastAtanf_rad((imgX1 - mid_x2) / fL2);
        float h2_ = imgY1 - mid_y2;
        h2_ /= std::sqrt(std::pow((imgX1 - mid_x2), 2) + fL2_pow);
        const int x2 = (int)std::round(fL2 * theta2 + mid_x2 + w1);
        const int y2 = (int)std::round(fL2 * h2_ + mid_y2);
        cal_dx += x2 - x1;
        cal_dy += h1 - y1 + y2;
    }
    int avg_dx = (float)cal_dx / (float)(n - 1);
    int avg_dy = (float)cal_dy / (float)(n - 1);
    if (avg_dx % 2 == 0) {
        if (avg_dx + 1 <= w1 && avg_dx + 1 <= w2) avg_dx += 1; else avg_dx -= 1;
    }
    if (avg_dy % 2 == 0) {
        if (avg_dy + 1 <= h1 && avg_dy + 1 <= h2) avg_dy += 1; else avg_dy -= 1;
    } else if (avg_dy % 2 == 1) {
        avg_dy += 1;
    }
    out[0] = avg_dx;
    out[1] = avg_dy;
}

// Blend.cpp:867-894 (H9 may be null: no homography).
float orc_get_focal(const float* H9, size_t img1Size, size_t img2Size) {
    const int img_total = 2;
    float f0 = 0.f, f1 = 0.f, ft = 0.f;
    bool f0ok = false, f1ok = false;
    std::vector<float> all_focals;
    if (H9) {
        focalsFromHomography(H9, f0, f1, f0ok, f1ok);
        if (f0ok && f1ok) all_focals.push_back(sqrtf(f0 * f1));
    }
    if ((int)all_focals.size() >= img_total - 1) {
        std::sort(all_focals.begin(), all_focals.end());
        if (all_focals.size() % 2 == 1) ft = all_focals[all_focals.size() / 2];
        else ft = (all_focals[all_focals.size() / 2 - 1] + all_focals[all_focals.size() / 2]) * 0.5f;
    } else {
        float focals_sum = 0.f;
        focals_sum += img1Size + img2Size;
        ft = focals_sum / (float)img_total;
    }
    return ft;
}

// Canvas size of blen2img for two images (Blend.cpp:1010-1031): out[0] = cols, out[1] = rows2.
void orc_blend_canvas_size(int w1, int h1, int w2, int h2, int dx, int dy, int* out) {
    int cols = 0, rows = 0;
    if (dy > h1) rows += h1 - h1; else rows += h1 - dy;
    rows += h2;
    cols += w1 - dx;
    cols += w2;
    out[0] = cols;
    out[1] = rows + 150;
}

// blen2img from the cylindrical projections on (Blend.cpp:969-1133): the two projected images,
// aligned by (dx, dy), are blended and placed on a zeroed canvas of orc_blend_canvas_size.
// w0/w1 are updated in place by the blend, as the reference updates warpingImg.
void orc_blend_compose(uint8_t* w0, int w0c, int w0r, uint8_t* w1, int w1c, int w1r, int dx, int dy, uint8_t* canvas,
                       int cols, int rows2) {
    if (dy > w0r) {
        const int dyy = -(w0r - std::abs(w0r - dy));
        orc_multiband_blend(w0, w0c, w0r, w1, w1c, w1r, dx, dyy);
    } else {
        orc_multiband_blend(w0, w0c, w0r, w1, w1c, w1r, dx, dy);
    }
    int distancey = w0r / 10;
    int distance = 0;
    for (int j = distancey; j < distancey + w0r; j++)
        for (int i = distance; i < w0c + distance; i++)
            if (j < rows2 && j >= 0 && i >= 0 && i < cols)  // (the reference tests i <= cols and can touch the next row's first pixel)
                for (int c = 0; c < 3; c++)
                    canvas[((size_t)j * cols + i) * 3 + c] = w0[((size_t)(j - distancey) * w0c + (i - distance)) * 3 + c];
    distance = distance + w0c - dx;
    if (dy > w0r) distancey = distancey - std::abs(w0r - dy);
    else distancey = distancey + std::abs(w0r - dy);
    for (int b = distancey; b < distancey + w1r; b++)
        for (int a = distance; a < w1c + distance; a++)
            if (b >= 0 && b < rows2 && a < cols) {  // (the reference does not test the canvas bounds here)
                const size_t idx = ((size_t)(b - distancey) * w1c + (a - distance)) * 3;
                if (w1[idx] != 0 || w1[idx + 1] != 0 || w1[idx + 2] != 0)
                    for (int c = 0; c < 3; c++) canvas[((size_t)b * cols + a) * 3 + c] = w1[idx + c];
            }
}

}  // extern "C"
