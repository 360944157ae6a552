// TEST INFRASTRUCTURE ONLY -- CPU oracle for the SIFT detect/describe/match path.
//
// A single-threaded restatement of the reference algorithm, written from the
// reference's behaviour (every numeric quirk kept: mixed float/double
// arithmetic, accumulation order, truncations).  Only tests, the smoke check and
// the cpu_baseline leg of the benchmark may load this library; the product
// (libsift_hip.so and the C++ host classes) never links or calls it.
//
// Pinning status: Gaus::gau_matrix / Gaus::GauBlur / Scaling::first are checked
// bit-for-bit against the reference's own imglib.cpp compiled into
// oracle/_ref (see oracle/Makefile and tests/test_oracle_pin.py).  The
// functions restated from Sift/Sift.cpp and Sift/Imgraw.cpp (detection,
// orientation, descriptor, brute-force matcher, fastAtan2f) are PARITY UNPINNED:
// those two translation units need headers that do not exist in this
// environment (timer.hpp, opencv2/opencv.hpp, back-slashed include paths) and
// the reference ships no tests or golden vectors.
//
// Build: g++ -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace {

// Sift/Sift.hpp:44-77
constexpr float kGauSigma = 1.6f;        // SIFT_GauSigma
constexpr float kDx = 0.03f;             // SIFT_Dx
constexpr float kDescrSclFctr = 3.f;     // SIFT_DESCR_SCL_FCTR
constexpr int kDescrHistRadius = 4;      // SIFT_DESCR_HIST_RADIUS (d)
constexpr int kDescrHistBins = 8;        // SIFT_DESCR_HIST_BINS (n)
constexpr double kDescrMagThr = 0.2;     // SIFT_DESCR_MAG_THR
constexpr double kIntDescrFctr = 512.0;  // SIFT_INT_DESCR_FCTR
constexpr double kPi = 3.14159265358979323846;  // Sift.cpp:25, Imgraw.cpp:24, imglib.cpp:16

using Img = std::vector<float>;

// ---------------------------------------------------------------------------
// Sift/imglib/imglib.cpp:120-125  Gaus::gau_meth
float gau_meth(size_t r, float p) {
    float two = 2.0;
    // pow(size_t,float) promotes to double; pow(float,float) stays float.
    float num = std::exp(-std::pow((double)r, (double)two) / (two * std::pow(p, two)));
    num /= std::sqrt(two * kPi) * p;
    return num;
}

// Sift/imglib/imglib.cpp:127-155  Gaus::gau_matrix
std::vector<float> gau_matrix(float p, size_t mat_len) {
    std::vector<float> gau_mat;
    if (mat_len == 0) {
        mat_len = (int)(std::round((p * 6 + 1))) | 1;
    }
    if (mat_len % 2 == 0) { ++mat_len; }
    gau_mat.resize(mat_len);
    float sum = 0;
    for (int i = 0, j = (int)(mat_len / 2); j < (int)mat_len; ++i, ++j) {
        float temp;
        if (i) {
            temp = gau_meth(i, p);
            gau_mat[j] = temp;
            gau_mat[mat_len - j - 1] = temp;
            sum += temp += temp;
        } else {
            gau_mat[j] = gau_meth(i, p);
            sum += gau_mat[j];
        }
    }
    for (auto& v : gau_mat) { v /= sum; }
    return gau_mat;
}

// Sift/imglib/imglib.cpp:72-118  Gaus::GauBlur (separable, clamp-to-edge,
// float product, double accumulate in ascending tap order).
void gau_blur(Img& dst, const Img& src, size_t width, size_t height, float p) {
    std::vector<float> gau_mat = gau_matrix(p, 0);
    dst.resize(height * width);
    Img tmp(src.size());
    const long r = (long)(gau_mat.size() / 2);
    const long K = (long)gau_mat.size();
    for (long j = 0; j < (long)height; ++j) {
        for (long i = 0; i < (long)width; ++i) {
            double sum = 0;
            for (long k = 0; k < K; ++k) {
                long idx = i - r + k;
                if (idx < 0) idx = 0;
                else if (idx > (long)width - 1) idx = (long)width - 1;
                sum += src[j * width + idx] * gau_mat[k];
            }
            tmp[j * width + i] = (float)sum;
        }
    }
    for (long j = 0; j < (long)height; ++j) {
        for (long i = 0; i < (long)width; ++i) {
            double sum = 0;
            for (long k = 0; k < K; ++k) {
                long idx = j - r + k;
                if (idx < 0) idx = 0;
                else if (idx > (long)height - 1) idx = (long)height - 1;
                sum += tmp[idx * width + i] * gau_mat[k];
            }
            dst[j * width + i] = (float)sum;
        }
    }
}

// Sift/imglib/imglib.cpp:203-226  Scaling::bilinear.  The neighbour indices are
// clamped to the image here; the reference reads one element past the row/image
// when ceil() lands outside, which is undefined there.
float bilinear(const Img& img, size_t width, size_t height, float y, float x) {
    size_t x0 = (size_t)std::floor(x);
    size_t x1 = (size_t)std::ceil(x);
    size_t y0 = (size_t)std::floor(y);
    size_t y1 = (size_t)std::ceil(y);
    float dx1 = x - x0;
    float dx2 = 1 - dx1;
    float dy1 = y - y0;
    float dy2 = 1 - dy1;
    if (x1 > width - 1) x1 = width - 1;
    if (y1 > height - 1) y1 = height - 1;
    const float A = img[y0 * width + x0];
    const float B = img[y0 * width + x1];
    const float C = img[y1 * width + x0];
    const float D = img[y1 * width + x1];
    float AB = A * dx2 + B * dx1;
    float CD = C * dx2 + D * dx1;
    float X = AB * dy2 + CD * dy1;
    return X;
}

// Sift/imglib/imglib.cpp:227-252  Scaling::first
void first(Img& dst, const Img& src, size_t width, size_t height, float ratio, int* outW, int* outH) {
    int newH = static_cast<int>(std::floor(height * ratio));
    int newW = static_cast<int>(std::floor(width * ratio));
    dst.resize((size_t)newH * newW);
    for (int j = 0; j < newH; ++j) {
        for (int i = 0; i < newW; ++i) {
            float srcY, srcX;
            if (ratio < 1) {
                srcY = ((j + 0.5f) / ratio) - 0.5;
                srcX = ((i + 0.5f) / ratio) - 0.5;
            } else {
                srcY = j * (height - 1.f) / (newH - 1.f);
                srcX = i * (width - 1.f) / (newW - 1.f);
            }
            dst[(size_t)j * newW + i] = bilinear(src, width, height, srcY, srcX);
        }
    }
    *outW = newW;
    *outH = newH;
}

// Sift/Imgraw.cpp:29-52  fastAtan2f (degrees, 7th-order odd polynomial)
float fast_atan2f(float dy, float dx) {
    static const float atan2_p1 = 0.9997878412794807f * (float)(180 / kPi);
    static const float atan2_p3 = -0.3258083974640975f * (float)(180 / kPi);
    static const float atan2_p5 = 0.1555786518463281f * (float)(180 / kPi);
    static const float atan2_p7 = -0.04432655554792128f * (float)(180 / kPi);
    static const float atan2_DBL_EPSILON = 2.2204460492503131e-016;
    float ax = std::abs(dx), ay = std::abs(dy);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + static_cast<float>(atan2_DBL_EPSILON));
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + static_cast<float>(atan2_DBL_EPSILON));
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (dx < 0) a = 180.f - a;
    if (dy < 0) a = 360.f - a;
    return a;
}

// Sift/Sift.cpp:130-139  getPyramidOctv
size_t pyramid_octaves(size_t width, size_t height) {
    int n = 1;
    for (float Limit = 1.0; Limit < width || Limit < height; ++n) {
        Limit = powf(2.0, n);
    }
    if (n < 7) { n = 1; } else { n -= 6; }
    return n;
}

struct Feat {  // the fields of Sift/imagedata.hpp:63-86 that the path fills
    float size;
    int kai;
    float sigmaOCT;
    int x, y;
    float mm;
    int sita;
    float hist[128];   // clamped raw histogram ("before quantisation")
    float descr[128];  // integer-valued descriptor
    float sub[3];      // sub-pixel offsets (xc, xr, xi) when the refinement is on, else 0
};

// Sift/Sift.cpp:86-104  findMaxMin
bool find_max_min(const std::vector<Img>& dog, size_t intvl_idx, size_t W, size_t y, size_t x) {
    const float Val = dog[intvl_idx][y * W + x];
    if (std::fabs(Val) > kDx / 2.0) {
        for (int k = (int)intvl_idx - 1; k <= (int)intvl_idx + 1; k++) {
            const Img& cur = dog[k];
            for (int j = (int)y - 1; j <= (int)y + 1; j++) {
                size_t ci = (size_t)j * W + x;
                for (int i = -1; i <= 1; i++) {
                    if (Val > 0 && Val < cur[ci + i]) return false;
                    if (Val < 0 && Val > cur[ci + i]) return false;
                }
            }
        }
        return true;
    }
    return false;
}

// Sift/Sift.cpp:105-129  harris_corner (called with r = histogram radius)
bool harris_corner(const Img& p, size_t w, size_t y, size_t x, float r) {
    float thre = ((r + 1) * (r + 1)) / r;
    auto at2d = [&](long yy, long xx) -> float { return p[yy * w + xx]; };
    float Dxx = 2.f * at2d(y, x) - at2d(y, x - 1) - at2d(y, x + 1);
    float Dyy = 2.f * at2d(y, x) - at2d(y - 1, x) - at2d(y + 1, x);
    float Dxy = at2d(y + 1, x + 1) + at2d(y - 1, x - 1) - at2d(y - 1, x + 1) - at2d(y + 1, x - 1);
    Dxy /= 4.f;
    float Tr = Dxx + Dyy;
    float Det = Dxx * Dyy - Dxy * Dxy;
    float val = (Tr * Tr / Det);
    return val < thre;
}

// Sift/Sift.cpp:250-295  getHistogramMS + FeatAppend (:229-249)
// quality != 0 (the mode the reference's own notes ask for, update.md:9-10; nothing in the
// reference to compare with): the weights come from weight_sigma, the orientation is the bin
// centre moved by the vertex of the parabola through the peak bin and its two neighbours, in
// whole degrees, and sigmaOCT keeps the un-truncated sigma.
void histogram_ms(std::vector<Feat>& out, const Img& img, size_t W, float Insize, size_t intvl_idx,
                  float sigma, size_t Iny, size_t Inx, size_t Inr, const float* sub = nullptr, int quality = 0,
                  float weight_sigma = 0.f) {
    const size_t matLen = Inr * 2 + 1;
    std::vector<float> mag(matLen * matLen);
    std::vector<float> sita(matLen * matLen);
    size_t idx = 0;
    for (long j = (long)Iny - (long)Inr; j <= (long)(Iny + Inr); j++) {
        for (long i = (long)Inx - (long)Inr; i <= (long)(Inx + Inr); i++, idx++) {
            const float dx = img[(j + 0) * W + (i + 1)] - img[(j + 0) * W + (i - 1)];
            const float dy = img[(j + 1) * W + (i + 0)] - img[(j - 1) * W + (i + 0)];
            mag[idx] = sqrtf(dx * dx + dy * dy);
            sita[idx] = fast_atan2f(dy, dx);
        }
    }
    std::vector<float> gauMat = gau_matrix(quality ? weight_sigma : sigma, matLen);
    for (size_t j = 0; j < matLen; j++)
        for (size_t i = 0; i < matLen; i++) mag[j * matLen + i] *= gauMat[i];
    for (size_t j = 0; j < matLen; j++)
        for (size_t i = 0; i < matLen; i++) mag[j * matLen + i] *= gauMat[j];
    // 36 bins; an angle that rounds to exactly 360.0f indexes a 37th slot, which
    // in the reference is a write one float past the vector that nothing reads.
    float magSum[37] = {0};
    for (size_t i = 0; i < mag.size(); i++) {
        magSum[(size_t)(sita[i] / 10.0)] += mag[i];
    }
    float maxMag = *std::max_element(magSum, magSum + 36);
    for (size_t i = 0; i < 36; i++) {
        if (magSum[i] >= maxMag * 0.8) {
            Feat f;
            std::memset(&f, 0, sizeof f);
            f.x = (int)Inx;
            f.y = (int)Iny;
            f.mm = magSum[i];
            f.sita = (int)(i * 10);
            f.size = Insize;
            f.kai = (int)intvl_idx;
            f.sigmaOCT = (float)(int)sigma;  // FeatAppend takes sigmaOCT as int
            if (quality) {
                const float l = magSum[(i + 35) % 36], c0 = magSum[i], r = magSum[(i + 1) % 36];
                const float den = l - 2.f * c0 + r;
                const float off = den < 0.f ? std::fmin(std::fmax(0.5f * (l - r) / den, -0.5f), 0.5f) : 0.f;
                f.sita = ((int)std::lround(10.f * ((float)i + 0.5f + off)) + 360) % 360;
                f.sigmaOCT = sigma;
            }
            if (sub) std::memcpy(f.sub, sub, sizeof f.sub);
            out.push_back(f);
        }
    }
}

// Sift/Sift.cpp:299-313  calc_grad_mag_ori
bool calc_grad_mag_ori(const Img& img, int col, int row, int r, int c, float& mag, float& ori) {
    if (r > 0 && r < row - 1 && c > 0 && c < col - 1) {
        float dx = img[(size_t)r * col + (c + 1)] - img[(size_t)r * col + (c - 1)];
        float dy = img[(size_t)(r - 1) * col + c] - img[(size_t)(r + 1) * col + c];
        mag = std::sqrt(dx * dx + dy * dy);
        ori = std::atan2(dy, dx);
        return true;
    }
    return false;
}

// Sift/Sift.cpp:314-341  interp_hist_entry
void interp_hist_entry(float* hist, float rbin, float cbin, float obin, float mag, int d, int n) {
    int r0 = (int)std::floor(rbin);
    int c0 = (int)std::floor(cbin);
    int o0 = (int)std::floor(obin);
    float d_r = rbin - r0;
    float d_c = cbin - c0;
    float d_o = obin - o0;
    for (int r = 0; r <= 1; r++) {
        int rb = r0 + r;
        if (rb >= 0 && rb < d) {
            float v_r = mag * ((r == 0) ? 1.f - d_r : d_r);
            for (int c = 0; c <= 1; c++) {
                int cb = c0 + c;
                if (cb >= 0 && cb < d) {
                    float v_c = v_r * ((c == 0) ? 1.f - d_c : d_c);
                    for (int o = 0; o <= 1; o++) {
                        int ob = (o0 + o) % n;
                        float v_o = v_c * ((o == 0) ? 1.f - d_o : d_o);
                        hist[(rb * d + cb) * n + ob] += v_o;
                    }
                }
            }
        }
    }
}

// Sift/Sift.cpp:342-387  descr_hist.  `ori` is the orientation in DEGREES and is
// fed to cos/sin/atan2 arithmetic as if it were radians, exactly as upstream.
void descr_hist(float* hist, const Img& img, int col, int row, int r, int c, float ori, float scl,
                int d, int n) {
    for (int i = 0; i < d * d * n; i++) hist[i] = 0.f;
    float cos_t = std::cos(ori);
    float sin_t = std::sin(ori);
    float PI2 = 2.f * kPi;
    float bins_per_rad = n / PI2;
    float exp_denom = d * d * 0.5f;
    float hist_width = (float)kDescrSclFctr * scl;
    int radius = (int)(hist_width * std::sqrt(2.f) * (d + 1.f) * 0.5f + 0.5f);
    for (int i = -radius; i <= radius; i++) {
        for (int j = -radius; j <= radius; j++) {
            float c_rot = (j * cos_t - i * sin_t) / hist_width;
            float r_rot = (j * sin_t + i * cos_t) / hist_width;
            float rbin = r_rot + d / 2.f - 0.5f;
            float cbin = c_rot + d / 2.f - 0.5f;
            if (rbin > -1.f && rbin < d && cbin > -1.f && cbin < d) {
                float grad_mag, grad_ori;
                if (calc_grad_mag_ori(img, col, row, r + i, c + j, grad_mag, grad_ori)) {
                    grad_ori -= ori;
                    while (grad_ori < 0.f) grad_ori += PI2;
                    while (grad_ori >= PI2) grad_ori -= PI2;
                    float obin = grad_ori * bins_per_rad;
                    float w = std::exp(-(c_rot * c_rot + r_rot * r_rot) / exp_denom);
                    interp_hist_entry(hist, rbin, cbin, obin, grad_mag * w, d, n);
                }
            }
        }
    }
}

// The descriptor the quality mode computes: Sift.cpp:342-387 with the orientation in RADIANS
// (Sift.cpp:351-352 feed degrees to cos/sin), measured in the sense the gradient angles of
// calc_grad_mag_ori are (row axis up, Sift.cpp:306; the orientation histogram measures with the
// row axis down, Sift.cpp:262, hence the minus sign), and the un-truncated sigma.
void descr_hist_quality(float* hist, const Img& img, int col, int row, int r, int c, int sita_deg, float scl,
                        int d, int n) {
    const float ori = -(float)sita_deg * (float)(kPi / 180.0);
    descr_hist(hist, img, col, row, r, c, ori, scl, d, n);
}

// Sift/Sift.cpp:389-399 normalize_descr as it was meant (over all 128 entries), then
// Sift.cpp:409-415: normalise, clamp at 0.2, normalise again, times 512, min 255.
void hist_to_descr_quality(Feat& f) {
    for (int pass = 0; pass < 2; pass++) {
        float len_sq = 0.f;
        for (int i = 0; i < 128; i++) len_sq += f.hist[i] * f.hist[i];
        const float len_inv = len_sq > 0.f ? 1.f / std::sqrt(len_sq) : 0.f;
        for (int i = 0; i < 128; i++) f.hist[i] *= len_inv;
        if (pass == 0)
            for (int i = 0; i < 128; i++)
                if (f.hist[i] > (float)kDescrMagThr) f.hist[i] = (float)kDescrMagThr;
    }
    for (int i = 0; i < 128; i++) f.descr[i] = (float)std::min(255, (int)(kIntDescrFctr * f.hist[i]));
}

// Sift/Sift.cpp:400-421  hist_to_descr.  Both normalize_descr() calls
// (:389-399) loop over feat->d, which is still 0 at that point, so they change
// nothing; the descriptor is the clamped, un-normalised histogram times 512.
void hist_to_descr(Feat& f) {
    for (int i = 0; i < 128; i++) {
        if (f.hist[i] > kDescrMagThr) f.hist[i] = (float)kDescrMagThr;
    }
    for (int i = 0; i < 128; i++) {
        int int_val = (int)(kIntDescrFctr * f.hist[i]);
        f.descr[i] = (float)std::min(255, int_val);
    }
}

// ---------------------------------------------------------------------------
// Sift/matrix.hpp:12-196  Lowe's sub-pixel refinement.  DEAD CODE in the reference
// (the header is included by no translation unit and the only call, Sift.cpp:187,
// is commented out with its row/column arguments mixed up), so nothing can pin
// it; it is restated expression by expression for the optional refinement stage.
constexpr int kMaxInterpSteps = 5;  // SIFT_MAX_INTERP_STEPS
constexpr int kImgBorder = 5;       // SIFT_IMG_BORDER

// matrix.hpp:12-33
void deriv_3d(const std::vector<Img>& dog, int Col, int intvl, int r, int c, float dI[3]) {
    dI[0] = (dog[intvl][r * Col + (c + 1)] - dog[intvl][r * Col + (c - 1)]) / 2.f;
    dI[1] = (dog[intvl][(r + 1) * Col + c] - dog[intvl][(r - 1) * Col + c]) / 2.f;
    dI[2] = (dog[intvl + 1][r * Col + c] - dog[intvl - 1][r * Col + c]) / 2.f;
}

// matrix.hpp:34-67
void hessian_3d(const std::vector<Img>& dog, int Col, int intvl, int r, int c, float H[9]) {
    const Img& d0 = dog[intvl - 1];
    const Img& d1 = dog[intvl];
    const Img& d2 = dog[intvl + 1];
    const float v = d1[r * Col + c];
    const float dxx = (d1[r * Col + (c + 1)] + d1[r * Col + (c - 1)] - 2.f * v);
    const float dyy = (d1[(r + 1) * Col + c] + d1[(r - 1) * Col + c] - 2.f * v);
    const float dss = (d2[r * Col + c] + d0[r * Col + c] - 2.f * v);
    const float dxy = (d1[(r + 1) * Col + (c + 1)] - d1[(r + 1) * Col + (c - 1)] - d1[(r - 1) * Col + (c + 1)] +
                       d1[(r - 1) * Col + (c - 1)]) / 4.f;
    const float dxs = (d2[r * Col + (c + 1)] - d2[r * Col + (c - 1)] - d0[r * Col + (c + 1)] +
                       d0[r * Col + (c - 1)]) / 4.f;
    const float dys = (d2[(r + 1) * Col + c] - d2[(r - 1) * Col + c] - d0[(r + 1) * Col + c] +
                       d0[(r - 1) * Col + c]) / 4.f;
    H[0] = dxx; H[1] = dxy; H[2] = dxs;
    H[3] = dxy; H[4] = dyy; H[5] = dys;
    H[6] = dxs; H[7] = dys; H[8] = dss;
}

// matrix.hpp:68-86
void invert_3x3(const float v[9], float m[9]) {
    float det_v = v[0] * (v[4] * v[8] - v[5] * v[7]);
    det_v -= v[3] * (v[1] * v[8] - v[2] * v[7]);
    det_v += v[6] * (v[1] * v[5] - v[2] * v[4]);
    m[0] = 1 * (v[4] * v[8] - v[5] * v[7]) / det_v;
    m[1] = -1 * (v[1] * v[8] - v[2] * v[7]) / det_v;
    m[2] = 1 * (v[1] * v[5] - v[2] * v[4]) / det_v;
    m[3] = -1 * (v[3] * v[8] - v[5] * v[6]) / det_v;
    m[4] = 1 * (v[0] * v[8] - v[2] * v[6]) / det_v;
    m[5] = -1 * (v[0] * v[5] - v[2] * v[3]) / det_v;
    m[6] = 1 * (v[3] * v[7] - v[4] * v[6]) / det_v;
    m[7] = -1 * (v[0] * v[7] - v[1] * v[6]) / det_v;
    m[8] = 1 * (v[0] * v[4] - v[1] * v[3]) / det_v;
}

// matrix.hpp:102-112 (GEMM_3X1 with alpha = -1, :87-96)
void interp_step(const std::vector<Img>& dog, int Col, int intvl, int r, int c, float& xi, float& xr, float& xc) {
    float dD[3], H[9], Hi[9];
    deriv_3d(dog, Col, intvl, r, c, dD);
    hessian_3d(dog, Col, intvl, r, c, H);
    invert_3x3(H, Hi);
    const float alpha = -1.0;
    xc = (Hi[0] * dD[0] + Hi[1] * dD[1] + Hi[2] * dD[2]) * alpha;
    xr = (Hi[3] * dD[0] + Hi[4] * dD[1] + Hi[5] * dD[2]) * alpha;
    xi = (Hi[6] * dD[0] + Hi[7] * dD[1] + Hi[8] * dD[2]) * alpha;
}

// matrix.hpp:113-124
float interp_contr(const std::vector<Img>& dog, int Col, int intvl, int r, int c, float xi, float xr, float xc) {
    float dD[3];
    deriv_3d(dog, Col, intvl, r, c, dD);
    const float t = dD[0] * xc + dD[1] * xr + dD[2] * xi;
    return dog[intvl][r * Col + c] + t * 0.5f;
}

// matrix.hpp:140-196.  Returns false where the reference returns nullptr.  One
// defined behaviour is added: a step whose offsets are not finite or exceed 1e9
// (singular Hessian) rejects the point; the reference converts such a value to
// int, which is undefined.
bool interp_extremum(const std::vector<Img>& dog, int COL, int ROW, int& intvl, int& r, int& c, int intvl_Len,
                     double contr_thr, float sub[3]) {
    float xi = 0.0, xr = 0.0, xc = 0.0;
    int i = 0;
    while (i < kMaxInterpSteps) {
        interp_step(dog, COL, intvl, r, c, xi, xr, xc);
        if (!(std::fabs(xi) < 1e9f && std::fabs(xr) < 1e9f && std::fabs(xc) < 1e9f)) return false;
        if (std::fabs(xi) < 0.5 && std::fabs(xr) < 0.5 && std::fabs(xc) < 0.5) break;
        c += (int)std::round(xc);
        r += (int)std::round(xr);
        intvl += (int)std::round(xi);
        if (intvl < 1 || intvl > intvl_Len || c < kImgBorder || c >= COL - kImgBorder || r < kImgBorder ||
            r >= ROW - kImgBorder)
            return false;
        i++;
    }
    if (i >= kMaxInterpSteps) return false;
    const float contr = interp_contr(dog, COL, intvl, r, c, xi, xr, xc);
    if (std::fabs(contr) < (float)(contr_thr)) return false;
    sub[0] = xc;
    sub[1] = xr;
    sub[2] = xi;
    return true;
}

struct Octave {
    int W = 0, H = 0;
    std::vector<Img> gau, dog;
};

// One iteration of the octave loop of Sift::pyramid (Sift/Sift.cpp:145-168):
// resample from the ORIGINAL image, 6 cascaded blurs with absolute sigmas, DoG.
// quality != 0: the ladder the reference's notes call for (update.md:10 doubts the sigmas of
// Sift.cpp:152-157): level i holds sigma0 * sqrt(2)^i in octave pixels at EVERY octave, reached
// from the level below by the increment sqrt(s_i^2 - s_below^2); the resampled image counts as
// blurred by half a source pixel.
void build_octave(Octave& oc, const Img& gray, size_t w, size_t h, size_t ph, size_t pyWidth, int quality = 0) {
    float curr_size = 2.f, coef = 1.f;
    for (size_t k = 0; k < ph; k++) { curr_size /= 2.f; coef *= 2.f; }
    Img cur;
    first(cur, gray, w, h, curr_size, &oc.W, &oc.H);
    const size_t W = (size_t)(w * curr_size), H = (size_t)(h * curr_size);
    oc.W = (int)W;
    oc.H = (int)H;
    oc.gau.assign(pyWidth, Img());
    if (quality) {
        float below = std::max(0.5f * curr_size, 0.1f);
        for (size_t i = 0; i < pyWidth; i++) {
            const float s = kGauSigma * powf(std::sqrt(2.0), i);
            const float s_blur = std::sqrt(std::max(s * s - below * below, 0.01f));
            gau_blur(oc.gau[i], i == 0 ? cur : oc.gau[i - 1], W, H, s_blur);
            below = s;
        }
    } else {
    gau_blur(oc.gau[0], cur, W, H, kGauSigma * coef);
    for (size_t i = 1; i < pyWidth; i++) {
        const float curr_Sigma = kGauSigma * powf(std::sqrt(2.0), i) * coef;
        gau_blur(oc.gau[i], oc.gau[i - 1], W, H, curr_Sigma);
    }
    }
    oc.dog.assign(pyWidth - 1, Img());
    for (size_t i = 0; i < pyWidth - 1; i++) {
        oc.dog[i].resize(W * H);
        for (size_t idx = 0; idx < W * H; idx++) oc.dog[i][idx] = oc.gau[i][idx] - oc.gau[i + 1][idx];
    }
}

// Sift/Sift.cpp:169-202 for one octave: scan, orientation, descriptors.
// subpixel != 0 inserts the refinement where the reference has it commented out
// (Sift.cpp:186-191), called as its signature intends (COL = width, r = j, c = i):
// a rejected point is dropped, an accepted one is moved to the refined (layer,
// row, column) and keeps the offsets; the orientation window must still fit the
// image at the new place (the scan bounds of the new layer), else it is dropped.
// quality != 0: windows of radius 3 x (1.5 sigma) weighted by 1.5 sigma with sigma the layer's
// own in octave pixels, refinement always on with the contrast bar per interval
// (contr_thr / intvls, the rule matrix.hpp:181 keeps commented out), interpolated orientation,
// radian descriptor frame, normalised descriptor.
void detect_octave(std::vector<Feat>& feats, const Octave& oc, size_t ph, size_t pyWidth, int describe,
                   int subpixel = 0, double contr_thr = 0.05, int quality = 0) {
    float curr_size = 2.f, coef = 1.f;
    for (size_t k = 0; k < ph; k++) { curr_size /= 2.f; coef *= 2.f; }
    const size_t W = oc.W, H = oc.H;
    const size_t firstNew = feats.size();
    const int intvls = (int)pyWidth - 3;
    if (quality) {
        subpixel = 1;
        contr_thr = (double)((float)contr_thr / (float)intvls);
    }
    auto layer_sigma = [&](size_t li) -> float {
        return quality ? kGauSigma * powf(std::sqrt(2.0), li) : kGauSigma * powf(std::sqrt(2.0), li) * coef;
    };
    auto layer_radius = [&](size_t li) -> size_t {
        if (quality) return (size_t)std::lround(3.0 * (1.5f * layer_sigma(li)));
        return (size_t)(curr_size * 3. * 1.5 * (li + 1));
    };
    for (size_t intvl_idx = 1; intvl_idx < pyWidth - 1 - 1; intvl_idx++) {
        const size_t r = layer_radius(intvl_idx);
        // The reference's unsigned bounds wrap when the octave is smaller than
        // the window; such octaves contribute nothing here.
        if (H < 2 * r + 3 || W < 2 * r + 3) continue;
        for (size_t j = r + 1; j < H - r - 1; j++) {
            for (size_t i = r + 1; i < W - r - 1; i++) {
                if (find_max_min(oc.dog, intvl_idx, W, j, i)) {
                    const Img& cur = oc.dog[intvl_idx];
                    const float currSigma = layer_sigma(intvl_idx);
                    if (harris_corner(cur, W, j, i, r)) {
                        if (!subpixel) {
                            histogram_ms(feats, cur, W, curr_size, intvl_idx, currSigma, j, i, r);
                            continue;
                        }
                        int li = (int)intvl_idx, rr = (int)j, cc = (int)i;
                        float sub[3];
                        if (!interp_extremum(oc.dog, (int)W, (int)H, li, rr, cc, (int)pyWidth - 3, contr_thr, sub))
                            continue;
                        const size_t r2 = layer_radius((size_t)li);
                        if (H < 2 * r2 + 3 || W < 2 * r2 + 3 || (size_t)rr < r2 + 1 || (size_t)rr >= H - r2 - 1 ||
                            (size_t)cc < r2 + 1 || (size_t)cc >= W - r2 - 1)
                            continue;
                        const float sigma2 = layer_sigma((size_t)li);
                        histogram_ms(feats, oc.dog[li], W, curr_size, li, sigma2, rr, cc, r2, sub, quality, 1.5f * sigma2);
                    }
                }
            }
        }
    }
    if (!describe) return;
    // Sift/Sift.cpp:450-480 FeatureDescrip
    for (size_t k = firstNew; k < feats.size(); k++) {
        Feat& f = feats[k];
        const Img& img = oc.gau[f.kai];
        if (quality) {
            descr_hist_quality(f.hist, img, (int)W, (int)H, f.y, f.x, f.sita, f.sigmaOCT, kDescrHistRadius, kDescrHistBins);
            hist_to_descr_quality(f);
            continue;
        }
        int ori = f.sita;
        int scl = (int)f.sigmaOCT;
        descr_hist(f.hist, img, (int)W, (int)H, f.y, f.x, (float)ori, (float)scl, kDescrHistRadius,
                   kDescrHistBins);
        hist_to_descr(f);
    }
}

}  // namespace

extern "C" {

struct orc_feat {  // mirrors include/sift_hip.h: sift_hip_feat (+ raw histogram)
    float size;
    int32_t kai;
    float sigmaOCT;
    int32_t x, y;
    float mm;
    int32_t sita;
    uint8_t descr[128];
};

int orc_gau_matrix(float p, int mat_len, float* out, int cap) {
    std::vector<float> g = gau_matrix(p, (size_t)mat_len);
    if ((int)g.size() > cap) return -(int)g.size();
    std::memcpy(out, g.data(), g.size() * sizeof(float));
    return (int)g.size();
}

void orc_gau_blur(const float* in, float* out, int w, int h, float p) {
    Img src(in, in + (size_t)w * h), dst;
    gau_blur(dst, src, w, h, p);
    std::memcpy(out, dst.data(), dst.size() * sizeof(float));
}

void orc_first_size(int w, int h, float ratio, int* ow, int* oh) {
    *oh = static_cast<int>(std::floor((size_t)h * ratio));
    *ow = static_cast<int>(std::floor((size_t)w * ratio));
}

void orc_first(const float* in, int w, int h, float ratio, float* out) {
    Img src(in, in + (size_t)w * h), dst;
    int ow, oh;
    first(dst, src, w, h, ratio, &ow, &oh);
    std::memcpy(out, dst.data(), dst.size() * sizeof(float));
}

float orc_fast_atan2f(float dy, float dx) { return fast_atan2f(dy, dx); }

int orc_pyramid_octaves(int w, int h) { return (int)pyramid_octaves(w, h); }

// Sift/Imgraw.cpp:257-269  ConverGray for interleaved RGB float input
void orc_convert_gray(const float* rgb, int n, float* gray) {
    for (int i = 0; i < n; i++) {
        const float R = rgb[i * 3 + 0], G = rgb[i * 3 + 1], B = rgb[i * 3 + 2];
        gray[i] = (float)(R * 0.299 + G * 0.587 + B * 0.114);
    }
}

// Gaussian and DoG levels of one octave, for kernel-level parity tests.
// gau: pyWidth*W*H floats, dog: (pyWidth-1)*W*H floats (either may be null).
void orc_octave_levels(const float* gray, int w, int h, int octave, int intvls, float* gau, float* dog,
                       int* ow, int* oh) {
    Img g(gray, gray + (size_t)w * h);
    Octave oc;
    const size_t pyWidth = intvls + 3;
    build_octave(oc, g, w, h, octave, pyWidth);
    *ow = oc.W;
    *oh = oc.H;
    const size_t P = (size_t)oc.W * oc.H;
    if (gau) for (size_t i = 0; i < pyWidth; i++) std::memcpy(gau + i * P, oc.gau[i].data(), P * 4);
    if (dog) for (size_t i = 0; i + 1 < pyWidth; i++) std::memcpy(dog + i * P, oc.dog[i].data(), P * 4);
}

// Sift::pyramid (Sift/Sift.cpp:140-225).  octaves==0 -> reference's own count.
// hist_out (optional): n x 128 clamped raw histograms.  Returns feature count
// (the full count even when it exceeds cap; only cap entries are written).
// sub_out (optional): n x 3 sub-pixel offsets (xc, xr, xi).
static int extract_impl(const float* gray, int w, int h, int octaves, int intvls, orc_feat* out, int cap,
                        float* hist_out, int describe, int subpixel, double contr_thr, float* sub_out, int quality) {
    Img g(gray, gray + (size_t)w * h);
    const size_t pyWidth = intvls + 3;
    size_t octvs = octaves > 0 ? (size_t)octaves : pyramid_octaves(w, h);
    std::vector<Feat> feats;
    for (size_t ph = 0; ph < octvs; ph++) {
        Octave oc;
        build_octave(oc, g, w, h, ph, pyWidth, quality);
        if (oc.W < 2 || oc.H < 2) break;
        detect_octave(feats, oc, ph, pyWidth, describe, subpixel, contr_thr, quality);
    }
    const int n = (int)feats.size();
    for (int k = 0; k < n && k < cap; k++) {
        const Feat& f = feats[k];
        orc_feat& o = out[k];
        o.size = f.size; o.kai = f.kai; o.sigmaOCT = f.sigmaOCT; o.x = f.x; o.y = f.y;
        o.mm = f.mm; o.sita = f.sita;
        for (int i = 0; i < 128; i++) o.descr[i] = (uint8_t)f.descr[i];
        if (hist_out) std::memcpy(hist_out + (size_t)k * 128, f.hist, 128 * 4);
        if (sub_out) std::memcpy(sub_out + (size_t)k * 3, f.sub, 3 * 4);
    }
    return n;
}

int orc_extract_ex(const float* gray, int w, int h, int octaves, int intvls, orc_feat* out, int cap,
                   float* hist_out, int describe, int subpixel, double contr_thr, float* sub_out) {
    return extract_impl(gray, w, h, octaves, intvls, out, cap, hist_out, describe, subpixel, contr_thr, sub_out, 0);
}

// The quality mode (see build_octave / detect_octave / histogram_ms); sigmaOCT comes back
// un-truncated, hist_out holds the normalised, clamped, re-normalised histograms.
int orc_extract_quality(const float* gray, int w, int h, int octaves, int intvls, orc_feat* out, int cap,
                        float* hist_out, double contr_thr, float* sub_out) {
    return extract_impl(gray, w, h, octaves, intvls, out, cap, hist_out, 1, 1, contr_thr, sub_out, 1);
}

int orc_extract(const float* gray, int w, int h, int octaves, int intvls, orc_feat* out, int cap,
                float* hist_out, int describe) {
    return orc_extract_ex(gray, w, h, octaves, intvls, out, cap, hist_out, describe, 0, 0.05, nullptr);
}

// Sift/Sift.cpp:537-567  forceMatchFeat on float descriptors (d == 128):
// exact NN + second NN, strict '<' updates, accept iff (d1/d2) < th.
// best[q] = ref index or -1; d1/d2 = squared distances (FLT_MAX when unset).
void orc_force_match(const float* q, int nq, const float* r, int nr, float th, int32_t* best,
                     float* d1o, float* d2o) {
    for (int a = 0; a < nq; a++) {
        const float* fa = q + (size_t)a * 128;
        int m = -1;
        float dist1, dist2;
        dist1 = dist2 = std::numeric_limits<float>::max();
        for (int b = 0; b < nr; b++) {
            const float* fb = r + (size_t)b * 128;
            float value = 0;
            for (int i = 0; i < 128; i++) {
                const float reduce = fa[i] - fb[i];
                value += reduce * reduce;
            }
            if (value < dist1) { dist2 = dist1; dist1 = value; m = b; }
            else if (value < dist2) { dist2 = value; }
        }
        best[a] = ((dist1 / dist2) < th) ? m : -1;
        if (d1o) d1o[a] = dist1;
        if (d2o) d2o[a] = dist2;
    }
}

// Exact NN over uchar descriptors with the live call site's acceptance test
// (Sift/Sift.cpp:574-578: d0 < d1 * matchTh in float; fewer than two
// references -> no match, as k != 2 there).  ratio_form 1 selects
// forceMatchFeat's (d1/d2) < th instead.  Distances are exact integers
// (Sift.cpp:522-536 accumulates integer-valued floats below 2^24).
void orc_match_u8(const uint8_t* q, int nq, const uint8_t* r, int nr, float th, int ratio_form,
                  int32_t* best, int32_t* d1o, int32_t* d2o) {
    for (int a = 0; a < nq; a++) {
        const uint8_t* fa = q + (size_t)a * 128;
        int m = -1;
        int32_t dist1 = INT32_MAX, dist2 = INT32_MAX;
        for (int b = 0; b < nr; b++) {
            const uint8_t* fb = r + (size_t)b * 128;
            int32_t value = 0;
            for (int i = 0; i < 128; i++) {
                const int32_t reduce = (int32_t)fa[i] - (int32_t)fb[i];
                value += reduce * reduce;
            }
            if (value < dist1) { dist2 = dist1; dist1 = value; m = b; }
            else if (value < dist2) { dist2 = value; }
        }
        const float f1 = dist1 == INT32_MAX ? FLT_MAX : (float)dist1;
        const float f2 = dist2 == INT32_MAX ? FLT_MAX : (float)dist2;
        bool ok;
        if (ratio_form == 1) ok = (f1 / f2) < th;
        else ok = (nr >= 2) && (f1 < f2 * th);
        best[a] = (ok && m >= 0) ? m : -1;
        if (d1o) d1o[a] = dist1;
        if (d2o) d2o[a] = dist2;
    }
}

}  // extern "C"
