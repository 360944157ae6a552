// The stitcher's two warps on the GPU (reference: Sift/stitch/stitch.cpp).
#include "stitch.hpp"

#include <stdexcept>
#include <string>

#include "../../include/sift_hip.h"
#include "Sift.hpp"

namespace {
void check(sift_hip_ctx* ctx, int rc, const char* what) {
    if (rc == SIFT_HIP_OK) return;
    std::string msg = std::string(what) + ": " + sift_hip_strerror(rc);
    const char* detail = sift_hip_last_error(ctx);
    if (detail && *detail) msg += std::string(" (") + detail + ")";
    throw std::runtime_error(msg);
}
}  // namespace

void _WarpPerspective(const Raw& src, Raw& dst, const std::vector<float>& H) {
    if (H.size() < 9) throw std::runtime_error("_WarpPerspective: homography needs 9 values");
    if (dst.RGB.empty() || src.RGB.empty()) return;  // the reference's loops do not run
    sift_amd::ContextUse use;
    sift_hip_ctx* ctx = use.ctx;
    check(ctx, sift_hip_warp_perspective(ctx, src.RGB.data(), src.getCol(), src.getRow(), H.data(), dst.RGB.data(),
                                         dst.getCol(), dst.getRow()),
          "sift_hip_warp_perspective");
}

void warping(const std::vector<Raw>& inputArrays, float FL2, std::vector<Raw>& Output, std::vector<fpoint>& upedge,
             std::vector<fpoint>& downedge) {
    Output.resize(inputArrays.size());
    sift_amd::ContextUse use;
    sift_hip_ctx* ctx = use.ctx;
    for (size_t idx = 0; idx < inputArrays.size(); idx++) {
        const Raw& image = inputArrays[idx];
        Output[idx] = Raw((size_t)image.getCol(), (size_t)image.getRow());
        if (image.RGB.empty()) continue;
        const size_t w = (size_t)image.getCol();
        std::vector<float> up(2 * w), down(2 * w);
        check(ctx, sift_hip_warp_cylinder(ctx, image.RGB.data(), image.getCol(), image.getRow(), FL2,
                                          Output[idx].RGB.data(), up.data(), down.data()),
              "sift_hip_warp_cylinder");
        for (size_t i = 0; i < w; i++) upedge.emplace_back(up[2 * i], up[2 * i + 1]);
        if (image.getRow() > 1)  // a one-row image has no separate last row (stitch.cpp:212-216)
            for (size_t i = 0; i < w; i++) downedge.emplace_back(down[2 * i], down[2 * i + 1]);
    }
}
