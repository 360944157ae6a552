// Drop-in for the reference's Sift/stitch/Blend.hpp:11-15: the compositing half of the
// stitcher.  The pixel work (cylindrical projection, multi-band blend) runs on the GPU through
// include/sift_hip.h; the few hundred scalar operations around it (focal length, alignment
// offsets, canvas assembly) stay on the host, as byte copies and float arithmetic in the
// reference's operand types.
#pragma once
#include <cstddef>
#include <vector>

#include "Imgraw.hpp"
#include "imagedata.hpp"

// multiBandBlend (Blend.cpp:638-740): both projected images are updated in place inside their
// dx x |dy| overlap.  Throws std::runtime_error if the GPU path fails.
void multiBandBlend(Raw& limg, Raw& rimg, int dx, int dy);

// getFocal (Blend.cpp:867-894): the common focal length from the homography (row-major 3x3; may
// be empty), falling back to the mean of the two image sizes.
float getFocal(const std::vector<float>& HomogMat, size_t img1Size, size_t img2Size);

// blen2img (Blend.cpp:898-1174): img1 and img2 (24-bit) are projected onto a cylinder of the common
// focal length, aligned by the mean offset of the consensus matches, blended and placed side by
// side on a canvas, which dst receives (24-bit, values in [0, 1]).  RANSAC_feat: RANSAC_num >= 2
// consensus features of image 2 whose fwd_match point into image 1.  The perspective-warped
// preview the reference also builds there (Blend.cpp:918-965) feeds nothing and is not built.
void blen2img(const ImgRaw& img1, const ImgRaw& img2, ImgRaw& dst, const std::vector<float>& HomogMat,
              const Feature* const* RANSAC_feat, int RANSAC_num);
