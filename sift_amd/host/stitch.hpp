// Drop-in for the two warps of the reference's Sift/stitch/stitch.hpp:6-9 that its
// blender calls (Blend.cpp:922, :974); both run on the GPU through include/sift_hip.h.
#pragma once
#include <vector>

#include "imagedata.hpp"

// dst pixel (column j, row i) = bilinear sample of src at H^-1 (j, i) (stitch.cpp:106-141);
// pixels that map outside src keep their value.  H: row-major 3x3 as ransac_xform returns it.
// Throws std::runtime_error if the GPU path fails or H does not hold 9 values.
void _WarpPerspective(const Raw& src, Raw& dst, const std::vector<float>& H);

// Cylindrical projection of every input image with focal length FL2 (stitch.cpp:178-221):
// Output[k] has the size of inputArrays[k]; upedge / downedge receive, appended image after
// image, the source positions of each image's first and last output row.
void warping(const std::vector<Raw>& inputArrays, float FL2, std::vector<Raw>& Output, std::vector<fpoint>& upedge,
             std::vector<fpoint>& downedge);
