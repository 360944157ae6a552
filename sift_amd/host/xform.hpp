// Drop-in for the reference's Sift/ransac/xform.hpp:4-6: the RANSAC homography
// between matched features, with no OpenCV underneath.
#pragma once
#include <vector>

struct Feature;

// Homography (row-major 3x3, H[8] = 1) taking each matched feature's position
// (rX(), rY()) to its fwd_match's, estimated by RANSAC over samples of m pairs
// (Sift/ransac/xform.cpp:37-168).  `features[0..n)`: only those with a non-null
// fwd_match take part.  On success *inliers (when not null) receives a new[]
// array of the consensus features, owned by the caller, and n_in their count --
// written only if n_in was 0 on entry, as in the reference.  Returns an empty
// vector when there are fewer than m matches or the best consensus is smaller
// than the minimum the error bound p_badxform asks for.
// The random stream is libc's rand(); its seed is sift_amd::ransac_seed().
extern std::vector<float> ransac_xform(Feature* features, int n, int m, float p_badxform, float err_tol,
                                       Feature*** inliers, int& n_in);

namespace sift_amd {
// Seed given to srand() at the start of every ransac_xform call.  0 (default)
// means the wall clock, which is what the reference does (xform.cpp:42).
unsigned& ransac_seed();
// Smallest consensus size for which the chance that a wrong model gathers it is
// below p_badxform, among n matches and samples of m (xform.cpp:261-281; the
// reference's formula, evaluated in double through lgamma).
int ransac_min_inliers(int n, int m, float p_badsupp, float p_badxform);
}  // namespace sift_amd
