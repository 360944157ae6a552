/*
 * sift_hip.h -- C ABI of the MI355X (gfx950) SIFT detect / describe / match path.
 *
 * The reference (hunandy14/Sift) has no FFI of its own: its boundary is the C++
 * class API (Sift, ImgRaw, Feature, Stitching).  The host classes in
 * sift_amd/host/ keep that API and call the entry points below; each entry point
 * names the reference code it stands in for (paths relative to the reference's
 * Sift/ directory).  Plain C types only; the caller owns every host buffer, the
 * context owns every device buffer.  No entry point throws; all return
 * SIFT_HIP_OK (0) or a negative sift_hip_status.
 *
 * Threading: a context is bound to one device and is not thread-safe.  Use one
 * context per GPU; distinct contexts may be driven concurrently from different
 * threads (one exception: sift_hip_match_u8_noalloc, whose work area is one static
 * array per GPU shared by every context -- see its comment).  A context overlaps the octaves of a frame on several HIP streams of its own
 * and joins them on the host; no stream of the library ever waits on another
 * stream's event, so it cannot stall behind other users of the device's hardware
 * queues (further contexts, the caller's own streams, RCCL).
 */
#ifndef SIFT_HIP_H
#define SIFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sift_hip_ctx sift_hip_ctx;

typedef enum sift_hip_status {
    SIFT_HIP_OK = 0,
    SIFT_HIP_ERR_ARG = -1,      /* null pointer, non-positive size, bad enum */
    SIFT_HIP_ERR_NOSPC = -2,    /* output capacity too small; *n holds the needed count */
    SIFT_HIP_ERR_HIP = -3,      /* a HIP runtime call failed (see sift_hip_last_error) */
    SIFT_HIP_ERR_NOMEM = -4,    /* device allocation failed */
    SIFT_HIP_ERR_SAME = -5,     /* in == out (the reference throws file_same, imglib/imglib.cpp:76-78) */
    SIFT_HIP_ERR_NODEVICE = -6, /* no usable gfx950 device */
    SIFT_HIP_ERR_LIMIT = -7     /* an internal fixed limit was exceeded (candidate table) */
} sift_hip_status;

/* Compile-time tunables of Sift.hpp:44-77 that the path reads, as run-time fields.
 * Always start from sift_hip_default_params() and change fields: every field is validated
 * (subpixel, blur_mode and textbook must be 0 or 1), so a struct with an uninitialised
 * field is rejected with SIFT_HIP_ERR_ARG or, worse, selects a mode by accident. */
typedef struct sift_hip_params {
    int32_t octaves;     /* 0 = the reference's own count, Sift.cpp:130-139 (getPyramidOctv) */
    int32_t intvls;      /* SIFT_INTVLS (3); Gaussian levels per octave = intvls + 3 */
    float sigma0;        /* SIFT_GauSigma (1.6f) */
    float dx_thr;        /* SIFT_Dx (0.03f); the contrast test is |D| > dx_thr / 2 */
    int32_t subpixel;    /* 0 = off, as shipped by the reference (call commented out at Sift.cpp:187);
                            1 = every candidate goes through interp_extremum (matrix.hpp:140-196): dropped
                            where that returns nullptr, else moved to the refined (layer, row, column),
                            which must still hold that layer's orientation window (Sift.cpp:178-179);
                            the offsets are read with sift_hip_last_offsets */
    int32_t blur_mode;   /* 0 = exact: float product, double accumulation in tap order
                            (imglib/imglib.cpp:85-117); 1 = fast: fp32 FMA accumulation */
    float contr_thr;     /* SIFT_CONTR_THR (0.05), used only when subpixel = 1 */
    int32_t textbook;    /* 0 = the reference's pyramid and descriptor, quirks included (parity mode, default);
                            1 = quality mode, which departs from the reference where its own notes call the
                            code unverified (update.md:9-10): every Gaussian level is blurred from the one
                            below by the INCREMENT that brings it to sigma0 * sqrt(2)^i (the reference applies
                            the absolute sigma again at every level), with the same sigmas in octave pixels
                            at every octave (the reference doubles them per octave on top of halving the
                            image); orientation windows of radius 4.5 sigma weighted by 1.5 sigma; the
                            descriptor frame rotated by the orientation in RADIANS (the reference feeds
                            degrees to cos/sin, which breaks rotation invariance), sized by the un-truncated
                            sigma, and normalised, clamped at 0.2 and normalised again before quantisation
                            (both normalisations are no-ops in the reference); sub-pixel refinement on.
                            Results are not comparable with the reference's; see the accuracy test. */
} sift_hip_params;

/* One keypoint + descriptor: the fields of `struct Feature` (imagedata.hpp:63-86)
 * that Sift::pyramid() fills.  descr holds the integer-valued descriptor
 * (Sift.cpp:416-420: min(255, (int)(512 * v))) as bytes. */
typedef struct sift_hip_feat {
    float size;      /* octave scale: 2, 1, 0.5, ... */
    int32_t kai;     /* DoG layer 1..intvls */
    float sigmaOCT;  /* integer-truncated sigma (Sift.cpp:229,287) */
    int32_t x, y;    /* octave-pixel coordinates */
    float mm;        /* orientation-histogram bin value */
    int32_t sita;    /* orientation in degrees, multiple of 10 */
    uint8_t descr[128];
} sift_hip_feat;

/* ---- context -------------------------------------------------------------------- */
int sift_hip_create(sift_hip_ctx** ctx, int device);
int sift_hip_destroy(sift_hip_ctx* ctx);
void sift_hip_default_params(sift_hip_params* p);
const char* sift_hip_strerror(int status);
const char* sift_hip_last_error(const sift_hip_ctx* ctx); /* text of the last HIP failure */
/* HIP stream the context launches on (a hipStream_t), for callers that order
 * their own device work against it. */
void* sift_hip_stream(sift_hip_ctx* ctx);
int sift_hip_synchronize(sift_hip_ctx* ctx);
/* Page-locked host memory (hipHostMalloc) for buffers the entry points fill or read: results
 * land in it by direct DMA, without the runtime's staging copies (0.13 ms less per 4K frame for
 * the feature array).  Any host pointer works everywhere; this is only faster.  Free with
 * sift_hip_host_free before or after the context is destroyed. */
int sift_hip_host_alloc(sift_hip_ctx* ctx, void** ptr, size_t bytes);
int sift_hip_host_free(sift_hip_ctx* ctx, void* ptr);

/* ---- extract: replaces the body of Sift::pyramid(), Sift.cpp:140-225 -------------
 * img: host pointer, row-major top-down floats in [0,1]; channels = 1 (gray) or 3
 * (interleaved RGB, converted as ImgRaw::ConverGray, Imgraw.cpp:257-269).
 * out: caller-allocated array of cap features, filled in the reference's order
 * (octave, layer, row, column, orientation bin -- Sift.cpp:207-224).
 * hist: optional (may be NULL) cap x 128 floats: the clamped histogram the
 * descriptor bytes are quantised from.
 * *n receives the feature count; if it exceeds cap the call returns
 * SIFT_HIP_ERR_NOSPC, *n is the needed capacity and out holds the first cap. */
int sift_hip_extract(sift_hip_ctx* ctx, const float* img, int w, int h, int channels,
                     const sift_hip_params* params, sift_hip_feat* out, int cap, int* n, float* hist);
/* Same, with the image already in device memory (gray only); out/hist stay host.  The frame
 * must be complete, or be produced by work already enqueued on sift_hip_stream(ctx) (the call
 * then waits for that work on the host before it uses its other streams). */
int sift_hip_extract_dev(sift_hip_ctx* ctx, const float* d_gray, int w, int h,
                         const sift_hip_params* params, sift_hip_feat* out, int cap, int* n, float* hist);
/* nframes gray frames of equal size, contiguous on the host; frame f writes
 * out[f*cap .. f*cap + n[f]) (the per-frame loop of StitchImg, Sift_main.cpp:31-48).
 * Uploads overlap the previous frame's kernels. */
int sift_hip_extract_batch(sift_hip_ctx* ctx, const float* frames, int w, int h, int nframes,
                           const sift_hip_params* params, sift_hip_feat* out, int cap, int* n);
/* Device-resident variant of the batch call: frames in device memory. */
int sift_hip_extract_batch_dev(sift_hip_ctx* ctx, const float* d_frames, int w, int h, int nframes,
                               const sift_hip_params* params, sift_hip_feat* out, int cap, int* n);

/* Sub-pixel offsets (xc, xr, xi of interp_extremum, matrix.hpp:154-194) of the features
 * that the last extract call on this context returned, three floats per feature in
 * output order; the refined position is (x + xc, y + yr) in octave pixels, i.e.
 * ((x + xc) / size, (y + xr) / size) in the source image (matrix.hpp:188-189).  Returns
 * the number of features written (at most cap); 0 if that call ran with subpixel = 0.
 * After a batch call it covers the last frame only. */
int sift_hip_last_offsets(sift_hip_ctx* ctx, float* xyz, int cap);

/* ---- image ops kept public by the reference (imglib/imglib.hpp:18-57) ----------- */
/* Gaus::gau_matrix, imglib.cpp:127-155 (host arithmetic, no GPU work). Returns the
 * tap count, or -needed if cap is too small. */
int sift_hip_gau_matrix(float p, int mat_len, float* taps, int cap);
/* Gaus::GauBlur, imglib.cpp:72-118: host in -> host out, w*h floats; mat_len = 0 derives the
 * tap count from p as the reference does. */
int sift_hip_gau_blur(sift_hip_ctx* ctx, const float* in, float* out, int w, int h, float p, int mat_len,
                      int blur_mode);
/* Scaling::first / ImgRaw::first, imglib.cpp:227-252: out is floor(w*ratio) x floor(h*ratio). */
int sift_hip_first(sift_hip_ctx* ctx, const float* in, int w, int h, float ratio, float* out);
/* ImgRaw::ConverGray, Imgraw.cpp:257-269: n RGB pixels -> n gray floats. */
int sift_hip_convert_gray(sift_hip_ctx* ctx, const float* rgb, int n, float* gray);
/* Gaussian and DoG levels of one octave as Sift::pyramid builds them
 * (Sift.cpp:145-168); gau: (intvls+3)*ow*oh floats, dog: (intvls+2)*ow*oh floats,
 * either may be NULL.  ow/oh receive the octave size. */
int sift_hip_octave_levels(sift_hip_ctx* ctx, const float* gray, int w, int h, const sift_hip_params* params,
                           int octave, float* gau, float* dog, int* ow, int* oh);
/* getPyramidOctv, Sift.cpp:130-139. */
int sift_hip_pyramid_octaves(int w, int h);

/* ---- match: replaces kdtreeMatchFeat / forceMatchFeat, Sift.cpp:537-584 ----------
 * Exact nearest / second-nearest neighbour of every query among the references
 * over 128-byte descriptors (squared L2 in int32, exact), lowest reference index
 * winning ties (forceMatchFeat's strict '<').  ratio_form 0: accept iff
 * d1 < d2 * ratio in float and nr >= 2 (the live call site, Sift.cpp:573-578);
 * ratio_form 1: accept iff (d1 / d2) < ratio (forceMatchFeat, Sift.cpp:563).
 * best[i] = reference index or -1; d1/d2 (optional) = the two squared distances
 * (INT32_MAX where no such reference exists). */
int sift_hip_match_u8(sift_hip_ctx* ctx, const uint8_t* q, int nq, const uint8_t* r, int nr, float ratio,
                      int ratio_form, int32_t* best, int32_t* d1, int32_t* d2);
/* All pointers in device memory; runs on the context's stream, no host sync. */
int sift_hip_match_u8_dev(sift_hip_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, float ratio,
                          int ratio_form, int32_t* d_best, int32_t* d_d1, int32_t* d_d2);

/* The same as sift_hip_match_u8_dev for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in
 * the reference.  The call allocates nothing, on the device or through any library: its work
 * areas (int8 copies of both sets, norms) live in a static device array of the library
 * (about 17 MB per GPU, held until the process ends), which bounds the sizes: nq and nr at
 * most SIFT_HIP_NOALLOC_MAX each, beyond which the call returns SIFT_HIP_ERR_LIMIT and
 * launches nothing.  All calls on one GPU, from any context, share that array: calls that
 * overlap in time on one GPU race and must be serialised by the caller.  Results are equal
 * to sift_hip_match_u8_dev's bit for bit.  Runs on the context's stream, no host sync. */
#define SIFT_HIP_NOALLOC_MAX 65536
int sift_hip_match_u8_noalloc(sift_hip_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_r, int nr, float ratio,
                              int ratio_form, int32_t* d_best, int32_t* d_d1, int32_t* d_d2);

/* ---- descriptors that stay on the device between the extract and the match: the call site of the
 * matcher (Stitching::Check, Sift/Sift.cpp:669-676) matches two feature arrays that two extracts
 * produced a moment earlier; their descriptor bytes are still in device memory then.
 * sift_hip_keep_descriptors copies the descriptors of the features that the LAST extract call on this
 * context returned (any extract entry point; after a batch call: its last frame) into a device buffer
 * of their own, n x 128 bytes in output order, owned by the returned set; *n (optional) receives the
 * count.  SIFT_HIP_ERR_ARG if no extract has delivered features since the last one began.
 * sift_hip_match_kept is sift_hip_match_u8 on two such sets (both made on this context's device):
 * nothing is packed or uploaded; best / d1 / d2 are host arrays of the query set's count.  Results equal
 * sift_hip_match_u8 on the descr bytes of the two feature arrays.  A set is freed with
 * sift_hip_release_descriptors (before its context is destroyed, or with ctx = NULL afterwards). */
typedef struct sift_hip_descr_set sift_hip_descr_set;
int sift_hip_keep_descriptors(sift_hip_ctx* ctx, sift_hip_descr_set** set, int* n);
void sift_hip_release_descriptors(sift_hip_ctx* ctx, sift_hip_descr_set* set);
int sift_hip_descriptor_count(const sift_hip_descr_set* set);
int sift_hip_match_kept(sift_hip_ctx* ctx, const sift_hip_descr_set* q, const sift_hip_descr_set* r, float ratio,
                        int ratio_form, int32_t* best, int32_t* d1, int32_t* d2);

/* Multi-GPU form, one process per GPU: every rank passes its own shard of the references
 * (the same nr_local on every rank) and its own queries; the shards are all-gathered with
 * ncclAllGather on the caller's communicator (comm: an ncclComm_t of nranks ranks; RCCL over
 * xGMI) into a buffer of the context, on the context's stream, and the rank's queries are
 * matched against all nranks * nr_local references.  best holds indices into the gathered
 * order (rank-major: rank k's reference i is k * nr_local + i).  No counterpart in the
 * reference, which is single-process; the comparison semantics are those of
 * sift_hip_match_u8.  No host synchronisation. */
int sift_hip_match_u8_sharded(sift_hip_ctx* ctx, void* comm, int nranks, const uint8_t* d_q, int nq,
                              const uint8_t* d_r_local, int nr_local, float ratio, int ratio_form, int32_t* d_best,
                              int32_t* d_d1, int32_t* d_d2);

/* The same when the ranks hold different numbers of references (real frames give unequal
 * feature counts).  nr_of_rank: host array of nranks counts, identical on every rank (e.g. the
 * gathered per-frame feature counts); this rank contributes nr_of_rank[rank] rows of d_r_local.
 * The shards are padded to the largest count for one ncclAllGather and compacted on the device,
 * so best indexes the rank-ordered concatenation WITHOUT padding: rank k's reference i is
 * (sum of nr_of_rank[0..k)) + i, as sift_hip_ragged_offsets computes.  No host synchronisation. */
int sift_hip_match_u8_sharded_ragged(sift_hip_ctx* ctx, void* comm, int nranks, int rank, const uint8_t* d_q, int nq,
                                     const uint8_t* d_r_local, const int32_t* nr_of_rank, float ratio, int ratio_form,
                                     int32_t* d_best, int32_t* d_d1, int32_t* d_d2);
/* Index arithmetic of the call above (host only, no GPU work): offsets[k] = first index of rank
 * k's references in the concatenation, offsets[nranks] = their total; returns the largest count,
 * or a negative status (a negative count, or a total beyond INT32_MAX). */
int sift_hip_ragged_offsets(const int32_t* nr_of_rank, int nranks, int64_t* offsets);

/* ---- RANSAC homography over the matches: replaces ransac_xform, ---------------------
 * Sift/ransac/xform.cpp:37-168 (its call site: Stitching::Check, Sift.cpp:684) --------
 * q_feats[0..nq): image-2 features (the matcher's queries), r_feats[0..nr): image-1
 * features, best[i]: the matcher's output for query i (index into r_feats, or -1).
 * Estimates the homography taking a matched query's source-image position
 * (x / size, y / size) to its match's, from samples of m pairs (4 <= m <= 8):
 * every sample is fitted (lsq_homog, xform.cpp:177-219, here a QR solve in double
 * instead of OpenCV's SVD) and every model's consensus counted
 * (find_consensus, xform.cpp:365-395) on the GPU, in batches; the stopping rule,
 * the minimum consensus (calc_min_inliers, xform.cpp:261-281) and the two final
 * refits follow the reference.  Samples come from libc's rand() seeded with `seed`
 * (0: the wall clock, as the reference does at xform.cpp:42).
 * Outputs (host memory): H9 = row-major 3x3 with H9[8] = 1; inlier_idx (capacity
 * nq) = query indices of the final consensus in ascending order; *n_in = its size,
 * or -1 when there are fewer than m matches or no model reached the minimum
 * consensus (H9 and inlier_idx are then left untouched); *iterations (optional) =
 * samples the loop consumed.  Blocks until done. */
int sift_hip_ransac_homog(sift_hip_ctx* ctx, const sift_hip_feat* q_feats, int nq, const sift_hip_feat* r_feats,
                          int nr, const int32_t* best, int m, float p_badxform, float err_tol, unsigned seed,
                          float* H9, int32_t* inlier_idx, int32_t* n_in, int32_t* iterations);
/* The same with q_feats, r_feats and best in device memory (e.g. the buffers
 * sift_hip_match_u8_dev wrote); the outputs stay host pointers. */
int sift_hip_ransac_homog_dev(sift_hip_ctx* ctx, const sift_hip_feat* d_q_feats, int nq,
                              const sift_hip_feat* d_r_feats, int nr, const int32_t* d_best, int m, float p_badxform,
                              float err_tol, unsigned seed, float* H9, int32_t* inlier_idx, int32_t* n_in,
                              int32_t* iterations);

/* The same over pairs the caller has formed: xyuv = n x (x, y, u, v) floats in host memory,
 * (x, y) a matched feature's position and (u, v) its match's, in the order ransac_xform
 * would meet them (ascending feature index).  inlier_idx (capacity n) = pair indices. */
int sift_hip_ransac_homog_pairs(sift_hip_ctx* ctx, const float* xyuv, int n, int m, float p_badxform, float err_tol,
                                unsigned seed, float* H9, int32_t* inlier_idx, int32_t* n_in, int32_t* iterations);
/* All three forms draw from libc's process-wide rand(), as the reference does: calls that
 * overlap in time in several threads interleave their sample streams. */

/* ---- geometric warps of the stitcher: replace _WarpPerspective and warping, ----------
 * Sift/stitch/stitch.cpp:106-141 and :178-221 (call sites: blen2img, Blend.cpp:922, :974).
 * Images are interleaved 8-bit RGB, row-major, cols x rows.
 * warp_perspective: dst pixel (column j, row i) samples src bilinearly (stitch.cpp:19-63)
 * at the position the closed-form inverse of H9 (row-major 3x3, as ransac_xform returns it)
 * gives; dst is in/out: pixels whose position falls outside src keep their value, samples
 * whose 2x2 neighbourhood leaves src are black, as in the reference. */
int sift_hip_warp_perspective(sift_hip_ctx* ctx, const uint8_t* src_rgb, int s_col, int s_row, const float* H9,
                              uint8_t* dst_rgb, int d_col, int d_row);
/* src and dst in device memory (H9 stays a host pointer); runs on the context's stream, no host sync. */
int sift_hip_warp_perspective_dev(sift_hip_ctx* ctx, const uint8_t* d_src_rgb, int s_col, int s_row, const float* H9,
                                  uint8_t* d_dst_rgb, int d_col, int d_row);
/* warp_cylinder: the cylindrical projection with focal length `focal` (> 0) of one w x h image
 * into out_rgb (w x h, every pixel written, black where the source position leaves the image);
 * upedge / downedge (optional, 2 * w floats each) receive the (x, y) source positions of the
 * first and the last output row. */
int sift_hip_warp_cylinder(sift_hip_ctx* ctx, const uint8_t* src_rgb, int w, int h, float focal, uint8_t* out_rgb,
                           float* upedge, float* downedge);
int sift_hip_warp_cylinder_dev(sift_hip_ctx* ctx, const uint8_t* d_src_rgb, int w, int h, float focal,
                               uint8_t* d_out_rgb, float* d_upedge, float* d_downedge);

/* ---- multi-band blend of the stitcher: replaces multiBandBlend, Sift/stitch/Blend.cpp:638-740
 * (call site: blen2img, Blend.cpp:999-1006).  l_rgb (lw x lh) and r_rgb (rw x rh) are the two
 * cylindrically projected images, interleaved 8-bit RGB; the right image lies dx columns into
 * the left one's right edge and dy rows against its bottom edge (alignMatch's offsets; dy < 0:
 * against its top edge).  The dx x |dy| overlap of both is cut out, its black border extended,
 * decomposed into five-level Laplacian pyramids (5-tap Gaussian, the reference's bicubic
 * resampler), blended level by level, collapsed and written back into BOTH images where both
 * were non-black.  1 <= dx <= 16384, 1 <= |dy| <= 16384. */
int sift_hip_multiband_blend(sift_hip_ctx* ctx, uint8_t* l_rgb, int lw, int lh, uint8_t* r_rgb, int rw, int rh, int dx,
                             int dy);
/* Both images in device memory; runs on the context's stream (one short host wait for the upload
 * of its constants). */
int sift_hip_multiband_blend_dev(sift_hip_ctx* ctx, uint8_t* d_l_rgb, int lw, int lh, uint8_t* d_r_rgb, int rw, int rh,
                                 int dx, int dy);

/* ---- the pixel work of blen2img in one call: replaces the sequence warping (Sift/stitch/Blend.cpp:974),
 * multiBandBlend (:994-996) and the canvas assembly (:1010-1031) for one pair of images.  Both
 * images (interleaved 8-bit RGB, host memory) are projected onto the cylinder of focal length
 * `focal`, the dx x |blend_dy| overlap is blended (the arguments multiBandBlend gets), and the
 * cols x rows canvas (host memory, every pixel written) receives the first projection top0 rows
 * down from its top-left corner and the non-black pixels of the second at column left1, row top1;
 * what falls outside the canvas is dropped.  Only the two images cross to the device and only
 * the canvas comes back.  Results equal the three separate calls followed by the host loop. */
int sift_hip_panorama_pair(sift_hip_ctx* ctx, const uint8_t* img1_rgb, int w1, int h1, const uint8_t* img2_rgb, int w2,
                           int h2, float focal, int dx, int blend_dy, int top0, int left1, int top1, uint8_t* canvas_rgb,
                           int cols, int rows);
/* The same with the two images as ImgRaw holds them (Sift/Imgraw.hpp:30-41): interleaved float RGB in host
 * memory, samples in [0,1] when `normalised` is non-zero and in [0,255] otherwise.  The 8-bit copies blen2img
 * makes first (Blend.cpp:907-916 through Imgraw.hpp:59-82: (unsigned char)(v * 255.0), the product in double,
 * truncated) are made on the device after the upload, so the host never walks the pixels.  Results equal
 * sift_hip_panorama_pair on the 8-bit copies. */
int sift_hip_panorama_pair_f32(sift_hip_ctx* ctx, const float* img1_rgb, int w1, int h1, const float* img2_rgb, int w2,
                               int h2, int normalised, float focal, int dx, int blend_dy, int top0, int left1, int top1,
                               uint8_t* canvas_rgb, int cols, int rows);

/* ---- measurement ------------------------------------------------------------------
 * With profiling on, every kernel launch is bracketed by HIP events on the
 * context's stream; totals are kept per kernel class. */
enum {
    SIFT_HIP_K_RESAMPLE = 0,
    SIFT_HIP_K_BLUR_ROW = 1,
    SIFT_HIP_K_BLUR_COL = 2,
    SIFT_HIP_K_DETECT = 3,
    SIFT_HIP_K_COMPACT = 4,
    SIFT_HIP_K_ORIENT = 5,
    SIFT_HIP_K_EMIT = 6,
    SIFT_HIP_K_DESCR = 7,
    SIFT_HIP_K_MATCH = 8,
    SIFT_HIP_K_GRAY = 9,
    SIFT_HIP_K_BLUR_ROW_SMALL = 10, /* the blur passes of octaves below 3 Mpixel (fewer outputs per thread) */
    SIFT_HIP_K_BLUR_COL_SMALL = 11,
    SIFT_HIP_K_RANSAC = 12, /* pair gather, sample fits, consensus counts */
    SIFT_HIP_K_WARP = 13,   /* perspective and cylindrical warps */
    SIFT_HIP_K_BLEND = 14,  /* multi-band blend */
    SIFT_HIP_K_BLUR_FUSED = 15, /* row and column pass of one level in one launch (short kernels, octaves of 3 Mpixel and more) */
    SIFT_HIP_K_COUNT = 16
};
/* Launch-shape switches of a context, for A/B measurements and tests (results never depend on
 * them; no counterpart in the reference).  key / value: "blur_seg" rows per segment of the
 * streaming column pass (0 = chosen per image); "blur_stream" -1 = where it pays, 0 = never,
 * 1 = wherever an instance exists; "blur_fused" -1 = where it pays, else the longest kernel that
 * takes the one-launch form at any image size; "blur_sweep" (float blur only) -1 = the sweeping
 * one-launch form where it pays, 0 = never, n > 0 = wherever an instance exists, n rows per segment;
 * "match_nb" 0 = by query count, 2 or 4;
 * "interleave" 0/1; "o3_place" where octave 3 of a single frame runs: 0 = behind octave 2 on the second
 * side stream, 1 = in front of octave 1 on the first, 2 = behind octave 1, 3 = in front of octave 2
 * (default), 4 = on the stream of the octaves beyond it.  The same switches are read from the environment once,
 * in sift_hip_create (SIFT_BLUR_SEG, SIFT_BLUR_STREAM, SIFT_BLUR_FUSED, SIFT_BLUR_SWEEP, SIFT_MATCH_NB,
 * SIFT_HIP_INTERLEAVE, SIFT_O3_PLACE); no other call looks at the environment. */
int sift_hip_set_tuning(sift_hip_ctx* ctx, const char* key, int value);
int sift_hip_profile(sift_hip_ctx* ctx, int enable);
/* Synchronises, then fills ms[k] = summed kernel time and launches[k] = launch
 * count of class k since the last reset; both arrays hold SIFT_HIP_K_COUNT entries. */
int sift_hip_profile_read(sift_hip_ctx* ctx, float* ms, int* launches, int reset);
const char* sift_hip_kernel_class_name(int k);

#ifdef __cplusplus
}
#endif
#endif /* SIFT_HIP_H */
