/*
 * lvk_oracle.h -- CPU ORACLE for the LiveVisionKit stabilization hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product (livevisionkit_amd/, include/)
 * never includes, links or calls anything in oracle/.
 *
 * What it is: a dependency-free CPU restatement of the reference algorithm for every stage on the
 * path `lvk::StabilizationFilter::filter` (reference: LiveVisionKit/Filters/StabilizationFilter.cpp:69-135).
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no tests, golden vectors or fixtures
 * (SURVEY.md section 4), cannot be compiled here (needs OpenCV 4.8.0 + OpenCL + Eigen 3.4, none present)
 * and most of the arithmetic lives in those absent third-party libraries.  The oracle is therefore
 * pinned only against (a) hand-computable known-answer tests and (b) synthetic ground truth
 * (tests/test_oracle_*.py).  Where the third-party library is not bit-defined (OpenCL fp contraction,
 * USAC randomisation, SIMD summation order) the oracle DEFINES the arithmetic; the definitions are
 * spelled out next to each function.
 *
 * Floating-point convention: all float math is IEEE-754 binary32 with round-to-nearest-even, compiled
 * with -ffp-contract=off; every fused multiply-add in the specification is written explicitly as
 * fmaf()/fma().  Division and sqrt are correctly rounded.
 */
#ifndef LVK_ORACLE_H
#define LVK_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Stage a14-a16: dense frame remap (EASU).  Reference: Functions/OpenCL/Sources/FSR.cl:55-452,
 * Functions/Image.cpp:28-151, Math/WarpMesh.cpp:183-223.
 * Frames are packed 8UC3 (3 bytes / pixel), `step` = row pitch in bytes.
 * `yuv` != 0 selects the program built with -D YUV_INPUT (Image.cpp:36-41), which -- reference quirk,
 * FSR.cl:229-241 -- uses the 3-channel pseudo luma; `yuv` == 0 uses channel 0 as luma.
 * ---------------------------------------------------------------------------------------------- */

/* FSR.cl:407-452 easu_remap_homography.  H = dst->src 3x3 (row major, already float: Image.cpp:137-139).
 * (off_x, off_y) = dst_bounds.xy (ROI offset, Image.cpp:121-123); dst is dst_cols x dst_rows. */
int lvko_remap_homography(const uint8_t* src, int src_step, int src_rows, int src_cols,
                          uint8_t* dst, int dst_step, int dst_rows, int dst_cols,
                          int off_x, int off_y, const float H[9], const uint8_t bg[3],
                          int yuv, int nthreads);

/* FSR.cl:362-403 easu_remap with the per-pixel offset map of WarpMesh.cpp:190-191 evaluated on the
 * fly: map = resize(mesh, dst size, INTER_LINEAR(_EXACT on f32 == INTER_LINEAR)) * (src_cols, src_rows).
 * mesh = mesh_rows x mesh_cols x 2 floats (normalised backward offsets).  dst size == src size. */
int lvko_remap_mesh(const uint8_t* src, int src_step, int src_rows, int src_cols,
                    uint8_t* dst, int dst_step,
                    const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3],
                    int yuv, int nthreads);

/* Materialise the W x H x 2 offset map exactly as WarpMesh.cpp:190-191 would (for tests). */
void lvko_mesh_to_map(const float* mesh, int mesh_rows, int mesh_cols, int rows, int cols, float* map);

/* cv::getPerspectiveTransform (OpenCV 4.8 imgproc): 8x8 double system, LU with partial
 * pivoting.  src/dst = 4 points (x0,y0,...).  M (row major 3x3, M[8] = 1) maps src -> dst. */
int lvko_get_perspective_transform(const float src[8], const float dst[8], double M[9]);

/* WarpMesh::apply (WarpMesh.cpp:183-223): 2x2 mesh -> homography branch, otherwise map branch. */
int lvko_warpmesh_apply(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dst, int dst_step,
                        const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3],
                        int yuv, int nthreads);

/* 2x2 branch only: the float 3x3 the kernel receives (WarpMesh.cpp:197-214 + Image.cpp:137-139). */
int lvko_mesh2x2_to_homography(const float mesh[8], int rows, int cols, float H[9]);


/* ------------------------------------------------------------------------------------------------
 * Stages a3-a7: tracker image operations (OpenCV 4.8.0 semantics; see oracle/imgproc.cpp, oracle/pyrlk.cpp).
 * ---------------------------------------------------------------------------------------------- */

/* a3+a4: VideoFrame::viewAsFormat(GRAY) for YUV (= channel 0, Data/VideoFrame.cpp:260) fused with
 * cv::resize(gray, T, INTER_AREA) (Vision/FrameTracker.cpp:117).  pix_stride = bytes per source pixel. */
int lvko_luma_area_resize(const uint8_t* src, int src_step, int pix_stride, int channel, int srows, int scols,
                          uint8_t* dst, int dst_step, int drows, int dcols);

/* cv::pyrDown 8UC1, BORDER_REFLECT_101; dst is ((cols+1)/2) x ((rows+1)/2). */
int lvko_pyr_down(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dst, int dst_step);

/* lkpyramid.cpp calcScharrDeriv: dst = rows x cols x (Ix, Iy) int16. */
int lvko_scharr_deriv(const uint8_t* src, int src_step, int rows, int cols, int16_t* dst);

/* a5 (inner): cv::FastFeatureDetector(threshold, nonmax=true, TYPE_9_16)->detect(img(roi)) (Vision/FeatureDetector.cpp:130-134).
 * Output triplets (x, y, score) in ROI-local coordinates, row-major.  Returns the total count. */
int lvko_fast9_16(const uint8_t* img, int step, int roi_x, int roi_y, int roi_w, int roi_h, int threshold,
                  int* out_xys, int cap);

/* a7: cv::SparsePyrLKOpticalFlow::calc (Vision/FrameTracker.cpp:140-146).  Returns the effective maxLevel. */
int lvko_pyrlk(const uint8_t* prev, int prev_step, const uint8_t* next, int next_step, int rows, int cols,
               const float* prev_pts, int n, float* next_pts, uint8_t* status,
               int win_w, int win_h, int max_level, int max_count, double epsilon, double min_eig_threshold);

/* the tracker with OpenCV's binary32 window sums instead of the specification's exact ones (lanes 1 / 4 / 8 / 16; pairs: v_dotprod pre-sums):
 * only there to measure the distance between the two (tests/test_pyrlk_float_order.py) */
int lvko_pyrlk_float(const uint8_t* prev, int prev_step, const uint8_t* next, int next_step, int rows, int cols,
                     const float* prev_pts, int n, float* next_pts, uint8_t* status,
                     int win_w, int win_h, int max_level, int max_count, double epsilon, double min_eig_threshold, int lanes, int pairs);
int lvko_pyramid_levels(int rows, int cols, int max_level, int win_w, int win_h, int* out_rows, int* out_cols);


/* ------------------------------------------------------------------------------------------------
 * Stage a9: robust global motion (oracle/ransac.cpp -- OUR deterministic specification).
 * ---------------------------------------------------------------------------------------------- */
int lvko_find_homography(const float* pts1, const float* pts2, int n, double threshold, double region_w, double region_h,
                         double H[9], uint8_t* mask);
int lvko_estimate_affine_partial(const float* pts1, const float* pts2, int n, double threshold, double region_w, double region_h,
                                 double H[9], uint8_t* mask);

/* Row a9, REFERENCE-SEMANTICS leg (oracle/usac_ref.cpp): the published algorithm of what FrameTracker.cpp:337-371 calls --
 * cv::findHomography(UsacParams{MAGSAC, LO_SIGMA, 50 iterations, ...}) and cv::estimateAffinePartial2D(RANSAC, thr, 50) of OpenCV 4.8 --
 * restated from knowledge of the upstream sources (unpinned: OpenCV is absent).  NOT the product's specification (that is the pair
 * above, frozen); used by tests/test_usac_semantics.py to bound the distance between the two.  max_thr: MAGSAC's maximum threshold
 * (<= 0: max(7.5, threshold)); rng_state: UsacParams::randomGeneratorState (the reference leaves it 0); final_lo: the one local
 * optimisation of the best model after a loop that ended before any had run (see usac_ref.cpp). */
int lvko_usac_find_homography(const float* pts1, const float* pts2, int n, double threshold, double max_thr, unsigned rng_state, int final_lo,
                              double H[9], uint8_t* mask, int* iters);
int lvko_ref_estimate_affine_partial(const float* pts1, const float* pts2, int n, double threshold, double H[9], uint8_t* mask);

/* ------------------------------------------------------------------------------------------------
 * Rows a1/a2/a5/a6/a8/a11/a12/a13: the stateful filter (oracle/stabilizer.cpp).
 * lvko_stab_settings flattens lvk::StabilizationFilterSettings (Filters/StabilizationFilter.hpp:28-39 and its
 * bases Vision/FrameTracker.hpp:31-44, Vision/FeatureDetector.hpp:28-37, Vision/PathSmoother.hpp:29-39).
 * The product's lvk_stab_settings (include/lvk_hip.h) has the same field order on purpose.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lvko_stab_settings
{
    int detection_width, detection_height;          /* FeatureDetectorSettings::detection_resolution */
    int detection_regions_x, detection_regions_y;   /* ::detection_regions */
    int force_detection;
    float max_feature_density, min_feature_density, accumulation_rate;
    int track_local_motions;                        /* FrameTrackerSettings */
    float temporal_smoothing, local_smoothing;
    int min_motion_samples;
    float acceptance_threshold, uniformity_threshold;
    int predictive_samples;                         /* PathSmootherSettings */
    float corrective_limit_x, corrective_limit_y;
    float smoothing_steps, response_rate;
    int motion_width, motion_height;                /* StabilizationFilterSettings::motion_resolution */
    float background[3];
    int crop_to_stable_region, stabilize_output;
    float min_scene_quality, min_tracking_quality;
} lvko_stab_settings;

typedef struct lvko_stab_stats
{
    float tracking_stability, scene_quality, trust, distribution;
    int n_detected, n_matched, n_tracked, frame_delay;
    double smoothing_factor;
    double homography[9];
} lvko_stab_stats;

typedef struct lvko_stab lvko_stab;

void lvko_stab_default_settings(lvko_stab_settings* s);
lvko_stab* lvko_stab_create(const lvko_stab_settings* settings);
void lvko_stab_destroy(lvko_stab* st);
void lvko_stab_configure(lvko_stab* st, const lvko_stab_settings* settings);
void lvko_stab_restart(lvko_stab* st);
void lvko_stab_set_lens(lvko_stab* st, const double* params /* 9 doubles or NULL; fused lens mode, restarts */);
int lvko_stab_push(lvko_stab* st, const uint8_t* frame, int step, int rows, int cols, uint64_t ts,
                   uint8_t* out, int out_step, uint64_t* out_ts, int nthreads);
/* same, for a 3-channel frame of VideoFrame::Format `format` (0 = BGR, 2 = RGB, 4 = YUV) */
int lvko_stab_push_fmt(lvko_stab* st, const uint8_t* frame, int step, int rows, int cols, uint64_t ts, int format,
                       uint8_t* out, int out_step, uint64_t* out_ts, int nthreads);
void lvko_stab_get_stats(const lvko_stab* st, lvko_stab_stats* out);
/* accumulated wall time per stage in ms: downscale, detect, LK, estimate, smooth, remap (bench.py's cpu_baseline; reset != 0 zeroes them) */
void lvko_stab_get_stage_ms(lvko_stab* st, double out[6], int reset);
int lvko_stab_get_meshes(const lvko_stab* st, float* motion, float* correction, int cap_floats);
int lvko_stab_get_features(const lvko_stab* st, float* xy_resp_age, int cap);
int lvko_stab_get_matches(const lvko_stab* st, float* p1, float* p2, int cap_pairs, int* estimator);   /* debug tap: pairs of the last estimate */


/* ------------------------------------------------------------------------------------------------
 * Stage a10: local motion (vector-field preset), oracle/mesh_solver.cpp.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lvko_mesh_solver lvko_mesh_solver;
lvko_mesh_solver* lvko_mesh_solver_create(int cols, int rows, float gen_region_w, float gen_region_h, float temporal_smoothing, float local_smoothing);
void lvko_mesh_solver_destroy(lvko_mesh_solver* s);
void lvko_mesh_solver_reset(lvko_mesh_solver* s);
int lvko_mesh_solver_static_rows(const lvko_mesh_solver* s);
int lvko_mesh_solver_static_triplets(const lvko_mesh_solver* s);
const float* lvko_mesh_solver_mesh(const lvko_mesh_solver* s);
int lvko_mesh_solver_solve(lvko_mesh_solver* s, const float* tracked, const float* matched, int n_pts,
                           float region_w, float region_h, float temporal_now, float threshold, uint8_t* inliers, float* offsets);


/* ------------------------------------------------------------------------------------------------
 * YUV420 <-> packed YUV444 either side of the filter (oracle/ingest.cpp;
 * reference Modules/OBS-Plugin/Interop/FrameIngest.cpp:494-602).
 * ---------------------------------------------------------------------------------------------- */
int lvko_ingest_yuv420(const uint8_t* y, int y_step, const uint8_t* u, int u_step, const uint8_t* v, int v_step, int nv12,
                       int rows, int cols, uint8_t* dst, int dst_step);
int lvko_egress_yuv420(const uint8_t* src, int src_step, int rows, int cols,
                       uint8_t* y, int y_step, uint8_t* u, int u_step, uint8_t* v, int v_step, int nv12);
/* every format of FrameIngest::Select (FrameIngest.cpp:36-75,476-753); fmt = libobs' enum video_format; Y800 frames have one channel, all others three */
int lvko_ingest_obs(int fmt, const uint8_t* const planes[3], const int steps[3], int rows, int cols, uint8_t* dst, int dst_step);
int lvko_egress_obs(int fmt, const uint8_t* src, int src_step, int rows, int cols, uint8_t* const planes[3], const int steps[3]);


/* ------------------------------------------------------------------------------------------------
 * lens correction (oracle/lens.cpp; reference Modules/OBS-Plugin/Sources/Enhancement/LCFilter.cpp:133-192)
 * and lvk::remap(src, dst, offset_map) with a materialised per-pixel offset map (Functions/Image.cpp:28-81).
 * ---------------------------------------------------------------------------------------------- */
int lvko_lens_offset_map(const double params[9], int rows, int cols, float* offsets, int view_xywh[4]);
int lvko_remap_map(const uint8_t* src, int src_step, int src_rows, int src_cols, uint8_t* dst, int dst_step,
                   const float* offsets /* rows x cols x 2, pixels */, const uint8_t bg[3], int yuv, int nthreads);

/* Fused lens mode (this repo's design for BASELINE config 5: one resampling instead of the reference chain's two).
 * model[17] = nfx, nfy, ncx, ncy, fx, fy, cx, cy, k1, k2, p1, p2, k3, kxc, vxc, kyc, vyc. */
int lvko_lens_model(const double params[9], int rows, int cols, double model[17]);
void lvko_lens_undistort_points(const double model[17], double sx, double sy, const float* pts, int n, float* out);
int lvko_warpmesh_apply_lens(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dst, int dst_step,
                             const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3],
                             int yuv, int nthreads, const double* model /* NULL = no lens */);

/* SURVEY section 8f row 4, second half: lvk::upscale (kernel easu_scale) and lvk::sharpen (kernel rcas) of ScalingFilter
 * (Functions/Image.cpp:155-233, Functions/OpenCL/Sources/FSR.cl:324-358,460-535, Filters/ScalingFilter.cpp:52-59).
 * `sharpness` in [0, 1] as the caller of lvk::sharpen passes it. */
int lvko_upscale(const uint8_t* src, int src_step, int src_rows, int src_cols,
                 uint8_t* dst, int dst_step, int dst_rows, int dst_cols, int yuv, int nthreads);
int lvko_sharpen(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dst, int dst_step, float sharpness, int nthreads);

/* Thread count of the row / point-parallel stages that take no `nthreads` argument (tracking-frame downscale, optical flow, 4:2:0
 * conversion); results do not depend on it.  Returns the previous value.  lvko_stab_push* set it from their own argument. */
int lvko_set_num_threads(int n);

/* native_recip / `1.0f / x` of FSR.cl = the DEVICE's reciprocal (v_rcp_f32 in oracle/_ref).  tab[m] = v_rcp_f32(as_float(0x3f800000 | m))
 * for the 2^23 mantissas m, read from the GPU by the tests; nullptr restores the default (the correctly rounded 1.0f / x).  See easu.cpp. */
int lvko_set_device_rcp_table(const float* tab, int n);

/* Debug overlays (oracle/draw.cpp; reference Functions/Drawing.tpp:53-93,146-196, Functions/OpenCL/Sources/Drawing.cl:22-39,75-105,
 * Filters/StabilizationFilter.cpp:163-188): drawn into the newest queued frame. */
int lvko_draw_grid(uint8_t* dst, int dst_step, int rows, int cols, int grid_w, int grid_h, const uint8_t colour[3], int thickness);
int lvko_draw_crosses(uint8_t* dst, int dst_step, int rows, int cols, const float* pts, int n, float scale_x, float scale_y,
                      const uint8_t colour[3], int cross_size, int cross_thickness);
void lvko_stab_draw_trackers(lvko_stab* st);
void lvko_stab_draw_motion_mesh(lvko_stab* st);

#ifdef __cplusplus
}
#endif
#endif
