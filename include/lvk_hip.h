/*
 * lvk_hip.h -- C-ABI of the MI355X-native LiveVisionKit stabilization hot path (liblvk_hip.so).
 *
 * This is the drop-in boundary: plain C, opaque handles, plain pointers and sizes, `int` status
 * (0 = ok, negative = error; lvk_hip_last_error() gives the text).  Nothing here throws and there are
 * no torch / OpenCV types.  Each entry point names the reference interface it replaces
 * (paths relative to the LiveVisionKit source tree).  The C++ facade in include/lvk/ (the
 * lvk::StabilizationFilter / lvk::VideoFilter API the OBS plugin compiles against) is a header-level
 * wrapper over these calls; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - "d_" pointers are device (HBM) pointers valid on the context's device; all others are host pointers.
 *   - Frames are packed 8UC3 (the reference's cv::UMat CV_8UC3 layout, Data/VideoFrame.hpp:25) with a
 *     row pitch `step` in bytes, or planar 8UC1 where stated.
 *   - All work is enqueued on the context's HIP stream and is asynchronous, exactly like the
 *     reference's `run_(..., false)` kernel launches (Functions/Image.cpp:76); results are complete
 *     after lvk_hip_sync() (reference: Stopwatch::sync_gpu -> cv::ocl::finish, Timing/Stopwatch.cpp:127-131).
 *   - One context is driven by one host thread at a time; different contexts are independent
 *     (reference threading contract: SURVEY.md section 8b).  Every entry point makes its context's device current for
 *     the duration of the call and restores the caller's: a context of device 3 may be driven from a thread that never
 *     called hipSetDevice (one host thread + one context per GPU).
 *
 * Layout of this header
 *   PART 1 -- STABLE ABI: what a host of the reference binds (the filter, its frames and memory, the lvk:: image operations).
 *             A change that breaks one of these signatures or contracts increments LVK_HIP_ABI_VERSION.
 *   PART 2 -- EXPERIMENTAL / DIAGNOSTICS: per-stage entry points the parity tests drive, taps, profiling, the device-frame
 *             look-ahead.  They may change between builds of the library without a version increment; production hosts
 *             do not need them.
 */
#ifndef LVK_HIP_H
#define LVK_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVK_HIP_OK              0
#define LVK_HIP_ERR_ARG        -1   /* violated pre-condition (the reference would LVK_ASSERT) */
#define LVK_HIP_ERR_RUNTIME    -2   /* HIP runtime error */
#define LVK_HIP_ERR_NO_DEVICE  -3   /* no usable gfx950 device / extension cannot run */

typedef struct lvk_hip_ctx lvk_hip_ctx;

/* =====================================================================================================================================
 * PART 1 -- STABLE ABI
 * ===================================================================================================================================== */

/* ---- context ------------------------------------------------------------------------------------
 * Replaces the implicit OpenCL context/queue of cv::ocl (Functions/OpenCL/Kernels.cpp:27-45).
 * lvk_hip_ctx_create makes its own non-blocking stream; lvk_hip_ctx_create_on_stream enqueues on the caller's
 * hipStream_t (NULL = the device's default stream) so the work is ordered with the caller's own GPU work. */
int  lvk_hip_ctx_create(int device, lvk_hip_ctx** out);
int  lvk_hip_ctx_create_on_stream(int device, void* hip_stream, lvk_hip_ctx** out);
void lvk_hip_ctx_destroy(lvk_hip_ctx* ctx);
int  lvk_hip_sync(lvk_hip_ctx* ctx);                 /* Stopwatch::sync_gpu, Timing/Stopwatch.cpp:127-131 */
void* lvk_hip_stream(lvk_hip_ctx* ctx);              /* the hipStream_t work is enqueued on */
const char* lvk_hip_last_error(lvk_hip_ctx* ctx);    /* NULL ctx: last error of a failed ctx_create */
const char* lvk_hip_version(void);                   /* human-readable build string */
/* ABI number of PART 1 of this header as the library was built (compare with LVK_HIP_ABI_VERSION of the header a host was compiled against;
 * tests/test_abi.py holds the two together). */
#define LVK_HIP_ABI_VERSION 6
int  lvk_hip_abi_version(void);
/* Devices of this process: contexts are addressed by HIP device index, and lvk_hip_device_count() is the number of indices worth trying -- the
 * highest gfx950 index + 1 (0 when there is no gfx950 device; never an error).  On the usual host every index below it is an MI355X; on a mixed
 * host (an integrated GPU or another architecture in between) lvk_hip_device_usable(d) says which indices lvk_hip_ctx_create accepts (the others
 * are refused with LVK_HIP_ERR_NO_DEVICE) -- or set HIP_VISIBLE_DEVICES.  One lvk_hip_ctx + one host thread per usable device is the multi-GPU
 * partitioning (no collective, no peer access). */
int  lvk_hip_device_count(void);
int  lvk_hip_device_usable(int device);

/* Ordering between contexts: everything enqueued so far on `producer` (its stream and the streams of its stabilizers) happens
 * before whatever is enqueued on `ctx` from now on.  GPU-side (an event per stream), no host wait.  What the reference gets from
 * OpenCV's single in-order OpenCL queue; needed here when a frame written on one context's stream is consumed on another's. */
int  lvk_hip_ctx_wait(lvk_hip_ctx* ctx, lvk_hip_ctx* producer);

/* ---- device memory helpers (for hosts without their own allocator) ------------------------------
 * Replace cv::UMat(USAGE_ALLOCATE_DEVICE_MEMORY) allocation / upload / download (Data/VideoFrame.cpp:27-29).
 * Like OpenCV's OpenCL buffer pool, lvk_hip_free keeps the block for the next lvk_hip_malloc of the same size (no device
 * synchronisation, no allocation in steady state; at most 4 GiB are kept, lvk_hip_trim gives them back).  A freed block may be
 * handed out again at once: work that still uses it must be on this context's stream, or have been fenced with lvk_hip_ctx_wait.
 * A block freed through another live context of the process goes back to the pool of the context that allocated it; freeing a block
 * that already sits in a pool (a double free) returns LVK_HIP_ERR_ARG. */
int lvk_hip_malloc(lvk_hip_ctx* ctx, size_t bytes, void** d_ptr);
int lvk_hip_free(lvk_hip_ctx* ctx, void* d_ptr);
int lvk_hip_trim(lvk_hip_ctx* ctx);
int lvk_hip_upload(lvk_hip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);     /* async on the stream */
int lvk_hip_download(lvk_hip_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);   /* async on the stream */

/* pinned host memory for the planes of lvk_hip_stab_push_yuv420_host */
int  lvk_hip_host_malloc(lvk_hip_ctx* ctx, size_t bytes, void** h_ptr);      /* pinned, device-visible host memory */
int  lvk_hip_host_free(lvk_hip_ctx* ctx, void* h_ptr);


/* ---- a15/a16: dense remap ------------------------------------------------------------------------
 * lvk::remap(src, dst, homography, background, inverted=true)  (Functions/Image.cpp:85-151) running
 * easu_remap_homography (Functions/OpenCL/Sources/FSR.cl:407-452).  H = dst->src 3x3, row major, already
 * cast to float as Image.cpp:137-139 does.  (off_x, off_y) = ROI offset of dst (Image.cpp:121-123).
 * yuv != 0 <=> src.format == VideoFrame::YUV (selects the "-D YUV_INPUT" program, Image.cpp:36-41). */
int lvk_hip_remap_homography(lvk_hip_ctx* ctx,
                             const void* d_src, int src_step, int src_rows, int src_cols,
                             void* d_dst, int dst_step, int dst_rows, int dst_cols,
                             int off_x, int off_y, const float H[9], const uint8_t bg[3], int yuv);

/* lvk::remap(src, dst, offset_map, background) (Functions/Image.cpp:28-81, easu_remap FSR.cl:362-403) with the
 * offset map of WarpMesh::apply (Math/WarpMesh.cpp:190-191) evaluated inside the kernel from the mesh
 * vertices instead of being materialised (saves 4 full-resolution float2 passes).  `mesh` is a HOST pointer
 * to mesh_rows x mesh_cols x 2 floats of normalised backward offsets; dst has the size of src. */
int lvk_hip_remap_mesh(lvk_hip_ctx* ctx,
                       const void* d_src, int src_step, int src_rows, int src_cols,
                       void* d_dst, int dst_step,
                       const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv);

/* lvk::remap(src, dst, offset_map, background) with a materialised map (Functions/Image.cpp:28-81, FSR.cl:362-403):
 * d_map = rows x cols float2 offsets in pixels resident in HBM (pitch map_step bytes); dst has the size of src. */
int lvk_hip_remap_map(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step,
                      const void* d_map, int map_step, const uint8_t bg[3], int yuv);

/* lvk::upscale(src, dst, size, yuv) (Functions/Image.cpp:155-202, kernel easu_scale FSR.cl:324-358): EASU upsampling of an
 * 8UC3 frame to dst_cols x dst_rows >= the source size (equal size = copy).  d_dst must not alias d_src. */
int lvk_hip_upscale(lvk_hip_ctx* ctx, const void* d_src, int src_step, int src_rows, int src_cols,
                    void* d_dst, int dst_step, int dst_rows, int dst_cols, int yuv);

/* lvk::sharpen(src, dst, sharpness) (Functions/Image.cpp:206-233, kernel rcas FSR.cl:460-535): RCAS with sharpness in [0, 1].
 * OUT OF PLACE: the reference's ScalingFilter sharpens in place (Filters/ScalingFilter.cpp:57), where neighbour reads race
 * with writes; the defined result is that of distinct buffers, and aliasing is rejected.  Border pixels are copied. */
int lvk_hip_sharpen(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step, float sharpness);

/* Lens correction: the offset map LCFilter::prepare_undistort_maps builds
 * (Modules/OBS-Plugin/Sources/Enhancement/LCFilter.cpp:133-171) for a camera profile in the plugin's format
 * (Modules/OBS-Plugin/Sources/Tools/CCTool.cpp:120-153); LCFilter::filter == lvk_hip_remap_map with it. */
typedef struct lvk_camera_params { double fx, fy, cx, cy, k1, k2, p1, p2, k3; } lvk_camera_params;
int lvk_hip_lens_map_create(lvk_hip_ctx* ctx, const lvk_camera_params* params, int rows, int cols, void** d_map, int view_xywh[4]);
int lvk_hip_lens_map_destroy(lvk_hip_ctx* ctx, void* d_map);
/* FUSED lens mode (BASELINE config 5; this library's design, the reference only has the two-pass chain): the same warp in
 * closed form, composed into the coordinate of the stabilizing remap -> one EASU resampling, no map traffic.
 * lvk_hip_warpmesh_apply_lens == WarpMesh::apply applied to the lens-corrected frame, sampled from the RAW frame d_src;
 * lvk_hip_lens_undistort_points = raw tracking-frame points (scale sx, sy = frame / tracking size) -> corrected positions. */
int lvk_hip_warpmesh_apply_lens(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step,
                                const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv, const lvk_camera_params* lens);
int lvk_hip_lens_undistort_points(lvk_hip_ctx* ctx, const lvk_camera_params* params, int rows, int cols, double sx, double sy,
                                  const float* pts, int n, float* out);

/* WarpMesh::apply followed by the plugin's 4:2:0 egress (I4XXIngest / NV12Ingest::to_obs, Modules/OBS-Plugin/Interop/FrameIngest.cpp:
 * 540-557,590-602) in one kernel: packed YUV 8UC3 in, I420 (y, u, v) or NV12 (y, uv; o_v ignored) planes out, even dimensions.
 * Bit-identical to lvk_hip_warpmesh_apply + lvk_hip_egress_yuv420. */
int lvk_hip_warpmesh_apply_yuv420(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols,
                                  void* o_y, int oy_step, void* o_u, int ou_step, void* o_v, int ov_step, int nv12,
                                  const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3]);

/* WarpMesh::apply(src, dst, background) (Math/WarpMesh.cpp:183-223): a 2x2 mesh goes through
 * cv::getPerspectiveTransform + the homography kernel, anything larger through the mesh kernel. */
int lvk_hip_warpmesh_apply(lvk_hip_ctx* ctx,
                           const void* d_src, int src_step, int rows, int cols,
                           void* d_dst, int dst_step,
                           const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv);

/* ---- section 8f row 2: YUV420 <-> packed YUV444 either side of the filter --------------------------------------
 * I4XXIngest::to_ocl / NV12Ingest::to_ocl (Modules/OBS-Plugin/Interop/FrameIngest.cpp:494-522,567-585): chroma
 * cv::resize(INTER_LINEAR) + merge into the packed 8UC3 frame the filter consumes; and ::to_obs (:526-557,589-602):
 * split + cv::resize(0.5, 0.5, INTER_AREA).  nv12 != 0: d_u is the interleaved UV plane and d_v is ignored.
 * rows and cols must be even. */
int lvk_hip_ingest_yuv420(lvk_hip_ctx* ctx, const void* d_y, int y_step, const void* d_u, int u_step, const void* d_v, int v_step, int nv12,
                          int rows, int cols, void* d_dst, int dst_step);
int lvk_hip_egress_yuv420(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols,
                          void* d_y, int y_step, void* d_u, int u_step, void* d_v, int v_step, int nv12);

/* Every video format FrameIngest::Select accepts (Modules/OBS-Plugin/Interop/FrameIngest.cpp:36-75): video_format is libobs' enum video_format
 * (media-io/video-io.h of libobs 27.2.4), i.e. obs_source_frame::format passed through.  d_planes / steps = the frame's data[] / linesize[] on the
 * device (3 entries; unused ones NULL / 0; the alpha planes of I40A / I42A / YUVA are not touched, as in the reference).
 *   I420 / I40A / NV12            -> lvk_hip_ingest_yuv420 / lvk_hip_egress_yuv420 above (I4XXIngest, NV12Ingest)
 *   I422 / I42A                   I4XXIngest with chroma (cols / 2) x rows: INTER_LINEAR upsampling of the width in, INTER_AREA (0.5, 1.0) out (:494-557)
 *   I444 / YUVA                   merge / split (:521,531)
 *   YUY2 / YVYU / UYVY            P422Ingest::to_ocl / to_obs (:615-666): 2 bytes per pixel, chroma shared by a pixel pair
 *   AYUV                          P444Ingest (:679-703): the last three of A Y U V in; A = 255 out
 *   Y800 / BGR3                   DirectIngest (:728-753): the bytes as they are (Y800: a one-channel frame, which lvk_hip_stab_push refuses like the
 *                                 reference's lvk::remap does, Functions/Image.cpp:32)
 *   RGBA / BGRA / BGRX            DirectIngest as written: rows * cols * 3 BYTES from data[0] viewed as 3-byte pixels (:743-747) -- the colour planes are
 *                                 not separated; steps[0] must be 4 * cols.  Reproduced, not endorsed.
 * cols even for the 4:2:2 formats, rows and cols even for 4:2:0.  The frame is 8UC3 (Y800: 8UC1) of rows x cols; lvk_hip_obs_frame_format gives the
 * LVK_FORMAT_* its pixels carry (the VideoFrame::Format each ingest declares), negative for a format FrameIngest::Select rejects. */
#define LVK_VIDEO_FORMAT_I420 1
#define LVK_VIDEO_FORMAT_NV12 2
#define LVK_VIDEO_FORMAT_YVYU 3
#define LVK_VIDEO_FORMAT_YUY2 4
#define LVK_VIDEO_FORMAT_UYVY 5
#define LVK_VIDEO_FORMAT_RGBA 6
#define LVK_VIDEO_FORMAT_BGRA 7
#define LVK_VIDEO_FORMAT_BGRX 8
#define LVK_VIDEO_FORMAT_Y800 9
#define LVK_VIDEO_FORMAT_I444 10
#define LVK_VIDEO_FORMAT_BGR3 11
#define LVK_VIDEO_FORMAT_I422 12
#define LVK_VIDEO_FORMAT_I40A 13
#define LVK_VIDEO_FORMAT_I42A 14
#define LVK_VIDEO_FORMAT_YUVA 15
#define LVK_VIDEO_FORMAT_AYUV 16
int lvk_hip_ingest_obs(lvk_hip_ctx* ctx, int video_format, const void* const d_planes[3], const int steps[3], int rows, int cols, void* d_dst, int dst_step);
int lvk_hip_egress_obs(lvk_hip_ctx* ctx, int video_format, const void* d_src, int src_step, int rows, int cols, void* const d_planes[3], const int steps[3]);
int lvk_hip_obs_frame_format(int video_format);

/* ---- a1/a2: the stabilization filter ----------------------------------------------------------------------
 * lvk_stab_settings flattens lvk::StabilizationFilterSettings (Filters/StabilizationFilter.hpp:28-39) and its bases
 * FrameTrackerSettings (Vision/FrameTracker.hpp:31-44) : FeatureDetectorSettings (Vision/FeatureDetector.hpp:28-37) and
 * PathSmootherSettings (Vision/PathSmoother.hpp:29-39); same names, same defaults (lvk_stab_default_settings). */
typedef struct lvk_stab_settings
{
    int detection_width, detection_height;          /* detection_resolution {256, 256} */
    int detection_regions_x, detection_regions_y;   /* detection_regions {2, 2} */
    int force_detection;                            /* false */
    float max_feature_density, min_feature_density, accumulation_rate;    /* 0.20, 0.05, 2.0 */
    int track_local_motions;                        /* true */
    float temporal_smoothing, local_smoothing;      /* 1.0, 20.0 */
    int min_motion_samples;                         /* 75 */
    float acceptance_threshold, uniformity_threshold;                     /* 8.0, 0.20 */
    int predictive_samples;                         /* 10 */
    float corrective_limit_x, corrective_limit_y;   /* corrective_limits {0.1, 0.1} */
    float smoothing_steps, response_rate;           /* 20.0, 0.04 */
    int motion_width, motion_height;                /* StabilizationFilterSettings::motion_resolution {2, 2} */
    float background[3];                            /* background_colour {255, 0, 255} */
    int crop_to_stable_region, stabilize_output;    /* false, true */
    float min_scene_quality, min_tracking_quality;  /* 0.8, 0.3 */
} lvk_stab_settings;

typedef struct lvk_stab_stats
{
    float tracking_stability;      /* FrameTracker::tracking_stability() */
    float scene_quality, trust;    /* m_SceneQuality, m_TrustFactor */
    float distribution;            /* FeatureDetector::detect() return value of the last frame */
    int n_detected, n_matched, n_tracked, frame_delay;
    double smoothing_factor;       /* PathSmoother m_SmoothingFactor */
    double homography[9];          /* last global motion estimate (tracking-resolution pixels) */
} lvk_stab_stats;

/* lvk::VideoFrame::Format (Data/VideoFrame.hpp:27) */
#define LVK_FORMAT_BGR 0
#define LVK_FORMAT_BGRA 1
#define LVK_FORMAT_RGB 2
#define LVK_FORMAT_RGBA 3
#define LVK_FORMAT_YUV 4
#define LVK_FORMAT_GRAY 5

typedef struct lvk_hip_stab lvk_hip_stab;

void lvk_stab_default_settings(lvk_stab_settings* s);
/* StabilizationFilter(settings) / configure(settings) (Filters/StabilizationFilter.cpp:34-65) */
int  lvk_hip_stab_create(lvk_hip_ctx* ctx, const lvk_stab_settings* settings, lvk_hip_stab** out);
void lvk_hip_stab_destroy(lvk_hip_stab* stab);
int  lvk_hip_stab_configure(lvk_hip_stab* stab, const lvk_stab_settings* settings);
int  lvk_hip_stab_restart(lvk_hip_stab* stab);            /* :139-144 */
/* Debug overlays into the newest queued frame, i.e. the caller's borrowed buffer of the last push (StabilizationFilter.cpp:163-188;
 * kernels Functions/OpenCL/Sources/Drawing.cl:22-39,75-105): crosses at the tracked features coloured lerp(RED, GREEN, trust);
 * BLUE grid with motion_resolution - 1 cells.  Asynchronous on the context's stream. */
int  lvk_hip_stab_draw_trackers(lvk_hip_stab* stab);
int  lvk_hip_stab_draw_motion_mesh(lvk_hip_stab* stab);
/* lvk::draw_grid / lvk::draw_crosses on a packed 8UC3 device frame (Functions/Drawing.tpp:53-93,146-196); pts_xy = n host (x, y)
 * floats, scaled by (scale_x, scale_y) and rounded to pixels like cv::multiply(.., CV_32S). */
int  lvk_hip_draw_grid(lvk_hip_ctx* ctx, void* d_dst, int dst_step, int rows, int cols, int grid_w, int grid_h, const uint8_t colour[3], int thickness);
int  lvk_hip_draw_crosses(lvk_hip_ctx* ctx, void* d_dst, int dst_step, int rows, int cols, const float* pts_xy, int n,
                          float scale_x, float scale_y, const uint8_t colour[3], int cross_size, int thickness);
/* Fused lens pre-warp for the stream this filter stabilizes: frames are pushed RAW (uncorrected); the tracker estimates the
 * motion between lens-corrected feature positions and the output remap composes lens map and stabilizing warp.
 * params = NULL switches it off.  Restarts the filter (queued frames are dropped). */
int  lvk_hip_stab_set_lens(lvk_hip_stab* stab, const lvk_camera_params* params);
int  lvk_hip_stab_reset_context(lvk_hip_stab* stab);      /* :155-159 */
int  lvk_hip_stab_ready(const lvk_hip_stab* stab);        /* :148-151 */
int  lvk_hip_stab_frame_delay(const lvk_hip_stab* stab);  /* :192-195 */
int  lvk_hip_stab_stable_region(const lvk_hip_stab* stab, int rows, int cols, int rect_xywh[4]);   /* :199-205 */

/* VideoFilter::apply(std::move(input), output) -> StabilizationFilter::filter (Filters/VideoFilter.cpp:46-51,
 * Filters/StabilizationFilter.cpp:69-135).  d_frame: packed 8UC3 device frame; like the reference's moved-in
 * input it is BORROWED (not copied) until it has been emitted `frame_delay` pushes later -- `*released` then
 * returns its pointer (or NULL); the caller may reuse that buffer once the stream has passed this call.
 * d_out receives the stabilized delayed frame when *produced == 1 (and carries *out_timestamp = that frame's
 * timestamp, Math/WarpMesh.cpp:221-222); *produced == 0 while the delay builds ("output.release()").
 * The output remap is only enqueued: call lvk_hip_sync() before reading d_out on the host.
 *
 * SIZE OF THE OUTPUT.  The queue holds whole frames (StabilizationFilter.cpp:118-131) and the emitted frame is the DELAYED one, at ITS
 * OWN size and format (WarpMesh::apply allocates dst from the delayed source, Math/WarpMesh.cpp:183-223 -> Functions/Image.cpp:53,116):
 * when the frame size changes in the middle of a stream (an OBS source that is resized -- VSFilter.cpp:352-364 does not restart its
 * filter), the next `frame_delay` pushes still emit frames of the OLD size.  lvk_hip_stab_next_output() tells the caller, before the
 * push, whether the push of a rows x cols frame will emit and what (1 / 0; *out = the delayed frame's rows, cols, format) -- size d_out
 * from it.  d_out is a buffer of out_rows rows of out_step bytes: a push whose output would not fit (out_step < 3 * cols or out_rows <
 * rows of the frame to be emitted, or d_out == NULL when a frame is due) is REFUSED with LVK_HIP_ERR_ARG BEFORE anything changes -- the
 * frame is not queued, nothing is released, the filter is as it was; the same push with a large enough d_out then succeeds.  The rows x
 * cols top-left part of d_out is written, nothing else.  *emitted (optional) = geometry and format of the frame written to d_out. */
typedef struct lvk_frame_info { int rows, cols, format; } lvk_frame_info;
int  lvk_hip_stab_next_output(const lvk_hip_stab* stab, int rows, int cols, int format, lvk_frame_info* out);
int  lvk_hip_stab_push(lvk_hip_stab* stab, const void* d_frame, int step, int rows, int cols, uint64_t timestamp, int format,
                       void* d_out, int out_step, int out_rows, int* produced, uint64_t* out_timestamp, const void** released,
                       lvk_frame_info* emitted);

/* The OBS asynchronous path in one call (Modules/OBS-Plugin/Interop/VisionFilter.cpp:151-212): YUV 4:2:0 planes in
 * (I420, or NV12 with nv12 != 0 and d_u = interleaved UV), ingest -> filter -> egress, 4:2:0 planes out.  The packed
 * frames the filter queues live in an internal pool; a warped frame leaves through one kernel that remaps and writes the planes.
 * Input planes are consumed when the call returns; output planes are complete after lvk_hip_sync().  A filter is fed EITHER
 * through this call OR through lvk_hip_stab_push -- switching needs lvk_hip_stab_restart() (the two own their queued frames
 * differently).
 * SIZE OF THE OUTPUT: as for lvk_hip_stab_push -- the emitted frame is the DELAYED one at its own size (the reference's queue holds whole
 * frames, StabilizationFilter.cpp:118-131), so after rows / cols CHANGE in the middle of a stream the next `frame_delay` pushes emit frames of
 * the OLD size (round 6; rounds 2-5 dropped them).  lvk_hip_stab_next_output(stab, rows, cols, LVK_FORMAT_YUV, &info) says what the push will
 * emit; the output planes hold o_rows luma rows (o_rows / 2 chroma rows) at the given pitches, and a push whose output would not fit is refused
 * with LVK_HIP_ERR_ARG before the filter changes.  *emitted (optional) = the geometry of the frame written. */
int  lvk_hip_stab_push_yuv420(lvk_hip_stab* stab, const void* d_y, int y_step, const void* d_u, int u_step, const void* d_v, int v_step, int nv12,
                              int rows, int cols, uint64_t timestamp,
                              void* o_y, int oy_step, void* o_u, int ou_step, void* o_v, int ov_step, int o_rows,
                              int* produced, uint64_t* out_timestamp, lvk_frame_info* emitted);
/* The same call for ANY video format FrameIngest::Select accepts except Y800 (LVK_VIDEO_FORMAT_*; d_planes / steps as for lvk_hip_ingest_obs, o_planes /
 * o_steps / o_rows the planes of the emitted frame and their row capacity): I420 / I40A / NV12 take lvk_hip_stab_push_yuv420's route; the other formats
 * are converted into the filter's frame pool (FrameIngest::to_ocl), pushed, and the emitted frame -- the DELAYED one, at its own size -- converted back
 * (::to_obs).  Input planes consumed when the call returns; output planes complete after lvk_hip_sync(); planes that cannot hold the emitted frame are
 * refused before anything changes.  Shares the frame queue with lvk_hip_stab_push_yuv420. */
int  lvk_hip_stab_push_obs(lvk_hip_stab* stab, int video_format, const void* const d_planes[3], const int steps[3], int rows, int cols, uint64_t timestamp,
                           void* const o_planes[3], const int o_steps[3], int o_rows, int* produced, uint64_t* out_timestamp, lvk_frame_info* emitted);

/* The same path for frames that live in HOST memory -- FrameIngest::upload_planes -> to_ocl -> filter -> to_obs -> download_planes
 * (Modules/OBS-Plugin/Interop/FrameIngest.cpp:415-474,494-602): h_* / oh_* are planes in PINNED host memory (lvk_hip_host_malloc,
 * hipHostMalloc or hipHostRegister; contiguous planes, the OBS layout, travel as one copy).  The library schedules the transfers: luma
 * first (the tracker starts while the chroma planes are still on the link), one upload stream; the output planes are written by the
 * remap kernel ITSELF, straight into the pinned host planes (zero copy; the download route -- remap to device planes, then a D2H copy --
 * is kept behind LVK_HIP_HOST_SINK=copy for comparison: the runtime performs that copy with a blit kernel that is slower for every
 * caller measured) -- same pixels.  Input planes are consumed when the call returns; output planes are complete after
 * lvk_hip_sync().  rows and cols even.  Shares the frame queue with lvk_hip_stab_push_yuv420 (the two may be mixed), and its output sizing: the
 * output planes have the DELAYED frame's size (lvk_hip_stab_next_output), o_rows luma rows of capacity.  Pageable plane
 * pointers are refused with LVK_HIP_ERR_ARG: both ends of every plane are looked up on every call (an address that was pinned once may
 * have been freed and handed out again as pageable memory). */
int  lvk_hip_stab_push_yuv420_host(lvk_hip_stab* stab, const void* h_y, int y_step, const void* h_u, int u_step, const void* h_v, int v_step, int nv12,
                                   int rows, int cols, uint64_t timestamp,
                                   void* oh_y, int oy_step, void* oh_u, int ou_step, void* oh_v, int ov_step, int o_rows,
                                   int* produced, uint64_t* out_timestamp, lvk_frame_info* emitted);
/* Look-ahead for streaming callers (the reader thread of VideoFilter::stream, Filters/VideoFilter.cpp:62-209, uploads ahead of the
 * filter thread): starts the upload of the planes that the NEXT lvk_hip_stab_push_yuv420_host call will push (same pointers), so that
 * the link carries frame n + 1 while frame n is tracked: announce frame n + 1, then push frame n.  Announced frames are pushed in the
 * order announced, at most two outstanding; the planes stay the caller's until their push returns. */
int  lvk_hip_stab_prefetch_yuv420_host(lvk_hip_stab* stab, const void* h_y, int y_step, const void* h_u, int u_step, const void* h_v, int v_step,
                                       int nv12, int rows, int cols);
/* Forget the frames that were announced and not pushed (a caller that stops, seeks or switches buffers after announcing frame n + 1 --
 * VideoFilter::stream's reader thread ending on a failed read, Filters/VideoFilter.cpp:77-103).  Returns once their uploads no longer
 * read the caller's planes.  lvk_hip_stab_restart() implies it. */
int  lvk_hip_stab_prefetch_cancel(lvk_hip_stab* stab);
/* Optional: run the bulk kernels (4:2:0 ingest, the output remap) on a second, low-priority HIP stream so that they overlap the
 * tracking of the next frame (the reference gets the same effect from OpenCL's asynchronous `run_(..., false)` launches,
 * Functions/Image.cpp:76); the remap then uses its occupancy-capped variants.  With overlap enabled d_out is complete only after
 * lvk_hip_sync(), and *released reports a borrowed frame one push later (after its remap has finished).
 * Ordering: INPUTS need nothing from the caller -- whatever was enqueued on the context's stream before a push (the decode / copy that
 * fills the frame or the planes) is made visible to the bulk stream by an event the push records.  OUTPUTS are produced on the
 * bulk stream: consume them there (lvk_hip_stab_output_stream) or after lvk_hip_sync(); reusing the same output buffer for the
 * next push is safe (same stream).  Frames dropped outside a push (queue shrunk by configure, mode toggled) come back through
 * *released of the following pushes. */
int  lvk_hip_stab_set_overlap(lvk_hip_stab* stab, int enable);
/* The same mode on a stream the caller owns: the bulk kernels run on the stream of `bulk` (a second context on the same device; NULL
 * switches overlap off).  For hosts whose output frames outlive the stabilizer or feed stream-ordered consumers: outputs belong to `bulk`. */
int  lvk_hip_stab_set_bulk_context(lvk_hip_stab* stab, lvk_hip_ctx* bulk);
/* The hipStream_t the outputs of the following pushes are produced on (the bulk stream in overlap mode, else the context's
 * stream): enqueue stream-ordered consumers of d_out / the output planes (a D2H copy, an encoder) there instead of calling
 * lvk_hip_sync().  Changes when lvk_hip_stab_set_overlap or stabilize_output change. */
void* lvk_hip_stab_output_stream(lvk_hip_stab* stab);


/* =====================================================================================================================================
 * PART 2 -- EXPERIMENTAL / DIAGNOSTICS  (no ABI promise: per-stage entry points of the parity tests, taps, profiling, device look-ahead)
 * ===================================================================================================================================== */

/* Look-ahead for DEVICE-resident frames, for callers that hold the next frame already (VideoFilter::stream's reader thread runs ahead of
 * its filter thread, Filters/VideoFilter.cpp:62-209; a transcoder whose clip is resident): announce frame n + 1, then push frame n.  The
 * push puts the luma downscale and the pyramid of frame n + 1 on the tracking stream behind its own chain, where the GPU runs them during
 * the host's turn, and the push of frame n + 1 starts at the optical flow.  Only the luma is read ahead (Y plane; channel 0 / the grey
 * value of a packed frame); it must not change between the announcement and the return of the push that carries it.  The announcement
 * holds for the very next push only; a push that carries other planes or another geometry, or that does not track, works as if nothing
 * had been announced.  Same pixels either way.  lvk_hip_stab_prefetch_cancel / lvk_hip_stab_restart forget it.
 * lvk_hip_stab_lookahead_frames: the pushes so far that found their pyramid built. */
int  lvk_hip_stab_prefetch(lvk_hip_stab* stab, const void* d_frame, int step, int rows, int cols, int format);
int  lvk_hip_stab_prefetch_yuv420(lvk_hip_stab* stab, const void* d_y, int y_step, const void* d_u, int u_step, const void* d_v, int v_step,
                                  int nv12, int rows, int cols);
long long lvk_hip_stab_lookahead_frames(lvk_hip_stab* stab);

/* Which schedule the pushes of this filter took so far.  The library picks per push, from what it sees the caller doing (is the bulk stream still
 * busy with the previous remap? did this push begin within 15 us of the last one's return?), between the schedule of a FREE-RUNNING caller
 * (persistent remap grid of 4 blocks per CU next to the tracker, completion through an event) and that of a caller that WAITS for every frame
 * (full remap grid, completion through a word in host memory): same pixels, different throughput / latency.  (After eight or more free-running pushes
 * the first push that looks synchronous still takes the free-running schedule -- one synchronisation does not make a synchronous caller --, the second
 * in a row switches.)  A host -- and bench.py, per leg --
 * reads here which one its pushes got.  reset != 0 zeroes the counters after reading. */
#define LVK_SCHED_PUSH_FREE_RUNNING   0   /* pushes taken as a free-running caller's ... */
#define LVK_SCHED_PUSH_SYNCHRONISED   1   /* ... and as those of a caller that waits for every frame */
#define LVK_SCHED_INGEST_ON_TRACKER   2   /* 4:2:0 conversions placed behind the chain on the tracking stream ... */
#define LVK_SCHED_INGEST_ON_BULK      3   /* ... on the bulk stream (overlap mode) ... */
#define LVK_SCHED_INGEST_INLINE       4   /* ... ahead of the tracker on the context's stream (no overlap) */
#define LVK_SCHED_REMAP_PERSISTENT    5   /* output remaps launched as the persistent co-scheduled grid ... */
#define LVK_SCHED_REMAP_FULL          6   /* ... as the full grid */
#define LVK_SCHED_WAIT_SIGNAL_WORD    7   /* chain completions taken from the host signal word ... */
#define LVK_SCHED_WAIT_EVENT          8   /* ... from hipEventSynchronize / hipStreamSynchronize */
#define LVK_SCHED_WAIT_WORD_TIMEOUT   9   /* signal-word waits that fell through to the event / stream wait (bounded spin) */
#define LVK_SCHED_COUNT              10
int  lvk_hip_stab_schedule_counters(lvk_hip_stab* stab, long long out[LVK_SCHED_COUNT], int reset);

int  lvk_hip_stab_get_stats(const lvk_hip_stab* stab, lvk_stab_stats* out);
/* Frames on which FeatureDetector::detect ran FAST so far (Vision/FeatureDetector.cpp:125-157), by where the corners went through the
 * suppression grid: on the device inside the tracker's chain of kernels, or in the host loop (grids beyond 4096 cells, detection regions off
 * the pixel grid, LVK_HIP_HOST_GRID=1).  Same features either way; a tap for tests and tuning. */
int  lvk_hip_stab_detector_frames(const lvk_hip_stab* stab, long long* on_device, long long* on_host);
/* last frame motion (after the trust factor) and last applied correction; each motion_height x motion_width x 2 floats */
int  lvk_hip_stab_get_meshes(const lvk_hip_stab* stab, float* motion, float* correction, int cap_floats);
/* FrameTracker::features(): (x, y, response, age) per tracked feature; returns the total count */
int  lvk_hip_stab_get_features(const lvk_hip_stab* stab, float* xy_resp_age, int cap);


/* ---- timing (reference: VideoFilter::timings() / Stopwatch::sync_gpu, Filters/VideoFilter.cpp:46-51) ---------
 * Per-stage GPU time from HIP events recorded on the launch stream around each stage's kernels. */
#define LVK_STAGE_DOWNSCALE 0   /* luma + INTER_AREA */
#define LVK_STAGE_PYRAMID   1   /* pyrDown x3 + Scharr x4 */
#define LVK_STAGE_FAST      2   /* FAST-9/16 + NMS + compaction */
#define LVK_STAGE_PYRLK     3   /* sparse optical flow */
#define LVK_STAGE_MOTION    4   /* RANSAC + local optimisation */
#define LVK_STAGE_REMAP     5   /* EASU remap of the delayed frame */
#define LVK_STAGE_INGEST    6   /* YUV420 -> packed 444 (lvk_hip_stab_push_yuv420 only) */
#define LVK_STAGE_EGRESS    7   /* packed 444 -> YUV420 */
#define LVK_STAGE_COUNT     8
/* enable: 0 = off, 1 = every stage, otherwise a set of (1 << (LVK_STAGE_x + 1)) bits -- each timed stage costs two event records
 * per frame on the host, so a measurement that needs one kernel should ask for that stage only; bits 16..23 = N: time the stages
 * of one push in N only (0 / 1 = every push), which keeps a live measurement from slowing the stream it measures */
int  lvk_hip_stab_set_profiling(lvk_hip_stab* stab, int enable);
int  lvk_hip_stab_get_profile(lvk_hip_stab* stab, double total_ms[LVK_STAGE_COUNT], long long launches[LVK_STAGE_COUNT]);

/* native_recip(x) of FSR.cl as the EASU / RCAS kernels of this library evaluate it (v_rcp_f32, what the reference's OpenCL source
 * compiles to for gfx950), elementwise over n binary32 values on the device.  OpenCL leaves native_recip implementation-defined:
 * this entry point lets a host (and the parity tests' CPU model of the kernels) read the device's definition. */
int lvk_hip_native_rcp(lvk_hip_ctx* ctx, const float* d_in, float* d_out, size_t n);

/* ---- a10: local motion estimate (vector-field preset) ---------------------------------------------------------------
 * FrameTracker::estimate_local_motions with the constraint system of generate_mesh_constraints (Vision/FrameTracker.cpp:200-321,
 * 380-457) for a mesh of cols x rows vertices: the least-squares positions of the mesh vertices from the tracked -> matched point pairs
 * (host arrays of n x 2 floats, tracking-frame coordinates), the temporal rows pulling towards the previous solution, which the
 * solver object keeps (the reference's m_OptimizedMesh; _reset zeroes it like FrameTracker::restart).  gen_region / the smoothing
 * weights are those in force when the reference (re)generates the constraints (:74-82); region / temporal_now those of the call.
 * Outputs: inlier flag per pair (L1 reprojection error < threshold) and the cols x rows x 2 normalised backward offsets of the motion
 * mesh.  Returns 0, or 2 / 3 when no estimate is possible (a point in the mesh's last cell row / column, singular system).
 * Solved on the device (normal equations, band L D L^T in binary64).  Any mesh size the remap takes (cols x rows x 8 bytes <= 64 KB,
 * up to 167 columns): meshes up to 16 columns and 2048 unknowns (16 x 64 vertices) run the register-window kernels (the 16 x 16 preset:
 * ~0.2 ms), larger or wider ones (17 x 17, 32 x 32, ...) generic kernels with the same arithmetic and the same bits (milliseconds). */
typedef struct lvk_hip_mesh_solver lvk_hip_mesh_solver;
int  lvk_hip_mesh_solver_create(lvk_hip_ctx* ctx, int cols, int rows, float gen_region_w, float gen_region_h,
                                float temporal_smoothing, float local_smoothing, int max_points, lvk_hip_mesh_solver** out);
void lvk_hip_mesh_solver_destroy(lvk_hip_mesh_solver* solver);
int  lvk_hip_mesh_solver_reset(lvk_hip_mesh_solver* solver);
int  lvk_hip_mesh_solver_solve(lvk_hip_mesh_solver* solver, const float* tracked, const float* matched, int n, float region_w, float region_h,
                               float temporal_now, float threshold, uint8_t* inliers, float* offsets);

/* ---- a3/a4: luma + INTER_AREA downscale ---------------------------------------------------------------
 * VideoFrame::viewAsFormat(GRAY) for YUV frames (= channel 0, Data/VideoFrame.cpp:260) fused with
 * cv::resize(gray, detection_resolution, INTER_AREA) (Vision/FrameTracker.cpp:117).
 * pix_stride = bytes per source pixel (3 packed 8UC3, 1 planar); d_dst is 8UC1 drows x dcols.  Any pair of sizes: integer and fractional
 * reductions, and (a frame smaller than the detection resolution) cv::resize's bilinear emulation of INTER_AREA towards a larger image. */
int lvk_hip_luma_area_resize(lvk_hip_ctx* ctx, const void* d_src, int src_step, int pix_stride, int channel,
                             int srows, int scols, void* d_dst, int dst_step, int drows, int dcols);

/* ---- a7 (pyramid): cv::pyrDown and the Scharr derivative image that cv::SparsePyrLKOpticalFlow::calc builds
 * internally (Vision/FrameTracker.cpp:140-146).  d_dst of pyr_down is ((cols+1)/2) x ((rows+1)/2) 8UC1;
 * d_dst of scharr is rows x cols x (Ix, Iy) int16, tightly packed. */
int lvk_hip_pyr_down(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step);
int lvk_hip_scharr(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst);

/* The whole optical-flow pyramid of buildOpticalFlowPyramid (levels until one would be <= the window) plus every
 * level's Scharr image, returned to the host tightly packed level after level.  Returns the level count.  Synchronous. */
int lvk_hip_build_pyramid(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols, int max_level, int win_w, int win_h,
                          uint8_t* levels, int16_t* derivs, int* level_rows, int* level_cols);

/* ---- a5 (inner): FAST-9/16 + non-max suppression per detection region -----------------------------------
 * cv::FastFeatureDetector(threshold, true, TYPE_9_16)->detect(frame(region)) (Vision/FeatureDetector.cpp:130-134).
 * regions = nregions x {x, y, w, h, threshold, active} ints; out = nregions x cap keypoints packed as
 * x | y << 12 | score << 24 (region-local, row-major like the CPU detector); counts = nregions totals.
 * Synchronous (returns after the results are on the host). */
int lvk_hip_fast_detect(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols,
                        const int* regions, int nregions, uint32_t* out, int cap, int* counts);

/* ---- a7: cv::SparsePyrLKOpticalFlow::calc(prev, next, prevPts, nextPts, status) -----------------------------
 * (Vision/FrameTracker.cpp:42-48,140-146).  Device images of the tracking resolution, host point arrays
 * (n x 2 floats), status n bytes.  Synchronous. */
int lvk_hip_pyrlk(lvk_hip_ctx* ctx, const void* d_prev, int prev_step, const void* d_next, int next_step, int rows, int cols,
                  const float* prev_pts, int n, float* next_pts, uint8_t* status,
                  int win_w, int win_h, int max_level, int max_count, double epsilon, double min_eig_threshold);

/* ---- a9: robust global motion --------------------------------------------------------------------------
 * cv::findHomography(tracked, matched, mask, UsacParams{threshold}) (full_homography != 0) or
 * cv::estimateAffinePartial2D(..., RANSAC, threshold, 50) + Homography::FromAffineMatrix (full_homography == 0)
 * as used by FrameTracker::estimate_global_motion (Vision/FrameTracker.cpp:325-375).  Host point arrays (n x 2
 * floats); H = 3x3 row-major double normalised by H22; mask = n bytes.  (region_w, region_h) = tracking resolution.
 * Returns the inlier count, or a negative value when no model exists (H = identity, mask = 0).  Synchronous.
 * The estimator is the deterministic RANSAC of DESIGN.md (OpenCV's USAC is randomised and not restated). */
int lvk_hip_estimate_global_motion(lvk_hip_ctx* ctx, const float* pts1, const float* pts2, int n, double threshold,
                                   double region_w, double region_h, int full_homography, double H[9], uint8_t* mask);

/* fast_filter (Functions/Container.tpp:97-121; call site Vision/FrameTracker.cpp:149) as the GPU runs it between the optical
 * flow and the motion estimate: keeps the pairs whose status is non-zero, in the order the reference's back-to-front swap-erase
 * leaves them.  Host arrays (n x 2 floats, n bytes); returns the number of pairs kept (>= 0) or LVK_HIP_ERR_*. */
int lvk_hip_fast_filter(lvk_hip_ctx* ctx, const float* prev, const float* matched, const uint8_t* status, int n, float* out_prev, float* out_matched);

#ifdef __cplusplus
}
#endif
#endif /* LVK_HIP_H */
