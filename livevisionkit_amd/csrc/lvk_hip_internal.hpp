// Internal definitions shared by the translation units of liblvk_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <functional>
#include <map>
#include <set>
#include <mutex>
#include <utility>

#include "lvk_hip.h"

struct LinTabEntry { int s0, s1; float a0, a1; };   // one column/row of the INTER_LINEAR mesh->frame table
struct Lin8Entry { int s0, s1, a0, a1; };            // 8-bit INTER_LINEAR table entry: two source indices, 11-bit coefficients
struct AreaTabEntry { int si; float alpha; };        // one source tap of the INTER_AREA "decimate alpha" table
struct FastRegion { int x, y, w, h, threshold, active; };   // one FAST detection region (integer ROI of the tracking frame)
constexpr int LVK_FAST_INLINE_REGIONS = 8;       // up to this many region descriptors travel as kernel arguments

constexpr int LVK_MAX_PYR_LEVELS = 8;
struct PyrLevel { const uint8_t* img; const short2* deriv; int rows, cols, step; };
struct PyrArgs { PyrLevel lv[LVK_MAX_PYR_LEVELS]; int nlevels; };

struct lvk_hip_ctx
{
    int co_blocks_per_cu = 0;               // persistent remap grid of the overlap mode: blocks per CU for the next launch (0: the default)
    int device = 0;
    int cu_count = 0;                   // compute units of the device (persistent-grid sizing of the remap)
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string last_error;

    // Pinned staging ring for small host->device parameter blocks (meshes, tables).
    static constexpr int kStageSlots = 16;
    static constexpr size_t kStageBytes = 64 * 1024;
    uint8_t* stage_host = nullptr;      // kStageSlots * kStageBytes, pinned
    uint8_t* stage_dev = nullptr;       // same size, device
    hipEvent_t stage_done[kStageSlots] = {};
    int stage_next = 0;

    // extra streams owned by objects of this context (synchronised by lvk_hip_sync as well)
    std::vector<hipStream_t> aux_streams;
    // lvk_hip_free through ANOTHER context (any thread: a frame dropped where it was last used) and lvk_hip_ctx_wait read this list while the
    // owning thread may be changing it (overlap switched on / off, transfer streams created): changes and foreign reads hold this mutex
    std::mutex aux_mutex;
    // work that objects of this context still have to enqueue before "everything is complete" can be waited for (deferred downloads):
    // (owner, hook) pairs run by lvk_hip_sync ahead of the stream synchronisations
    std::vector<std::pair<void*, std::function<int()>>> sync_hooks;

    // lvk_hip_malloc / lvk_hip_free: freed blocks are kept by size and handed out again (cv::UMat's OpenCL buffer pool plays this role
    // in the reference: Image.cpp:53,116 `dst.create` allocates nothing in steady state).  Guarded: frames may be dropped on any thread.
    std::mutex pool_mutex;
    std::multimap<size_t, void*> pool_free;
    std::map<void*, size_t> pool_sizes;            // every live block of this context -> its size
    std::set<void*> pool_cached;                   // the blocks that sit in pool_free right now (a second free of one is refused)
    size_t pool_cached_bytes = 0;
    static constexpr size_t kPoolMaxCachedBytes = (size_t)4 << 30;

    // lvk_hip_ctx_wait: events that carry "everything enqueued on the other context so far" onto this context's stream
    std::vector<hipEvent_t> wait_events;

    // Cached INTER_LINEAR tables: key = (mesh extent, frame extent, vertical?)
    std::map<std::tuple<int, int, int>, LinTabEntry*> lintabs;

    std::map<std::tuple<int, int, int>, Lin8Entry*> lin8tabs;     // 8-bit INTER_LINEAR tables (chroma upsampling)

    // Cached INTER_AREA tables: key = (source extent, destination extent)
    // max_taps: longest tap list of one destination index; span64 / span4: most source samples under 64 / 4 consecutive destination indices
    struct AreaTabDev { int2* range = nullptr; AreaTabEntry* tab = nullptr; int max_taps = 0, span64 = 0, span4 = 0; };
    std::map<std::pair<int, int>, AreaTabDev> areatabs;
    // Cached tables of the INTER_AREA ENLARGEMENT (2 taps per destination index, 11-bit fixed point): key = (source extent, destination extent)
    std::map<std::pair<int, int>, int4*> enlargetabs;        // per destination index: (s0, s1, w0, w1)

    // (a runtime error is reported HERE: the runtime's sticky "last error" is cleared with it, so that the hipGetLastError() check behind a later, perfectly
    //  good launch does not report it a second time -- round 6, the failure of one call made the next valid push fail as well)
    int fail(int code, const std::string& msg) { last_error = msg; if (code == LVK_HIP_ERR_RUNTIME) (void)hipGetLastError(); return code; }
};

// The current device is a per-thread setting of the HIP runtime, and events / streams / allocations are made on it: an entry point that may
// be called from any host thread (one filter per source, each on whatever thread the host gives it -- VisionFilter.cpp:157-162) makes its
// context's device current for the duration of the call and puts the caller's back.  One hipGetDevice (a thread-local read) when they match.
struct lvk_device_guard
{
    int prev = -1; bool switched = false;
    explicit lvk_device_guard(const lvk_hip_ctx* ctx)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~lvk_device_guard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    lvk_device_guard(const lvk_device_guard&) = delete;
    lvk_device_guard& operator=(const lvk_device_guard&) = delete;
};

// First statement of every public entry point that takes a context and touches the HIP runtime: NULL check + device guard
#define LVK_HIP_ENTRY(ctx)                         \
    if (!(ctx)) return LVK_HIP_ERR_ARG;            \
    lvk_device_guard lvk_entry_device_guard(ctx)

#define LVK_HIP_CHECK(ctx, expr)                                                                      \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
        {                                                                                             \
            (void)hipGetLastError();   /* reported HERE: not again by the hipGetLastError() behind a later, successful launch */ \
            return (ctx)->fail(LVK_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                                                             \
    } while (0)

// First statement of the tracker's small, latency-bound kernels: their waves take instruction-issue priority over the waves of the
// VALU-bound bulk kernels (remap) that share the SIMDs in overlap mode (s_setprio; no effect when they run alone).
#ifndef LVK_TRACKER_PRIO
#define LVK_TRACKER_PRIO 3
#endif
#define LVK_TRACKER_PRIORITY() __builtin_amdgcn_s_setprio(LVK_TRACKER_PRIO)

// Instrumented kernels (in-kernel clocks, printf) never go into the product library: the defines below are only accepted together with
// -DLVK_PROBE_BUILD, which scripts/variant_build.sh passes for the variants under livevisionkit_amd/variants/ (round-4 ADVICE).
// (the same for kernels that are NOT bit-exact: LVK_EASU_TOLERANT, the tolerance-mode A / B partner of the remap -- remap_core.hpp)
#if (defined(LVK_TIMELINE) || defined(LVK_RANSAC_TIMING) || defined(LVK_MESH_TIMING) || defined(LVK_EASU_TOLERANT)) && !defined(LVK_PROBE_BUILD)
#error "instrumented build: pass -DLVK_PROBE_BUILD (scripts/variant_build.sh); never the library the tests and the bench load by default"
#endif

// Debug builds with -DLVK_TIMELINE (scripts/variant_build.sh timeline -DLVK_TIMELINE): every instrumented kernel logs the wall clock (100 MHz) at which its
// first block started and its last block finished into a per-translation-unit ring, read back by lvk_tl_read_<unit>().  rocprofv3's
// kernel trace slows the host enough to change how the two streams overlap; this does not.  Expands to nothing in product builds.
#ifdef LVK_TIMELINE
#define LVK_TL_SLOTS 4
#define LVK_TL_RING 8192
static __device__ unsigned lvk_tl_head[LVK_TL_SLOTS];
static __device__ unsigned long long lvk_tl_log[LVK_TL_SLOTS][LVK_TL_RING][2];
struct LvkTimelineScope
{
    int slot; bool sampled; unsigned long long t0;
    __device__ explicit LvkTimelineScope(int s) : slot(s), sampled(false), t0(0)
    {
        // one workgroup in 64 (plus the first and the last) appends its own (start, end) record; the reader groups records into launches
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        sampled = threadIdx.x == 0 && threadIdx.y == 0 && ((b & 63u) == 63u || b == 0 || b == total - 1);
        if (sampled) t0 = (unsigned long long)wall_clock64();
    }
    __device__ ~LvkTimelineScope()
    {
        if (sampled)
        {
            const unsigned i = atomicAdd(&lvk_tl_head[slot], 1u) % LVK_TL_RING;
            lvk_tl_log[slot][i][0] = t0; lvk_tl_log[slot][i][1] = (unsigned long long)wall_clock64();
        }
    }
};
#define LVK_TL(slot) LvkTimelineScope lvk_tl_scope_(slot)
// out: LVK_TL_SLOTS x (count, then LVK_TL_RING x (start, end)) int64
#define LVK_TL_EXPORT(unit)                                                                                     \
    extern "C" int lvk_tl_read_##unit(long long* out)                                                           \
    {                                                                                                           \
        unsigned idx[LVK_TL_SLOTS];                                                                             \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                                    \
        if (hipMemcpyFromSymbol(idx, HIP_SYMBOL(lvk_tl_head), sizeof(idx)) != hipSuccess) return -1;           \
        for (int s = 0; s < LVK_TL_SLOTS; s++) out[(size_t)s * (1 + 2 * LVK_TL_RING)] = idx[s];                 \
        for (int s = 0; s < LVK_TL_SLOTS; s++)                                                                  \
            if (hipMemcpyFromSymbol(out + (size_t)s * (1 + 2 * LVK_TL_RING) + 1, HIP_SYMBOL(lvk_tl_log), sizeof(long long) * 2 * LVK_TL_RING, \
                                    sizeof(long long) * 2 * LVK_TL_RING * s) != hipSuccess) return -1;         \
        return 0;                                                                                               \
    }
#else
#define LVK_TL(slot) do { } while (0)
#define LVK_TL_EXPORT(unit)
#endif

#define LVK_HIP_REQUIRE(ctx, cond)                                                                    \
    do { if (!(cond)) return (ctx)->fail(LVK_HIP_ERR_ARG, "pre-condition failed: " #cond); } while (0)

// Copies `bytes` (<= kStageBytes) of host data into a device staging slot, asynchronously on the
// context's stream, and returns the device address.  The slot is recycled after kStageSlots uses.
// With `slot` != nullptr the slot is NOT marked consumed by the copy: the caller launches the kernel that reads it and then calls
// lvk_stage_consumed(ctx, *slot, stream), so that a reuse of the slot (from either stream of a stabilizer) waits for that kernel.
int lvk_stage_params(lvk_hip_ctx* ctx, hipStream_t stream, const void* host, size_t bytes, void** d_out, int* slot = nullptr);
int lvk_stage_consumed(lvk_hip_ctx* ctx, int slot, hipStream_t stream);

// Device-resident INTER_LINEAR table for resizing a mesh axis of `msize` vertices to `fsize` pixels.
int lvk_get_lintab(lvk_hip_ctx* ctx, int msize, int fsize, bool vertical, const LinTabEntry** d_out);

// Device-resident INTER_AREA table (per destination index: [start, count) into the tap list).
int lvk_get_areatab(lvk_hip_ctx* ctx, int ssize, int dsize, const int2** d_range, const AreaTabEntry** d_tab, int* max_taps = nullptr, int* span64 = nullptr, int* span4 = nullptr);

// Asynchronous launches on ctx->stream (device pointers).
int lvk_launch_luma_area_resize(lvk_hip_ctx* ctx, const void* d_src, int src_step, int pix_stride, int channel,
                                int srows, int scols, void* d_dst, int dst_step, int drows, int dcols);
int lvk_launch_pyr_down(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step);
int lvk_launch_scharr(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst);
struct PyrArgs;
int lvk_launch_pyramid(lvk_hip_ctx* ctx, const PyrArgs& args, bool derivs = false);   // derivs: also fill the Scharr images (test entry only)

// FAST-9/16 + NMS per region (fast.hip)
int lvk_fast_workspace_bytes(int nregions, int max_rw, int max_rh, size_t* masks_bytes, size_t* scores_bytes);
int lvk_launch_fast(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols,
                    const FastRegion* d_regions, int nregions, int max_rw, int max_rh,
                    void* d_masks, void* d_scores, uint32_t* d_out, int cap, int* d_counts, const FastRegion* host_regions = nullptr);

// ... with the suppression grid on the device (k_fast_detect's per-cell slots + k_fast_insert): the new features land behind the held ones in `pts`
struct FastInsertDesc
{
    const uint16_t* col_of; const uint32_t* row_base; const uint8_t* bucket;      // device copies of FeatureGridH's tables
    uint32_t* cell_first; void* cell_best /* 8 bytes per cell */; int* region_count;                 // device: one slot pair per cell, one counter per region (lvk_fast_cells_reset once; the kernels leave them cleared)
    const uint32_t* occ;                   // host memory: one bit per cell that holds a propagated feature (travels as a kernel argument)
    const int* occ_bucket;                 // host memory: 16 counts, the propagated cells per distribution bucket
    int capacity; bool small_grid; int n_held, min_samples; float uniformity, homography_threshold;
    float2* pts; uint32_t* new_kp; int* result; int* d_n; int* d_full; int* counts;
};
int lvk_launch_fast_insert(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols, const FastRegion* host_regions, int nregions,
                           int max_rw, int max_rh, void* d_masks, void* d_scores, const FastInsertDesc& d);
bool lvk_fast_insert_fits(int cells, int nregions, int max_rw, int max_rh, int cols, int rows);      // what the kernels cover (else: the host loop)
int lvk_fast_cells_reset(lvk_hip_ctx* ctx, uint32_t* d_first, void* d_best, int cells, int* d_region_count /* LVK_FAST_INLINE_REGIONS ints */);

// Pyramidal LK (pyrlk.hip)
struct LensModel;
int lvk_pyramid_geometry(int rows, int cols, int max_level, int win_w, int win_h, int* lrows, int* lcols);
int lvk_launch_pyrlk(lvk_hip_ctx* ctx, const PyrArgs& prev, const PyrArgs& next, const float2* d_prev_pts, int n,
                     float2* d_next_pts, uint8_t* d_status, int win_w, int win_h, int max_count, double epsilon, double min_eig,
                     float2* d_prev_copy = nullptr,      // pts may be device-visible host memory; d_prev_copy receives a device copy
                     const LensModel* lens = nullptr, double lens_sx = 0.0, double lens_sy = 0.0, float2* d_und = nullptr,
                     const int* d_n = nullptr);         // d_n: the point count lives on the device (n = the launch's upper bound)
                     // lens + d_und: the flow kernel also writes the lens-corrected (previous | matched) positions, 2 n entries (fused lens mode)

// Image pyramid + Scharr derivative images of one tracking frame, resident in HBM.
struct DevicePyramid
{
    uint8_t* img_base = nullptr;
    uint8_t* deriv_base = nullptr;
    PyrArgs args{};
    int allocate(lvk_hip_ctx* ctx, int rows, int cols, int max_level, int win_w, int win_h);
    int build(lvk_hip_ctx* ctx, bool derivs = false);   // level 0 image must be filled; enqueues pyrDown (+ the Scharr images on request)
    void release();
};

// "The chain's results are in host memory": the LAST kernel of the tracker's chain stores `seq` into a word of pinned host memory after its
// result stores (system-scope release), and a host that waits for every frame spins on that word instead of waiting for the kernel's
// completion signal -- the results are there ~3 us before the runtime reports the kernel as finished (teardown, signal, wake-up).
// flag == nullptr: no signal.
struct LvkHostSignal { unsigned* flag; unsigned seq; };

// Robust global motion (motion.hip)
size_t lvk_ransac_workspace_bytes(int n);
int lvk_launch_ransac(lvk_hip_ctx* ctx, const float2* d_p1, const float2* d_p2, int n, double threshold, double region_w, double region_h,
                      bool full_homography, void* d_ws, double* d_H, int* d_ninl, uint8_t* d_mask, const int* d_n = nullptr,
                      const int* d_full = nullptr,       // d_full: the model choice lives on the device (full_homography ignored)
                      LvkHostSignal done = LvkHostSignal{nullptr, 0});
// fast_filter (Functions/Container.tpp:97-121) of the optical-flow result on the GPU: compacts (prev, matched) by `status` into
// (d_p1, d_p2) in exactly the order the host's back-to-front swap-erase produces, writes the count to d_count, and mirrors the raw
// matched points / status flags into device-visible host memory for the host's own bookkeeping.
int lvk_launch_match_compact(lvk_hip_ctx* ctx, const float2* d_prev, const float2* d_matched, const uint8_t* d_status, int n,
                             float2* d_p1, float2* d_p2, int* d_count, int* h_count, float2* h_matched, uint8_t* h_status,
                             const float2* d_und = nullptr, float region_w = 0.0f, float region_h = 0.0f,   // d_und: lens-corrected (prev | matched), see k_match_compact
                             const int* d_n_raw = nullptr);                                                    // the raw count lives on the device (n: upper bound)
// both in two kernels (the hypotheses kernel compacts the flow result itself): same results; n <= 2048 pairs
constexpr int LVK_COMPACT_RANSAC_MAX = 2048;
int lvk_launch_compact_ransac(lvk_hip_ctx* ctx, const float2* d_prev, const float2* d_matched, const uint8_t* d_status, int n,
                              float2* d_p1, float2* d_p2, int* d_count, int* h_count, float2* h_matched, uint8_t* h_status,
                              const float2* d_und, float region_wf, float region_hf,
                              double threshold, double region_w, double region_h, bool full_homography, void* d_ws, double* d_H, int* d_ninl, uint8_t* d_mask,
                              const int* d_n_raw = nullptr, const int* d_full = nullptr, LvkHostSignal done = LvkHostSignal{nullptr, 0});

struct LensArgs;
// Dense remap on an explicit stream (remap.hip)
int lvk_launch_remap_homography(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int src_rows, int src_cols,
                                void* d_dst, int dst_step, int dst_rows, int dst_cols, int off_x, int off_y,
                                const float H[9], const uint8_t bg[3], int yuv, const LensArgs* lens = nullptr, bool co_scheduled = false);
int lvk_launch_remap_mesh(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int src_rows, int src_cols,
                          void* d_dst, int dst_step, const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv,
                          const LensArgs* lens = nullptr, bool co_scheduled = false);
int lvk_launch_warpmesh_apply(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                              void* d_dst, int dst_step, const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv);

// YUV420 <-> packed 444 (ingest.hip)
int lvk_launch_ingest_yuv420(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_y, int y_step, const void* d_u, int u_step,
                             const void* d_v, int v_step, int nv12, int rows, int cols, void* d_dst, int dst_step);
int lvk_launch_egress_yuv420(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                             void* d_y, int y_step, void* d_u, int u_step, void* d_v, int v_step, int nv12);
int lvk_launch_ingest_obs(lvk_hip_ctx* ctx, hipStream_t stream, int video_format, const void* const d_planes[3], const int steps[3],
                          int rows, int cols, void* d_dst, int dst_step);
int lvk_launch_egress_obs(lvk_hip_ctx* ctx, hipStream_t stream, int video_format, const void* d_src, int src_step, int rows, int cols,
                          void* const d_planes[3], const int steps[3]);


int lvk_launch_remap_map(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                         void* d_dst, int dst_step, const void* d_map, int map_step, const uint8_t bg[3], int yuv);

// Fused lens pre-warp (lens.hip): the camera profile reduced to what the closed-form map needs for one frame size.
// d = nfx, nfy, ncx, ncy (new camera matrix), fx, fy, cx, cy, k1, k2, p1, p2, k3, kxc, vxc, kyc, vyc (crop_in term, pixels);
// f = the binary32 kernel parameters: 1/nfx, 1/nfy, then d[2..16] rounded.
struct LensModel { double d[17]; float f[17]; int view[4]; };
// F^-1 of the fused lens map for one tracked point, binary64 (two passes of {remove the crop_in term, the 5 fixed-point iterations of
// cv::undistortPoints, apply P}); the points live at tracking resolution (sx, sy: frame / tracking size), the model at frame resolution.
struct LensModelD { double d[17]; };
__device__ __forceinline__ float2 lvk_lens_undistort_point(const LensModelD& M, double sx, double sy, float2 p)
{
    const double nfx = M.d[0], nfy = M.d[1], ncx = M.d[2], ncy = M.d[3], fx = M.d[4], fy = M.d[5], cx = M.d[6], cy = M.d[7];
    const double k1 = M.d[8], k2 = M.d[9], p1 = M.d[10], p2 = M.d[11], k3 = M.d[12];
    const double kxc = M.d[13], vxc = M.d[14], kyc = M.d[15], vyc = M.d[16];
    const double s = (double)p.x * sx, t = (double)p.y * sy;
    double u = s, v = t;
    for (int pass = 0; pass < 2; pass++)
    {
        const double s1 = s - (u * kxc + vxc), t1 = t - (v * kyc + vyc);
        const double x0 = (s1 - cx) / fx, y0 = (t1 - cy) / fy;
        double x = x0, y = y0;
        for (int j = 0; j < 5; j++)
        {
            const double r2 = x * x + y * y;
            const double icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            if (icdist < 0) { x = x0; y = y0; break; }
            const double dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            const double dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - dX) * icdist;
            y = (y0 - dY) * icdist;
        }
        u = x * nfx + ncx; v = y * nfy + ncy;
    }
    return make_float2((float)(u / sx), (float)(v / sy));
}
struct LensArgs { float f[17]; };
int lvk_lens_model_build(const lvk_camera_params& params, int rows, int cols, LensModel& out);
// (a | b)[i] raw tracking-frame points -> lens-corrected positions, binary64, written to out[0 .. na + nb)
int lvk_launch_lens_undistort(lvk_hip_ctx* ctx, hipStream_t stream, const LensModel& model, double sx, double sy,
                              const float2* a, int na, const float2* b, int nb, float2* out);
// lens != nullptr composes the closed-form lens map into the coordinate (fused mode)
int lvk_launch_warpmesh_apply_lens(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                                   void* d_dst, int dst_step, const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv,
                                   const LensArgs* lens, bool co_scheduled = false);   // co_scheduled: occupancy-capped kernels for overlap mode

// Debug overlays (draw.hip)
int lvk_launch_draw_grid(lvk_hip_ctx* ctx, hipStream_t stream, void* d_dst, int dst_step, int rows, int cols, int grid_w, int grid_h,
                         const uint8_t colour[3], int thickness);
int lvk_launch_draw_crosses(lvk_hip_ctx* ctx, hipStream_t stream, void* d_dst, int dst_step, int rows, int cols, const float* pts, int n,
                            float scale_x, float scale_y, const uint8_t colour[3], int cross_size, int thickness);

// ScalingFilter's two kernels: EASU upscale (remap.hip) and RCAS (sharpen.hip)
int lvk_launch_upscale(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int src_rows, int src_cols,
                       void* d_dst, int dst_step, int dst_rows, int dst_cols, int yuv);
int lvk_launch_sharpen(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                       void* d_dst, int dst_step, float sharpness);

// Local motion estimate on the device (mesh.hip): the least-squares mesh of FrameTracker::estimate_local_motions
struct lvk_mesh_solver_dev;
int lvk_mesh_solver_create(lvk_hip_ctx* ctx, int cols, int rows, float gen_w, float gen_h, float temporal, float local, lvk_mesh_solver_dev** out);
void lvk_mesh_solver_free(lvk_mesh_solver_dev* s);
int lvk_mesh_solver_reset(lvk_mesh_solver_dev* s, hipStream_t stream);
int lvk_mesh_solver_cols(const lvk_mesh_solver_dev* s);
int lvk_mesh_solver_rows(const lvk_mesh_solver_dev* s);
int lvk_launch_mesh_solve(lvk_mesh_solver_dev* s, hipStream_t stream, void* d_scratch /* 32 bytes per pair */, const float2* d_p1, const float2* d_p2, const int* d_count, int n_pts,
                          int min_samples, float region_w, float region_h, float temporal_now, float threshold,
                          float* h_offsets, uint8_t* h_mask, int* h_status);

// remap + 4:2:0 egress in one kernel (remap.hip)
int lvk_launch_warpmesh_apply_420(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                                  void* o_y, int oy_step, void* o_u, int ou_step, void* o_v, int ov_step, int nv12,
                                  const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], const LensArgs* lens,
                                  bool co_scheduled);
bool lvk_remap_obs_fusable(int video_format);
int lvk_launch_warpmesh_apply_obs(lvk_hip_ctx* ctx, hipStream_t stream, int video_format, const void* d_src, int src_step, int rows, int cols,
                                  void* const planes[3], const int steps[3], const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3],
                                  const LensArgs* lens, bool co);          // co_scheduled: the persistent grid of the overlap mode
