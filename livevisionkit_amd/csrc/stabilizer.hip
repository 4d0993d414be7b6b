// The stabilization filter proper: per-frame orchestration of the HIP kernels plus the host state machines.
//
// Replaces (reference, paths relative to LiveVisionKit/):
//   StabilizationFilter::{configure,filter,restart,ready,reset_context,frame_delay,stable_region}  Filters/StabilizationFilter.cpp:42-206
//   FrameTracker::{configure,track,restart,estimate_global_motion}                                  Vision/FrameTracker.cpp:57-196,325-375
//   FeatureDetector::{detect,propagate,reset} bookkeeping                                           Vision/FeatureDetector.cpp:114-214
//
// Per frame, on the context's stream:
//   luma (Y / BGR / RGB -> gray) + INTER_AREA downscale -> pyramid (pyrDown x3, Scharr x4)   [imgproc.hip]
//   FAST-9/16 + NMS per due region, ordered compaction                                  [fast.hip]      -> host: suppression grid
//   pyramidal LK, one block per feature, one wave per level (points read from pinned host memory) [pyrlk.hip]
//   fast_filter in the reference's swap-erase order                                     [motion.hip k_match_compact]
//   RANSAC hypotheses + local optimisation, pair count from the device                  [motion.hip]    -> ONE sync; host: ageing, propagate, QA, smoothing
//   (vector-field preset / lens modes: sync after LK, point filter + mesh solve on the host)
// and on the bulk stream (overlap mode; else the same stream):
//   4:2:0 ingest of the new frame, EASU remap of the delayed frame (homography or in-kernel mesh, optionally with the lens
//   pre-warp composed in, optionally with the 4:2:0 egress fused)                         [ingest.hip, remap.hip]
// Packed frames are never copied: the filter borrows the caller's device buffer until that frame has been emitted
// (the reference moves the input frame into its queue, StabilizationFilter.cpp:118).
#include "lvk_hip_internal.hpp"
#include "host_logic.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>

using lvkh::Feature;
using lvkh::WarpMeshF;

namespace {

constexpr int LK_WIN = 11, LK_LEVELS = 3, LK_ITERS = 5;        // FrameTracker.cpp:33-35
constexpr double LK_EPS = 0.01, LK_MIN_EIG = 1e-4;
constexpr float HOMOGRAPHY_DISTRIBUTION_THRESHOLD = 0.6f;      // FrameTracker.cpp:37
constexpr float QA_UPDATE_RATE = 0.1f, QA_BLEND_STEP = 0.05f;   // StabilizationFilter.cpp:30-31

inline float step_toward(float current, float target, float amount)   // Functions/Math.tpp:133-142
{
    return current > target ? std::max(current - amount, target) : std::min(current + amount, target);
}

// Host-side wall-clock trace of one push (LVK_HIP_HOST_TRACE=1: summary on stderr at destroy) -- development aid
struct HostTrace
{
    enum { ENTER, DOWN_PYR_LAUNCH, FAST_SYNC, GRID, LK_LAUNCH, LK_SYNC, FILTER, RANSAC_LAUNCH, RANSAC_SYNC, POST, SMOOTH, REMAP_LAUNCH, EXIT, EMIT_WAITS, EMIT_KERNEL, EMIT_EVENT, EXIT_PRE, EXIT_WAIT, N };
    bool on = std::getenv("LVK_HIP_HOST_TRACE") != nullptr;
    double acc[N] = {0}; long cnt[N] = {0};
    std::chrono::steady_clock::time_point last;
    void begin() { if (on) last = std::chrono::steady_clock::now(); }
    void mark(int k)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        acc[k] += std::chrono::duration<double, std::micro>(now - last).count(); cnt[k]++; last = now;
    }
    void dump() const
    {
        if (!on) return;
        static const char* names[N] = {"enter", "downscale+pyramid launch", "fast launch+sync", "grid (host)", "lk upload+launch", "lk sync", "filter (host)",
                                       "ransac upload+launch", "ransac sync", "post (host)", "qa+smoother (host)", "remap launch", "exit",
                                       "  emit: stream waits", "  emit: remap kernel launch", "  emit: slot event record", "  exit: up to the conversion wait", "  exit: conversion wait"};
        double total = 0; long frames = cnt[DOWN_PYR_LAUNCH] ? cnt[DOWN_PYR_LAUNCH] : 1;
        for (int i = 0; i < N; i++) total += acc[i];
        std::fprintf(stderr, "[lvk host trace] %ld frames, %.1f us/frame inside push\n", frames, total / frames);
        for (int i = 0; i < N; i++) if (cnt[i]) std::fprintf(stderr, "  %-28s %8.1f us/frame (%ld marks)\n", names[i], acc[i] / frames, cnt[i]);
    }
};

struct QueuedFrame { const void* d_ptr; int step, rows, cols; uint64_t ts; int format; };

} // namespace

struct lvk_hip_stab
{
    lvk_hip_ctx* ctx = nullptr;
    lvk_stab_settings s{};
    bool configured = false;
    bool buffers_ok = false;                   // the tracker's buffers match the committed settings (false after a failed allocation)

    // ---- tracker device state
    DevicePyramid pyr[2];
    int cur = 0;                               // pyr[cur] = current frame, pyr[cur ^ 1] = previous frame
    int pyr_w = 0, pyr_h = 0;                  // resolution the pyramids are allocated for
    int prev_w = 0, prev_h = 0, cur_w = 0, cur_h = 0;
    bool initialized = false;
    size_t cap_features = 0;                   // suppression-grid capacity (max features)
    int fast_cap = 0, fast_max_rw = 0, fast_max_rh = 0, fast_regions = 0;
    void* d_fast_masks = nullptr; void* d_fast_scores = nullptr;
    float2 *d_pts = nullptr, *d_matched = nullptr, *d_p1 = nullptr; uint8_t* d_status = nullptr;      // d_p1: 2 * cap_features pairs (p1 | p2)
    void* d_ransac_ws = nullptr;
    int* d_count = nullptr;                    // number of matches after the GPU-side fast_filter
    // the suppression grid on the device (fast.hip k_fast_insert): the grid's tables, the point count / model choice the chain's kernels read,
    // and (pinned) which cells hold propagated features, the new features and the kernel's verdicts
    uint16_t* d_grid_col = nullptr; uint32_t* d_grid_row = nullptr; uint8_t* d_grid_bucket = nullptr;
    uint32_t* d_cell_first = nullptr; void* d_cell_best = nullptr; int* d_region_count = nullptr;      // per-cell slots / per-region counters the detector folds its corners into
    int* d_n_points = nullptr; int* d_full = nullptr;
    uint32_t* h_occ = nullptr; uint32_t* h_new_kp = nullptr; int* h_insert = nullptr;
    bool device_grid = [] { const char* e = std::getenv("LVK_HIP_HOST_GRID"); return !(e && e[0] == '1'); }();      // LVK_HIP_HOST_GRID=1: the host loop (A/B, tests)
    long device_grid_frames = 0, host_grid_frames = 0;
    float2* d_und = nullptr;                   // fused lens mode, chained path: lens-corrected (previous | matched) positions
    // pinned host mirrors
    uint32_t* h_fast_out = nullptr; int* h_fast_counts = nullptr; FastRegion* h_regions = nullptr;
    float2 *h_pts = nullptr, *h_matched = nullptr, *h_p1 = nullptr; uint8_t* h_status = nullptr;
    double* h_H = nullptr; int* h_ninl = nullptr; uint8_t* h_mask = nullptr;
    int* h_count = nullptr;                    // d_count as the GPU-side fast_filter reported it (checked against the host's own)
    float2* h_und = nullptr;                   // fused lens mode: lens-corrected (previous | matched) point positions

    // ---- fused lens pre-warp (lvk_hip_stab_set_lens): model of the current frame size
    bool lens = false;
    lvk_camera_params lens_params{};
    LensModel lens_model{}; LensArgs lens_args{};
    int lens_rows = 0, lens_cols = 0;
    int ensure_lens(int rows, int cols)
    {
        if (!lens || (rows == lens_rows && cols == lens_cols)) return LVK_HIP_OK;
        if (lvk_lens_model_build(lens_params, rows, cols, lens_model) != LVK_HIP_OK) return fail(LVK_HIP_ERR_ARG, "invalid camera profile for this frame size");
        std::memcpy(lens_args.f, lens_model.f, sizeof(lens_args.f));
        lens_rows = rows; lens_cols = cols;
        return LVK_HIP_OK;
    }

    HostTrace trace;

    // ---- host state
    lvkh::FeatureGridH grid;
    lvkh::PathSmootherH smoother;
    // FrameTracker's m_MeshConstraints + m_OptimizedMesh live on the device (mesh.hip); the parameters the constraints were generated with:
    struct MeshGen { int cols = 0, rows = 0; float w = 0, h = 0, temporal = 0, local = 0; } mesh_gen;
    lvk_mesh_solver_dev* mesh_dev = nullptr;
    void* d_mesh_scratch = nullptr; float* h_offsets = nullptr; int* h_mesh_status = nullptr; size_t h_offsets_floats = 0;
    lvk_stab_settings tracker_s{};             // FrameTracker::m_Settings (what the tracker was last configured with)
    std::vector<Feature> tracked;
    std::vector<FastRegion> plan;
    std::deque<QueuedFrame> queue;
    size_t queue_capacity = 1;
    float tracking_stability = 0.0f, scene_quality = 0.0f, trust = 0.0f;
    // taps for stats / tests
    float last_distribution = 0.0f; int last_detected = 0, last_matched = 0;
    double last_H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    WarpMeshF last_motion, last_correction;

    // ---- optional overlap of the output remap with the next frame's tracking (second stream)
    bool overlap = false;
    hipStream_t remap_stream = nullptr;
    bool remap_stream_owned = false;             // created by lvk_hip_stab_set_overlap (else: the caller's, lvk_hip_stab_set_bulk_context)
    hipEvent_t remap_done[2] = {nullptr, nullptr};
    int remap_slot = 0;
    const void* pending_release = nullptr;     // frame whose remap is still in flight on remap_stream
    bool pool_frames = false;                  // the queued frames are pool slots that only stream-ordered kernels of remap_stream touch
    int queue_kind = 0;                        // who owns the queued frames: 0 = queue empty, 1 = borrowed from the caller, 2 = pool slots
    std::function<int()> deferred_ingest;      // the newest frame's 4:2:0 conversion, not yet launched (see lvk_hip_stab_push_yuv420)
    int run_deferred_ingest() { auto f = std::move(deferred_ingest); deferred_ingest = nullptr; return f ? f() : LVK_HIP_OK; }
    hipEvent_t ingest_done = nullptr;          // 4:2:0 ingest of the newest frame
    // Overlap mode with a frame delay: the conversion runs on the TRACKING stream, in the slot that stream has free between the last
    // kernel of a frame's chain and the first of the next frame's (the host's turn: ~25 us) -- behind an event the push waits on instead
    // of the whole stream.  On the bulk stream it sat between two remaps: 13 us + a kernel boundary of every bulk-stream period, which
    // bounds the frame rate.  The pool slot it writes was last read by a remap on the bulk stream: one event per slot orders the two.
    hipEvent_t chain_done = nullptr;
    bool ingest_on_tracker = false, tracker_ingest_capable = false;
    bool bulk_busy_at_push = false;            // the previous remap was still running when this push began
    // a free-running caller: the bulk stream still busy, or this push began within 15 us of the previous one's return (a caller that waits
    // for its frames synchronises and reads back in between: at least a remap's duration)
    bool caller_runs_free = false;
    std::chrono::steady_clock::time_point last_push_end{};
    // tests: LVK_HIP_INGEST_PLACEMENT=tracker|bulk pins the placement that is otherwise decided per push (see track())
    int ingest_placement = [] { const char* e = std::getenv("LVK_HIP_INGEST_PLACEMENT"); return !e ? 0 : (e[0] == 't' ? 1 : (e[0] == 'b' ? 2 : 0)); }();
    std::vector<hipEvent_t> slot_read_done;    // parallel to pool_all: the remap that read the slot (recorded on the bulk stream), or nullptr
    std::vector<char> slot_read_armed;
    int slot_index(const void* p) const { for (size_t i = 0; i < pool_all.size(); i++) if (pool_all[i] == p) return (int)i; return -1; }
    int pending_slot = -1;
    // Overlap mode: what the caller enqueued on the context's stream before a push (a decode / copy that fills the frame or the planes)
    // must be visible to the kernels of the bulk stream that read it.  The event is recorded when the push starts -- before the tracker's
    // own kernels, so that the bulk stream never waits for those -- and the bulk stream waits for it ahead of its first launch of the push.
    hipEvent_t caller_ready = nullptr;
    bool caller_wait_pending = false;
    int mark_caller_work()
    {
        if (!(overlap && s.stabilize_output && remap_stream)) return LVK_HIP_OK;
        // nothing pending on the context's stream (the steady state of a caller whose frames are already resident): nothing to order
        if (hipStreamQuery(ctx->stream) == hipSuccess) { caller_wait_pending = false; return LVK_HIP_OK; }
        (void)hipGetLastError();
        if (!caller_ready) LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&caller_ready, hipEventDisableTiming));
        LVK_HIP_CHECK(ctx, hipEventRecord(caller_ready, ctx->stream));
        caller_wait_pending = true;
        return LVK_HIP_OK;
    }
    int bulk_stream_sees_caller_work()
    {
        if (!caller_wait_pending) return LVK_HIP_OK;
        caller_wait_pending = false;
        LVK_HIP_CHECK(ctx, hipStreamWaitEvent(remap_stream, caller_ready, 0));
        return LVK_HIP_OK;
    }
    // Borrowed frames that left the queue outside a push (queue shrunk by configure(), overlap / stabilize_output toggled while a remap
    // was pending): handed back through *released by the following pushes, one per push.
    std::deque<const void*> orphaned;

    // ---- optional per-stage GPU timing (HIP events on the launch stream)
    bool profiling = false;
    unsigned prof_mask = ~0u;                  // stages that are timed while profiling is on (bit = LVK_STAGE_*)
    unsigned prof_every = 1, prof_tick = 0;    // time the stages of one push in `prof_every` (the event records cost host time per frame)
    struct EvPair { hipEvent_t a, b; int kind; };
    std::vector<EvPair> ev_pool; size_t ev_used = 0;
    double prof_ms[LVK_STAGE_COUNT] = {0}; long prof_n[LVK_STAGE_COUNT] = {0};
    int prof_begin(int kind, hipStream_t stream = nullptr);
    void prof_end(int idx, hipStream_t stream = nullptr);
    int prof_collect();

    int fail(int code, const std::string& msg) { return ctx->fail(code, msg); }
    void free_tracker_buffers();
    int alloc_tracker_buffers();
    int alloc_pyramids();
    int configure(const lvk_stab_settings& st);
    void tracker_restart();
    void reset_context() { tracker_restart(); smoother.restart(); }
    // Chained path: the host's own fast_filter pass, the ageing of the features and the re-seeding of the suppression grid are not
    // needed to launch the remap (the motion estimate, the match count and the inlier mask come from the GPU): they run after the
    // launch, before the push returns.  post_n >= 0: pending for a frame with post_n tracked points / post_m matches.
    int post_n = -1, post_m = 0;
    bool post_error = false;
    void finish_post();
    int track(const QueuedFrame& f, const void* luma, int luma_step, int luma_pix, int luma_channel, WarpMeshF& motion, bool& have_motion);

    // ---- host-resident frames (lvk_hip_stab_push_yuv420_host): the transfers either side of the 4:2:0 path.  One copy stream per
    // direction (scripts/pcie_probe.hip, profiles/r03_pcie_probe.txt: ONE copy engine stream each way moves 46.8 GB/s each way at once,
    // two per direction fall to 31), the luma plane first so that the tracker starts while the chroma planes are still on the link.
    struct HostIO
    {
        static constexpr int K_IN = 2, K_OUT = 3;
        hipStream_t up = nullptr, up2 = nullptr, down = nullptr, down2 = nullptr;
        struct Pending { bool valid = false; int slot = 0; void* y; void* u; void* v; int ys, us, vs, nv12; } pending;      // a download not yet handed to the copy engine
        int rows = 0, cols = 0;
        void* d_in[K_IN] = {nullptr, nullptr}; void* d_out[K_OUT] = {nullptr, nullptr, nullptr};      // contiguous planes: Y | U | V  (or Y | UV)
        hipEvent_t y_done[K_IN] = {}, c_done[K_IN] = {}, out_ready[K_OUT] = {}, down_done[K_OUT] = {};
        bool down_armed[K_OUT] = {false, false, false};
        bool y_is_c[K_IN] = {false, false};                      // the slot's frame came as one copy: c_done covers the luma plane too
        // look-ahead (lvk_hip_stab_prefetch_yuv420_host): the planes whose upload is already under way, and the slot they go to
        struct Ahead { int slot; const void* key[3]; int rows, cols, nv12; };
        std::deque<Ahead> ahead;                                // in upload order; a push consumes the oldest
        const uint8_t* last_dst_lo = nullptr; const uint8_t* last_dst_hi = nullptr;      // luma plane of the newest download's destination
        int in_next = 0, out_next = 0, last_down = -1;       // last_down: slot of the newest download (its event orders a later direct write behind it)
        std::chrono::steady_clock::time_point last_end{};      // when the previous host push returned
    } hostio;
    bool host_free_running_hint = false;                 // lvk_hip_stab_push_yuv420_host's own finding, for the push it wraps
    bool host_direct_now = false;                        // the push being wrapped writes its output planes straight into host memory
    hipEvent_t ingest_wait[2] = {nullptr, nullptr};      // events the newest frame's 4:2:0 conversion waits for (the plane uploads), or nullptr
    hipEvent_t remap_wait = nullptr;                     // event the next remap waits for (the download that last read its output planes)
    int ensure_hostio(int rows, int cols);
    // The host entry points hand these pointers to copy engines and (output planes) to a kernel: pageable memory there is a GPU fault, not an
    // error code.  Looked up on EVERY call (hipPointerGetAttributes: ~1 us) -- an address that was pinned once may be pageable memory the next
    // time it is seen (hipHostFree / hipHostUnregister, then malloc) -- and at BOTH ends of the byte range, so that a plane that runs past
    // its registration is refused too.
    int require_pinned(const void* p, size_t bytes, const char* what)
    {
        if (!p || bytes == 0) return LVK_HIP_OK;
        for (const uint8_t* q : {(const uint8_t*)p, (const uint8_t*)p + (bytes - 1)})
        {
            hipPointerAttribute_t attr{};
            const hipError_t e = hipPointerGetAttributes(&attr, q);
            if (e != hipSuccess) (void)hipGetLastError();
            if (e != hipSuccess || (attr.type != hipMemoryTypeHost && attr.type != hipMemoryTypeManaged && attr.type != hipMemoryTypeDevice))
                return fail(LVK_HIP_ERR_ARG, std::string(what) + ": the planes of the host entry points must be PINNED host memory "
                                             "(lvk_hip_host_malloc, hipHostMalloc or hipHostRegister) over their whole extent; this pointer is pageable memory");
        }
        return LVK_HIP_OK;
    }
    // the planes of one 4:2:0 frame: one range when they are contiguous (the OBS layout), else plane by plane
    int require_pinned_planes(const void* y, int y_step, const void* u, int u_step, const void* v, int v_step, int nv12, int rows, int cols, const char* what)
    {
        if (!y) return LVK_HIP_OK;
        const int crows = rows / 2, ccols = nv12 ? cols : cols / 2;
        const size_t yb = (size_t)y_step * (rows - 1) + cols, ub = (size_t)u_step * (crows - 1) + ccols, vb = nv12 ? 0 : (size_t)v_step * (crows - 1) + ccols;
        const uint8_t* ye = (const uint8_t*)y + yb; const uint8_t* ue = (const uint8_t*)u + ub;
        if (y_step == cols && u_step == ccols && (const uint8_t*)u == ye && (nv12 || (v_step == ccols && (const uint8_t*)v == ue)))
            return require_pinned(y, yb + ub + vb, what);
        int rc;
        if ((rc = require_pinned(y, yb, what)) != LVK_HIP_OK || (rc = require_pinned(u, ub, what)) != LVK_HIP_OK) return rc;
        return nv12 ? LVK_HIP_OK : require_pinned(v, vb, what);
    }
    int host_stream(hipStream_t& s)                          // a transfer stream, created on first use; lvk_hip_sync() covers it
    {
        if (s) return LVK_HIP_OK;
        LVK_HIP_CHECK(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        ctx->aux_streams.push_back(s);
        return LVK_HIP_OK;
    }
    int flush_download(bool wait);
    int cancel_lookahead();
    // ---- look-ahead for DEVICE-resident frames (lvk_hip_stab_prefetch / _yuv420): the luma of the frame the next push will carry.  Its
    // downscale and pyramid are put on the tracking stream BEHIND this push's chain (into the pyramid that becomes `cur` at the next push), where
    // the GPU runs them during the host's turn between two chains; the next push then starts at the optical flow.
    struct LumaAhead { const void* luma = nullptr; int step = 0, pix = 0, channel = 0, rows = 0, cols = 0;
                       bool same(const void* l, int st, int px, int ch, int r, int c) const { return luma && luma == l && step == st && pix == px && channel == ch && rows == r && cols == c; } };
    LumaAhead ahead_announced;                 // announced, not yet on the stream (cleared by the push that follows, whatever it does with it)
    LumaAhead ahead_built;                     // what pyr[cur ^ 1] holds already
    unsigned long long push_seq = 0, ahead_built_for = 0;      // a built pyramid is only good for the very next push
    long lookahead_frames = 0;
    bool early_post_off = std::getenv("LVK_HIP_LATE_POST") != nullptr;      // experiments: keep the bookkeeping at the start of the next push
    void forget_device_lookahead() { ahead_announced = LumaAhead(); ahead_built = LumaAhead(); ahead_built_for = 0; }
    int host_upload(const void* h_y, int y_step, const void* h_u, int u_step, const void* h_v, int v_step, int nv12, int rows, int cols, int k, bool ahead);
    void free_hostio();
    bool caller_free_running_now();
    // experiments (scripts/host_feed_probe.py, scripts/host_feed_matrix.sh): LVK_HIP_HOST_UP2=1 announced frames alternate between two upload streams
    // (default: one, see host_upload); LVK_HIP_HOST_H2D=<blocks> the uploads as copy kernels of that many workgroups instead of hipMemcpyAsync
    double host_trace_acc[6] = {0, 0, 0, 0, 0, 0}; long host_trace_n = 0;      // LVK_HIP_HOST_TRACE: us inside lvk_hip_stab_push_yuv420_host, by phase
    int host_up2 = [] { const char* e = std::getenv("LVK_HIP_HOST_UP2"); return e ? std::atoi(e) : 0; }();
    int host_h2d_blocks = [] { const char* e = std::getenv("LVK_HIP_HOST_H2D"); return e ? std::atoi(e) : 0; }();
    int host_sink_mode = [] { const char* e = std::getenv("LVK_HIP_HOST_SINK"); return !e ? 0 : (e[0] == 'd' ? 1 : (e[0] == 'c' ? 2 : 0)); }();      // tests: direct | copy

    // ---- YUV420 front/back end: pool of packed frames the planes are converted into
    std::vector<void*> pool_all; std::deque<void*> pool_free;      // free slots are reused oldest first: the remap that read a slot is long done
    void* pool_out = nullptr;
    int pool_rows = 0, pool_cols = 0;
    int ensure_pool(int rows, int cols);
    void free_pool();
};

int lvk_hip_stab::prof_begin(int kind, hipStream_t stream)
{
    if (!stream) stream = ctx->stream;
    if (!profiling || !((prof_mask >> kind) & 1u) || (prof_tick % prof_every) != 0) return -1;
    if (ev_used >= 1024 && prof_collect() != LVK_HIP_OK) return -1;      // long sessions: fold the pending pairs in (one stream sync) and reuse them
    if (ev_used == ev_pool.size())
    {
        EvPair p{nullptr, nullptr, kind};
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return -1;
        ev_pool.push_back(p);
    }
    ev_pool[ev_used].kind = kind;
    (void)hipEventRecord(ev_pool[ev_used].a, stream);
    return (int)ev_used++;
}

void lvk_hip_stab::prof_end(int idx, hipStream_t stream) { if (idx >= 0) (void)hipEventRecord(ev_pool[(size_t)idx].b, stream ? stream : ctx->stream); }

int lvk_hip_stab::prof_collect()
{
    LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (remap_stream) LVK_HIP_CHECK(ctx, hipStreamSynchronize(remap_stream));
    for (size_t i = 0; i < ev_used; i++)
    {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev_pool[i].a, ev_pool[i].b) == hipSuccess) { prof_ms[ev_pool[i].kind] += ms; prof_n[ev_pool[i].kind]++; }
    }
    ev_used = 0;
    return LVK_HIP_OK;
}

int lvk_hip_stab::alloc_pyramids()
{
    int rc;
    if ((rc = pyr[0].allocate(ctx, s.detection_height, s.detection_width, LK_LEVELS, LK_WIN, LK_WIN)) != LVK_HIP_OK) return rc;
    if ((rc = pyr[1].allocate(ctx, s.detection_height, s.detection_width, LK_LEVELS, LK_WIN, LK_WIN)) != LVK_HIP_OK) return rc;
    pyr_w = s.detection_width; pyr_h = s.detection_height;
    forget_device_lookahead();
    return LVK_HIP_OK;
}

void lvk_hip_stab::free_tracker_buffers()
{
    void* dev[] = {d_fast_masks, d_fast_scores, d_pts, d_matched, d_p1, d_status, d_ransac_ws, d_count, d_und, d_mesh_scratch,
                   d_grid_col, d_grid_row, d_grid_bucket, d_n_points, d_full, d_cell_first, d_cell_best, d_region_count};
    for (void* p : dev) if (p) (void)hipFree(p);
    void* host[] = {h_fast_out, h_fast_counts, h_regions, h_pts, h_matched, h_p1, h_status, h_H, h_ninl, h_mask, h_und, h_count, h_occ, h_new_kp, h_insert};
    for (void* p : host) if (p) (void)hipHostFree(p);
    d_grid_col = nullptr; d_grid_row = nullptr; d_grid_bucket = nullptr; d_n_points = d_full = nullptr; d_cell_first = nullptr; d_cell_best = nullptr; d_region_count = nullptr; h_occ = h_new_kp = nullptr; h_insert = nullptr;
    d_fast_masks = d_fast_scores = nullptr;
    d_pts = d_matched = d_p1 = nullptr; d_status = nullptr; d_ransac_ws = nullptr; d_count = nullptr; d_und = nullptr; d_mesh_scratch = nullptr;
    h_fast_out = nullptr; h_fast_counts = nullptr; h_regions = nullptr; h_pts = h_matched = h_p1 = nullptr; h_status = nullptr;
    h_H = nullptr; h_ninl = nullptr; h_mask = nullptr; h_und = nullptr; h_count = nullptr;
}

int lvk_hip_stab::alloc_tracker_buffers()
{
    LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    free_tracker_buffers();
    // the grid holds at most one feature per cell; features left over from before a reset() can add as many again
    cap_features = 2 * grid.capacity() + 16;
    fast_regions = (int)grid.zones.size();
    fast_max_rw = fast_max_rh = 1;
    grid.plan(plan);
    for (const FastRegion& r : plan) { fast_max_rw = std::max(fast_max_rw, r.w); fast_max_rh = std::max(fast_max_rh, r.h); }
    // NMS keeps at most one pixel of every 2x2 block: that bounds the raw corner count of a region.
    fast_cap = ((fast_max_rw + 1) / 2) * ((fast_max_rh + 1) / 2);
    size_t mb, sb;
    lvk_fast_workspace_bytes(fast_regions, fast_max_rw, fast_max_rh, &mb, &sb);
    const size_t n = cap_features;
    LVK_HIP_CHECK(ctx, hipMalloc(&d_fast_masks, mb));
    LVK_HIP_CHECK(ctx, hipMalloc(&d_fast_scores, sb));
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_pts, n * sizeof(float2)));
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_matched, n * sizeof(float2)));
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_p1, 2 * n * sizeof(float2)));
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_status, n));
    LVK_HIP_CHECK(ctx, hipMalloc(&d_ransac_ws, lvk_ransac_workspace_bytes((int)n)));
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_count, sizeof(int)));
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_und, 2 * n * sizeof(float2)));
    LVK_HIP_CHECK(ctx, hipMalloc(&d_mesh_scratch, 32 * n));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_fast_out, (size_t)fast_regions * fast_cap * sizeof(uint32_t), hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_fast_counts, fast_regions * sizeof(int), hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_regions, fast_regions * sizeof(FastRegion), hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_pts, n * sizeof(float2), hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_matched, n * sizeof(float2), hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_p1, 2 * n * sizeof(float2), hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_status, n, hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_H, 9 * sizeof(double), hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_ninl, sizeof(int), hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_mask, n, hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_und, 2 * n * sizeof(float2), hipHostMallocDefault));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_count, sizeof(int), hipHostMallocDefault));
    // the suppression grid's tables for k_fast_insert (constant per configuration)
    {
        const auto& col = grid.col_table(); const auto& row = grid.row_base_table(); const auto& bucket = grid.bucket_table();
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_grid_col, std::max<size_t>(col.size(), 1) * sizeof(uint16_t)));
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_grid_row, std::max<size_t>(row.size(), 1) * sizeof(uint32_t)));
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_grid_bucket, std::max<size_t>(bucket.size(), 1)));
        LVK_HIP_CHECK(ctx, hipMemcpy(d_grid_col, col.data(), col.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        LVK_HIP_CHECK(ctx, hipMemcpy(d_grid_row, row.data(), row.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        LVK_HIP_CHECK(ctx, hipMemcpy(d_grid_bucket, bucket.data(), bucket.size(), hipMemcpyHostToDevice));
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_cell_first, std::max<size_t>(grid.capacity(), 1) * sizeof(uint32_t)));
        LVK_HIP_CHECK(ctx, hipMalloc(&d_cell_best, std::max<size_t>(grid.capacity(), 1) * sizeof(unsigned long long)));
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_region_count, LVK_FAST_INLINE_REGIONS * sizeof(int)));
        { const int crc = lvk_fast_cells_reset(ctx, d_cell_first, d_cell_best, (int)std::max<size_t>(grid.capacity(), 1), d_region_count); if (crc != LVK_HIP_OK) return crc; }
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_n_points, sizeof(int)));
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_full, sizeof(int)));
        LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_occ, ((grid.capacity() + 31) / 32 + 1) * sizeof(uint32_t), hipHostMallocDefault));
        LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_new_kp, std::max<size_t>(grid.capacity(), 1) * sizeof(uint32_t), hipHostMallocDefault));
        LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_insert, 8 * sizeof(int), hipHostMallocDefault));
    }
    return LVK_HIP_OK;
}

void lvk_hip_stab::tracker_restart()            // FrameTracker::restart (FrameTracker.cpp:97-104)
{
    finish_post();                              // the last frame's bookkeeping first: the detector keeps its propagated features across a reset
    tracking_stability = 0.0f;
    tracked.clear();
    grid.reset();
    initialized = false;
    if (mesh_dev) (void)lvk_mesh_solver_reset(mesh_dev, ctx->stream);
    post_n = -1;
    forget_device_lookahead();
}

void lvk_hip_stab::finish_post()
{
    if (post_n < 0) return;
    const int n = post_n, m_gpu = post_m;
    post_n = -1;
    // fast_filter(features, tracked points, matched points; keep = status): back-to-front swap-erase (Container.tpp:97-121)
    int m = n;
    for (int k = n - 1; k >= 0; k--)
        if (!h_status[k])
        {
            m--;
            std::swap(tracked[k], tracked[m]);
            std::swap(h_pts[k], h_pts[m]);
            std::swap(h_matched[k], h_matched[m]);
        }
    tracked.resize(m);
    if (m != m_gpu) { post_error = true; tracked.clear(); return; }                       // reported by the next push
    for (int i = m - 1; i >= 0; i--)                                                     // FrameTracker.cpp:183-192
    {
        if (h_mask[i]) { tracked[i].age++; tracked[i].x = h_matched[i].x; tracked[i].y = h_matched[i].y; }
        else { std::swap(tracked[i], tracked.back()); tracked.pop_back(); }
    }
    grid.propagate(tracked);
}

int lvk_hip_stab::configure(const lvk_stab_settings& st)
{
    // pre-conditions the reference asserts: StabilizationFilter.cpp:44-45, FrameTracker.cpp:59-65, FeatureDetector.cpp:50-57, PathSmoother.cpp:38-44
    LVK_HIP_REQUIRE(ctx, st.min_tracking_quality >= 0 && st.min_tracking_quality <= 1 && st.min_scene_quality >= 0 && st.min_scene_quality <= 1);
    LVK_HIP_REQUIRE(ctx, st.motion_width >= 2 && st.motion_height >= 2);
    LVK_HIP_REQUIRE(ctx, st.acceptance_threshold >= 0 && st.temporal_smoothing >= 0 && st.local_smoothing >= 0 && st.min_motion_samples >= 4);
    LVK_HIP_REQUIRE(ctx, st.uniformity_threshold >= 0 && st.uniformity_threshold <= 1);
    LVK_HIP_REQUIRE(ctx, st.detection_regions_x > 0 && st.detection_regions_y > 0);
    LVK_HIP_REQUIRE(ctx, st.detection_regions_x <= st.detection_width && st.detection_regions_y <= st.detection_height);
    LVK_HIP_REQUIRE(ctx, st.min_feature_density <= st.max_feature_density && st.min_feature_density > 0 && st.max_feature_density <= 1 && st.accumulation_rate > 0);
    LVK_HIP_REQUIRE(ctx, st.corrective_limit_x >= 0 && st.corrective_limit_x <= 1 && st.corrective_limit_y >= 0 && st.corrective_limit_y <= 1);
    LVK_HIP_REQUIRE(ctx, st.predictive_samples > 0 && st.smoothing_steps > 0 && st.response_rate >= 0 && st.response_rate <= 1);
    LVK_HIP_REQUIRE(ctx, st.detection_width >= 8 && st.detection_height >= 8 && st.detection_width < 4096 && st.detection_height < 4096);
    // the remap kernels take the mesh through a staging slot
    LVK_HIP_REQUIRE(ctx, (size_t)st.motion_width * (size_t)st.motion_height * 2 * sizeof(float) <= lvk_hip_ctx::kStageBytes);

    // ---- everything that can be refused is decided BEFORE any state changes: a configure() that returns an error leaves the filter as it was
    lvk_stab_settings prev_tracker = tracker_s;
    MeshGen gen = mesh_gen;
    if (gen.cols == 0)
    {
        // The reference's FrameTracker member is default-constructed first: FrameTracker(FrameTrackerSettings{}) generates the
        // mesh constraints for a 16x16 mesh over its default 256x256 region with weights 1.0 / 20.0 (FrameTracker.cpp:41-53,
        // FrameTracker.hpp:31-44).  configure() below then only regenerates them when the motion resolution changes.
        lvk_stab_default_settings(&prev_tracker);
        prev_tracker.motion_width = 16; prev_tracker.motion_height = 16;
        gen = MeshGen{16, 16, 256.0f, 256.0f, prev_tracker.temporal_smoothing, prev_tracker.local_smoothing};
    }
    const bool regenerate = st.motion_width != prev_tracker.motion_width || st.motion_height != prev_tracker.motion_height;
    // FrameTracker.cpp:74-82: new region, but the PREVIOUS settings' smoothing weights; m_OptimizedMesh starts from zero again
    if (regenerate) gen = MeshGen{st.motion_width, st.motion_height, (float)st.detection_width, (float)st.detection_height, prev_tracker.temporal_smoothing, prev_tracker.local_smoothing};
    lvk_mesh_solver_dev* new_solver = nullptr;
    float* new_offsets = nullptr;
    const size_t want_offsets = (size_t)st.motion_width * st.motion_height * 2;
    if (st.track_local_motions)
    {
        // the mesh the tracker solves for has the motion resolution; a configuration whose constraints were generated for another one
        // (cannot happen: a resolution change regenerates them) would index past the mesh
        LVK_HIP_REQUIRE(ctx, gen.cols == st.motion_width && gen.rows == st.motion_height);
        if (regenerate || !mesh_dev)
        {
            const int mrc = lvk_mesh_solver_create(ctx, gen.cols, gen.rows, gen.w, gen.h, gen.temporal, gen.local, &new_solver);
            if (mrc != LVK_HIP_OK) return mrc;
        }
        if (h_offsets_floats < want_offsets && hipHostMalloc((void**)&new_offsets, want_offsets * sizeof(float), hipHostMallocDefault) != hipSuccess)
        { lvk_mesh_solver_free(new_solver); return fail(LVK_HIP_ERR_RUNTIME, "mesh offsets: pinned allocation failed"); }
        if (!h_mesh_status && hipHostMalloc((void**)&h_mesh_status, sizeof(int), hipHostMallocDefault) != hipSuccess)
        { lvk_mesh_solver_free(new_solver); if (new_offsets) (void)hipHostFree(new_offsets); return fail(LVK_HIP_ERR_RUNTIME, "mesh status: pinned allocation failed"); }
    }

    // ---- commit
    if (configured && s.stabilize_output != st.stabilize_output && remap_stream)
    {
        // the 4:2:0 conversions change streams with this flag: drain the bulk stream so that no pool slot is shared across the switch
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(remap_stream));
        if (pending_release) { if (queue_kind == 1) orphaned.push_back(pending_release); pending_release = nullptr; pending_slot = -1; }
    }
    if (configured && s.stabilize_output && !st.stabilize_output) reset_context();        // StabilizationFilter.cpp:49-52
    const bool res_changed = !configured || st.detection_width != s.detection_width || st.detection_height != s.detection_height;
    const bool layout_changed = res_changed || st.detection_regions_x != s.detection_regions_x || st.detection_regions_y != s.detection_regions_y
                                || st.max_feature_density != s.max_feature_density;
    mesh_gen = gen;
    if (regenerate || new_solver)
    {
        // the solver of the previous motion resolution (or none): nothing on the stream may still be using it
        if (mesh_dev) { (void)hipStreamSynchronize(ctx->stream); lvk_mesh_solver_free(mesh_dev); }
        mesh_dev = new_solver;
    }
    if (new_offsets)
    {
        (void)hipStreamSynchronize(ctx->stream);
        if (h_offsets) (void)hipHostFree(h_offsets);
        h_offsets = new_offsets; h_offsets_floats = want_offsets;
    }
    tracker_s = st;
    smoother.configure(st);
    queue_capacity = (size_t)st.predictive_samples + 1;
    while (queue.size() > queue_capacity)
    {
        if (queue_kind == 1) orphaned.push_back(queue.front().d_ptr);      // a borrowed frame nobody will emit: give it back
        queue.pop_front();
    }
    grid.configure(st);
    if (configured && res_changed && initialized) grid.reset();                          // FrameTracker.cpp:86-91
    s = st;
    configured = true;
    if (layout_changed || !buffers_ok)
    {
        // New tracking geometry: the cached frame no longer matches, which costs one nullopt frame exactly as the
        // reference's size check does (FrameTracker.cpp:120-124).  (An allocation failure here -- out of device memory -- leaves the
        // filter unusable until a later configure() succeeds: buffers_ok stays false and every push reports it.)
        buffers_ok = false;
        int rc = alloc_tracker_buffers();
        if (rc != LVK_HIP_OK) return rc;
        if (res_changed || pyr_w != s.detection_width || pyr_h != s.detection_height)
        {
            if ((rc = alloc_pyramids()) != LVK_HIP_OK) return rc;
            prev_w = prev_h = cur_w = cur_h = 0;
        }
        buffers_ok = true;
    }
    return LVK_HIP_OK;
}

// FrameTracker::track (FrameTracker.cpp:108-196)
int lvk_hip_stab::track(const QueuedFrame& f, const void* luma, int luma_step, int luma_pix, int luma_channel, WarpMeshF& motion, bool& have_motion)
{
    have_motion = false;
    tracking_stability = 0.0f;
    last_detected = last_matched = 0; last_distribution = 0.0f;
    hipStream_t st = ctx->stream;
    int rc;

    cur ^= 1;
    std::swap(prev_w, cur_w); std::swap(prev_h, cur_h);
    cur_w = s.detection_width; cur_h = s.detection_height;
    DevicePyramid& C = pyr[cur];
    DevicePyramid& P = pyr[cur ^ 1];
    int pe = 0;
    // (look-ahead: this frame's downscale and pyramid were put behind the previous push's chain -- same planes, same geometry, announced for
    //  exactly this push --, so `C` holds them already, in stream order)
    const bool built_ahead = ahead_built_for == push_seq && ahead_built.same(luma, luma_step, luma_pix, luma_channel, f.rows, f.cols) && pyr_w == cur_w && pyr_h == cur_h;
    ahead_built = LumaAhead(); ahead_built_for = 0;
    if (built_ahead) lookahead_frames++;
    else
    {
        pe = prof_begin(LVK_STAGE_DOWNSCALE);
        if ((rc = lvk_launch_luma_area_resize(ctx, luma, luma_step, luma_pix, luma_channel, f.rows, f.cols, const_cast<uint8_t*>(C.args.lv[0].img), C.args.lv[0].step, cur_h, cur_w)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        pe = prof_begin(LVK_STAGE_PYRAMID);
        if ((rc = C.build(ctx)) != LVK_HIP_OK) return rc;
        prof_end(pe);
    }
    trace.mark(HostTrace::DOWN_PYR_LAUNCH);
    // The previous frame's list bookkeeping (fast_filter, ageing, the suppression grid's re-seed) runs HERE, in the shadow of the two kernels
    // just launched: the next kernel of this frame (optical flow) cannot start before they are done anyway, while at the end of the
    // previous push those 9 us delayed this frame's first launch -- the tracker chain and the host take turns, and that turn-taking, not
    // either of them, bounds the frame rate.
    finish_post();
    if (post_error) { post_error = false; return fail(LVK_HIP_ERR_RUNTIME, "GPU-side fast_filter disagrees with the host's"); }
    // A push that ends before the chain's synchronisation below still has the downscale reading the caller's Y plane (4:2:0 entry, luma_pix
    // == 1: "the input planes are consumed before the call returns"): wait for it.  Packed frames stay borrowed until they are released.
    auto leave_early = [&]() -> int { if (luma_pix == 1) LVK_HIP_CHECK(ctx, hipStreamSynchronize(st)); return LVK_HIP_OK; };
    if (!initialized || cur_w != prev_w || cur_h != prev_h) { initialized = true; return leave_early(); }

    // ---- FeatureDetector::detect
    grid.plan(plan);
    bool any = false;
    for (size_t i = 0; i < plan.size(); i++) { h_regions[i] = plan[i]; any = any || plan[i].active; }
    // On a frame on which the detector runs, the corners go through the suppression grid ON THE DEVICE, inside the chain (k_fast_insert): the
    // new features land behind the held ones in the flow kernel's point list, and the kernels that follow take the point count and the
    // model choice from device memory -- one chain and one synchronisation per frame, like a frame without detection.  The host loop stays
    // for what the kernel does not cover (huge grids, regions off the pixel grid, more points than the on-device fast_filter takes).
    const size_t n_held = grid.held.size();
    const size_t n_bound = std::min(cap_features, n_held + (grid.capacity() - grid.used_cells()));      // every free cell takes at most one corner
    const bool dev_insert = any && device_grid && grid.device_insert_ok() && (int)plan.size() <= LVK_FAST_INLINE_REGIONS &&
                            lvk_fast_insert_fits((int)grid.capacity(), (int)plan.size(), fast_max_rw, fast_max_rh, cur_w, cur_h) && n_bound >= 1 && n_bound <= 4096;
    float distribution = 0.0f;
    if (dev_insert)
    {
        for (size_t i = 0; i < n_held; i++) h_pts[i] = make_float2(grid.held[i].x, grid.held[i].y);
        int occ_bucket[16];
        grid.occupancy(h_occ, occ_bucket);
        const FastInsertDesc d{d_grid_col, d_grid_row, d_grid_bucket, d_cell_first, d_cell_best, d_region_count, h_occ, occ_bucket, (int)grid.capacity(), grid.grid_cols() <= 4 || grid.grid_rows() <= 4, (int)n_held,
                               s.min_motion_samples, s.uniformity_threshold, HOMOGRAPHY_DISTRIBUTION_THRESHOLD, h_pts, h_new_kp, h_insert, d_n_points, d_full, h_fast_counts};
        pe = prof_begin(LVK_STAGE_FAST);
        if ((rc = lvk_launch_fast_insert(ctx, C.args.lv[0].img, C.args.lv[0].step, cur_h, cur_w, h_regions, (int)plan.size(), fast_max_rw, fast_max_rh,
                                         d_fast_masks, d_fast_scores, d)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        device_grid_frames++;
    }
    else
    {
    if (any)
    {
        // The small per-frame parameter / result blocks live in pinned, device-visible host memory: the kernels read the
        // region descriptors from it and write the keypoint list straight into it (posted PCIe writes), so the only host
        // call besides the launches is the stream synchronisation.
        pe = prof_begin(LVK_STAGE_FAST);
        if ((rc = lvk_launch_fast(ctx, C.args.lv[0].img, C.args.lv[0].step, cur_h, cur_w, h_regions, (int)plan.size(), fast_max_rw, fast_max_rh,
                                  d_fast_masks, d_fast_scores, h_fast_out, fast_cap, h_fast_counts, h_regions)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
        trace.mark(HostTrace::FAST_SYNC);
        host_grid_frames++;
    }
    for (size_t i = 0; i < plan.size(); i++)
        if (plan[i].active) grid.absorb(i, h_fast_out + i * (size_t)fast_cap, std::min(h_fast_counts[i], fast_cap));
    distribution = grid.finish(tracked);
    last_distribution = distribution; last_detected = (int)tracked.size();
    if (tracked.size() < (size_t)s.min_motion_samples || distribution < s.uniformity_threshold) { tracked.clear(); return leave_early(); }
    if (tracked.size() > cap_features) return fail(LVK_HIP_ERR_RUNTIME, "feature count exceeds the suppression grid capacity");
    }

    trace.mark(HostTrace::GRID);
    // ---- sparse optical flow prev -> cur
    // (device grid: n is the upper bound the kernels are launched for; they read the count itself from d_n_points)
    int n = dev_insert ? (int)n_bound : (int)tracked.size();
    if (!dev_insert) for (int i = 0; i < n; i++) h_pts[i] = make_float2(tracked[i].x, tracked[i].y);
    const int* dn = dev_insert ? d_n_points : nullptr;
    const int* dfull = dev_insert ? d_full : nullptr;
    const bool full = distribution > HOMOGRAPHY_DISTRIBUTION_THRESHOLD;         // (device grid: decided by the kernel, d_full)
    // Global-motion mode without a lens model: the whole chain optical flow -> fast_filter -> RANSAC runs on the GPU without a
    // host round trip in between (the flow kernel reads the points from pinned host memory, k_match_compact reproduces the host's
    // swap-erase order); the host synchronises once and then repeats the cheap bookkeeping on its own copies.
    const bool chained = n <= 4096;
    const bool field = s.track_local_motions != 0;
    pe = prof_begin(LVK_STAGE_PYRLK);
    if (chained)
    {
        // fused lens mode: the motion is estimated between lens-corrected positions (what the reference chain LC -> VS tracks); the flow
        // kernel writes them itself (d_und: previous | matched)
        if ((rc = lvk_launch_pyrlk(ctx, P.args, C.args, h_pts, n, d_matched, d_status, LK_WIN, LK_WIN, LK_ITERS, LK_EPS, LK_MIN_EIG, d_pts,
                                   lens ? &lens_model : nullptr, (double)f.cols / (double)cur_w, (double)f.rows / (double)cur_h, lens ? d_und : nullptr, dn)) != LVK_HIP_OK) return rc;
        const bool fused_compact = !field && n <= LVK_COMPACT_RANSAC_MAX;        // the RANSAC's first kernel compacts the flow result itself
        if (!fused_compact && (rc = lvk_launch_match_compact(ctx, d_pts, d_matched, d_status, n, d_p1, d_p1 + cap_features, d_count, h_count, h_matched, h_status,
                                                             lens ? d_und : nullptr, (float)cur_w, (float)cur_h, dn)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        pe = prof_begin(LVK_STAGE_MOTION);
        if (field)
        {
            // estimate_local_motions (FrameTracker.cpp:200-321): least-squares mesh through the matches, solved on the device
            if ((rc = lvk_launch_mesh_solve(mesh_dev, st, d_mesh_scratch, d_p1, d_p1 + cap_features, d_count, n, s.min_motion_samples, (float)cur_w, (float)cur_h,
                                            s.temporal_smoothing, s.acceptance_threshold, h_offsets, h_mask, h_mesh_status)) != LVK_HIP_OK) return rc;
        }
        else if (fused_compact)
        {
            if ((rc = lvk_launch_compact_ransac(ctx, d_pts, d_matched, d_status, n, d_p1, d_p1 + cap_features, d_count, h_count, h_matched, h_status,
                                                lens ? d_und : nullptr, (float)cur_w, (float)cur_h,
                                                s.acceptance_threshold, (double)cur_w, (double)cur_h, full, d_ransac_ws, h_H, h_ninl, h_mask, dn, dfull)) != LVK_HIP_OK) return rc;
        }
        else if ((rc = lvk_launch_ransac(ctx, d_p1, d_p1 + cap_features, n, s.acceptance_threshold, (double)cur_w, (double)cur_h, full, d_ransac_ws, h_H, h_ninl, h_mask, d_count, dfull)) != LVK_HIP_OK) return rc;
        prof_end(pe);
    }
    else
    {
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_pts, h_pts, n * sizeof(float2), hipMemcpyHostToDevice, st));
        if ((rc = lvk_launch_pyrlk(ctx, P.args, C.args, d_pts, n, h_matched, h_status, LK_WIN, LK_WIN, LK_ITERS, LK_EPS, LK_MIN_EIG)) != LVK_HIP_OK) return rc;
        prof_end(pe);
    }
    trace.mark(HostTrace::LK_LAUNCH);
    bool chain_event_armed = false;              // local: an error return below must not leave a stale event armed for the next push
    if (deferred_ingest && tracker_ingest_capable)
    {
        // Where the conversion goes: a bulk stream that is idle (a caller that synchronises every frame) takes it now, next to the
        // chain -- the remap that follows then has the GPU to itself; a bulk stream that is still busy with the previous remap (a
        // free-running caller) would only get to it after that, so it goes behind the chain on this stream, and the push waits for
        // the chain through an event instead of for the stream.
        const hipError_t q = hipStreamQuery(remap_stream);
        if (q != hipSuccess) (void)hipGetLastError();
        // (host-resident frames whose chroma planes are still on the link: behind the chain as well -- on the bulk stream the conversion would
        //  hold the output remap back until they have arrived)
        // (a remap whose stores cross the host link leaves the bulk stream time to spare -- 283 us of link time per frame against ~235 us of a
        //  slowed-down chain --: an announced host frame, whose planes have arrived, is converted there; 3 320 against 3 215 frames/s, and the
        //  pushes on which the detector runs no longer stand out: p90 0.312 instead of 0.364 ms)
        ingest_on_tracker = ingest_placement == 1 || (ingest_placement == 0 && ((q == hipErrorNotReady && !host_direct_now) || ingest_wait[0] != nullptr));
    }
    // (an announced next frame: its downscale + pyramid go behind the chain too, so the push waits for the chain through the event as well)
    const bool build_ahead = chained && ahead_announced.luma != nullptr && pyr_w == cur_w && pyr_h == cur_h;
    if (chained && ((deferred_ingest && tracker_ingest_capable && ingest_on_tracker) || build_ahead))
    {
        if (!chain_done) LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&chain_done, hipEventDisableTiming));
        LVK_HIP_CHECK(ctx, hipEventRecord(chain_done, st)); chain_event_armed = true;
    }
    if (deferred_ingest && (rc = run_deferred_ingest()) != LVK_HIP_OK) return rc;
    if (build_ahead)
    {
        // `P` (the previous frame's pyramid) has been read for the last time by the flow kernel above; at the next push it is `C`
        const LumaAhead a = ahead_announced;
        ahead_announced = LumaAhead();
        pe = prof_begin(LVK_STAGE_DOWNSCALE);
        if ((rc = lvk_launch_luma_area_resize(ctx, a.luma, a.step, a.pix, a.channel, a.rows, a.cols, const_cast<uint8_t*>(P.args.lv[0].img), P.args.lv[0].step, cur_h, cur_w)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        pe = prof_begin(LVK_STAGE_PYRAMID);
        if ((rc = P.build(ctx)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        ahead_built = a; ahead_built_for = push_seq + 1;
    }
    if (lens && !chained)
    {
        if ((rc = lvk_launch_lens_undistort(ctx, st, lens_model, (double)f.cols / (double)cur_w, (double)f.rows / (double)cur_h,
                                            d_pts, n, h_matched, n, h_und)) != LVK_HIP_OK) return rc;
    }
    if (chain_event_armed) LVK_HIP_CHECK(ctx, hipEventSynchronize(chain_done));
    else LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
    trace.mark(HostTrace::LK_SYNC);

    if (dev_insert)
    {
        // what the host loop would have left behind: thresholds, the feature list (held + new), the distribution quality -- recomputed here
        // from the new features and held against the kernel's own figure
        const int n_new = h_insert[0];
        float q_dev; std::memcpy(&q_dev, &h_insert[2], sizeof(float));
        if (n_new < 0 || (size_t)n_new > grid.capacity()) return fail(LVK_HIP_ERR_RUNTIME, "device suppression grid returned an impossible count");
        distribution = grid.finish_device(tracked, h_new_kp, n_new, h_fast_counts, fast_cap);
        last_distribution = distribution; last_detected = (int)tracked.size();
        if (distribution != q_dev || tracked.size() != n_held + (size_t)n_new)
            return fail(LVK_HIP_ERR_RUNTIME, "device suppression grid disagrees with the host's bookkeeping");
        // FrameTracker.cpp:127-131 (the kernels of the chain saw a point count of zero and did nothing)
        if (tracked.size() < (size_t)s.min_motion_samples || distribution < s.uniformity_threshold) { tracked.clear(); return LVK_HIP_OK; }
        if (tracked.size() > cap_features) return fail(LVK_HIP_ERR_RUNTIME, "feature count exceeds the suppression grid capacity");
        n = (int)tracked.size();
    }
    if (chained)
    {
        // everything the remap launch needs is in the pinned result block; the list bookkeeping follows in finish_post()
        const int m = *h_count;
        last_matched = m;
        if (m < 0 || m > n) return fail(LVK_HIP_ERR_RUNTIME, "GPU-side fast_filter returned an impossible count");
        if ((size_t)m < (size_t)s.min_motion_samples) { tracked.clear(); return LVK_HIP_OK; }
        motion = WarpMeshF(s.motion_height, s.motion_width);
        if (field)
        {
            if (*h_mesh_status != 0) { tracked.clear(); return LVK_HIP_OK; }             // no estimate this frame (identity motion)
            std::memcpy(motion.off.data(), h_offsets, motion.off.size() * sizeof(float));
        }
        else
        {
            std::memcpy(last_H, h_H, sizeof(last_H));
            motion.from_homography(last_H, (float)cur_w, (float)cur_h);
        }
        size_t inliers = 0;
        for (int i = 0; i < m; i++) inliers += h_mask[i] ? 1 : 0;
        tracking_stability = (float)inliers / (float)m;                                  // ratio_of(inlier_status, 1)
        have_motion = true;
        post_n = n; post_m = m;
        trace.mark(HostTrace::POST);
        return LVK_HIP_OK;
    }

    if (lens && !chained)                     // (the chained path has folded this test into the status flags on the GPU)
    {
        // a match whose corrected positions leave the tracking region is not visible in the lens-corrected frame: drop it
        const float w = (float)cur_w, h = (float)cur_h;
        auto inside = [&](const float2& p) { return p.x >= 0.0f && p.x < w && p.y >= 0.0f && p.y < h; };
        for (int k = 0; k < n; k++)
            if (!(inside(h_und[k]) && inside(h_und[n + k]))) h_status[k] = 0;
    }
    // fast_filter(features, tracked points, matched points; keep = status): back-to-front swap-erase (Container.tpp:97-121)
    int m = n;
    for (int k = n - 1; k >= 0; k--)
        if (!h_status[k])
        {
            m--;
            std::swap(tracked[k], tracked[m]);
            std::swap(h_pts[k], h_pts[m]);
            std::swap(h_matched[k], h_matched[m]);
            if (lens && !chained) { std::swap(h_und[k], h_und[m]); std::swap(h_und[n + k], h_und[n + m]); }
        }
    tracked.resize(m);
    last_matched = m;
    if (chained && *h_count != m) return fail(LVK_HIP_ERR_RUNTIME, "GPU-side fast_filter disagrees with the host's");
    if ((size_t)m < (size_t)s.min_motion_samples) { tracked.clear(); return LVK_HIP_OK; }

    trace.mark(HostTrace::FILTER);
    // ---- motion estimate
    motion = WarpMeshF(s.motion_height, s.motion_width);
    const float2* e1 = lens ? h_und : h_pts;
    const float2* e2 = lens ? h_und + n : h_matched;
    if (s.track_local_motions)
    {
        // estimate_local_motions (FrameTracker.cpp:200-321): least-squares mesh through the feature matches (more matches than the GPU-side
        // fast_filter handles: the pairs go up in one copy, as for the RANSAC below)
        std::memcpy(h_p1, e1, m * sizeof(float2));
        std::memcpy(h_p1 + m, e2, m * sizeof(float2));
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_p1, h_p1, 2 * (size_t)m * sizeof(float2), hipMemcpyHostToDevice, st));
        if ((rc = lvk_launch_mesh_solve(mesh_dev, st, d_mesh_scratch, d_p1, d_p1 + m, nullptr, m, 0, (float)cur_w, (float)cur_h,
                                        s.temporal_smoothing, s.acceptance_threshold, h_offsets, h_mask, h_mesh_status)) != LVK_HIP_OK) return rc;
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
        if (*h_mesh_status != 0)
        {
            tracked.clear();                                                          // like the other no-motion exits
            return LVK_HIP_OK;                                                        // no estimate this frame (identity motion)
        }
        std::memcpy(motion.off.data(), h_offsets, motion.off.size() * sizeof(float));
        size_t inl = 0;
        for (int i = 0; i < m; i++) inl += h_mask[i] ? 1 : 0;
        tracking_stability = (float)inl / (float)m;
        for (int i = m - 1; i >= 0; i--)
        {
            if (h_mask[i]) { tracked[i].age++; tracked[i].x = h_matched[i].x; tracked[i].y = h_matched[i].y; }
            else { std::swap(tracked[i], tracked.back()); tracked.pop_back(); }
        }
        grid.propagate(tracked);
        have_motion = true;
        return LVK_HIP_OK;
    }
    // both point sets travel in one copy (h_p1 and d_p1 each hold p1 | p2 in one allocation); results come back through
    // the pinned host block the kernel writes directly
    if (!chained)
    {
        std::memcpy(h_p1, e1, m * sizeof(float2));
        std::memcpy(h_p1 + m, e2, m * sizeof(float2));
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_p1, h_p1, 2 * (size_t)m * sizeof(float2), hipMemcpyHostToDevice, st));
        pe = prof_begin(LVK_STAGE_MOTION);
        if ((rc = lvk_launch_ransac(ctx, d_p1, d_p1 + m, m, s.acceptance_threshold, (double)cur_w, (double)cur_h, full, d_ransac_ws, h_H, h_ninl, h_mask)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        trace.mark(HostTrace::RANSAC_LAUNCH);
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
        trace.mark(HostTrace::RANSAC_SYNC);
    }
    std::memcpy(last_H, h_H, sizeof(last_H));
    motion.from_homography(last_H, (float)cur_w, (float)cur_h);

    size_t inliers = 0;
    for (int i = 0; i < m; i++) inliers += h_mask[i] ? 1 : 0;
    tracking_stability = (float)inliers / (float)m;                                      // ratio_of(inlier_status, 1)

    for (int i = m - 1; i >= 0; i--)                                                     // FrameTracker.cpp:183-192
    {
        if (h_mask[i]) { tracked[i].age++; tracked[i].x = h_matched[i].x; tracked[i].y = h_matched[i].y; }
        else { std::swap(tracked[i], tracked.back()); tracked.pop_back(); }
    }
    grid.propagate(tracked);
    have_motion = true;
    trace.mark(HostTrace::POST);
    return LVK_HIP_OK;
}

extern "C" {

void lvk_stab_default_settings(lvk_stab_settings* s)
{
    if (!s) return;
    // FeatureDetector.hpp:28-37, FrameTracker.hpp:31-44, PathSmoother.hpp:29-39, StabilizationFilter.hpp:28-39
    s->detection_width = 256; s->detection_height = 256; s->detection_regions_x = 2; s->detection_regions_y = 2; s->force_detection = 0;
    s->max_feature_density = 0.20f; s->min_feature_density = 0.05f; s->accumulation_rate = 2.0f;
    s->track_local_motions = 1; s->temporal_smoothing = 1.0f; s->local_smoothing = 20.0f;
    s->min_motion_samples = 75; s->acceptance_threshold = 8.0f; s->uniformity_threshold = 0.20f;
    s->predictive_samples = 10; s->corrective_limit_x = 0.1f; s->corrective_limit_y = 0.1f; s->smoothing_steps = 20.0f; s->response_rate = 0.04f;
    s->motion_width = 2; s->motion_height = 2;
    s->background[0] = 255; s->background[1] = 0; s->background[2] = 255;
    s->crop_to_stable_region = 0; s->stabilize_output = 1; s->min_scene_quality = 0.8f; s->min_tracking_quality = 0.3f;
}

int lvk_hip_stab_create(lvk_hip_ctx* ctx, const lvk_stab_settings* settings, lvk_hip_stab** out)
{
    if (!ctx) return LVK_HIP_ERR_ARG;
    LVK_HIP_REQUIRE(ctx, settings && out);
    lvk_device_guard device_guard(ctx);
    *out = nullptr;
    auto* st = new lvk_hip_stab();
    st->ctx = ctx;
    const int rc = st->configure(*settings);
    if (rc != LVK_HIP_OK)
    {
        st->free_tracker_buffers(); lvk_mesh_solver_free(st->mesh_dev); st->pyr[0].release(); st->pyr[1].release();
        if (st->h_offsets) (void)hipHostFree(st->h_offsets);
        if (st->h_mesh_status) (void)hipHostFree(st->h_mesh_status);
        delete st; return rc;
    }
    *out = st;
    return LVK_HIP_OK;
}

static void rehome_stage_events(lvk_hip_ctx* ctx);

void lvk_hip_stab_destroy(lvk_hip_stab* st)
{
    if (!st) return;
    lvk_device_guard device_guard(st->ctx);
    (void)hipStreamSynchronize(st->ctx->stream);
    st->trace.dump();
    if (st->trace.on && st->host_trace_n)
        std::fprintf(stderr, "[lvk host trace] push_yuv420_host, us/frame: uploads enqueued %.1f, stream wait + sink choice %.1f, inner push %.1f, chroma wait %.1f, download enqueued %.1f\n",
                     st->host_trace_acc[0] / st->host_trace_n, st->host_trace_acc[1] / st->host_trace_n, st->host_trace_acc[2] / st->host_trace_n,
                     st->host_trace_acc[3] / st->host_trace_n, st->host_trace_acc[4] / st->host_trace_n);
    st->free_tracker_buffers();
    lvk_mesh_solver_free(st->mesh_dev);
    if (st->h_offsets) (void)hipHostFree(st->h_offsets);
    if (st->h_mesh_status) (void)hipHostFree(st->h_mesh_status);
    st->pyr[0].release(); st->pyr[1].release();
    st->free_pool();
    st->free_hostio();
    if (st->remap_stream)
    {
        (void)hipStreamSynchronize(st->remap_stream);
        rehome_stage_events(st->ctx);
        auto& aux = st->ctx->aux_streams;
        aux.erase(std::remove(aux.begin(), aux.end(), st->remap_stream), aux.end());
        if (st->remap_stream_owned) (void)hipStreamDestroy(st->remap_stream);
    }
    for (int i = 0; i < 2; i++) if (st->remap_done[i]) (void)hipEventDestroy(st->remap_done[i]);
    if (st->ingest_done) (void)hipEventDestroy(st->ingest_done);
    if (st->chain_done) (void)hipEventDestroy(st->chain_done);
    if (st->caller_ready) (void)hipEventDestroy(st->caller_ready);
    for (auto& p : st->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    delete st;
}

// Overlap mode: the EASU remap of the delayed frame runs on a second stream, concurrently with the tracking of the
// next frame (the two are independent: the delayed frame was uploaded at least one push earlier).  The output of a
// push is then complete only after lvk_hip_sync(), and a borrowed frame is handed back (*released) one push later,
// after its remap has finished.
// The stream the output of the next pushes is produced on (the bulk stream in overlap mode, else the context's): a caller that
// wants to chain its own stream-ordered work behind an output (a D2H copy, an encoder) enqueues it there instead of synchronising.
void* lvk_hip_stab_output_stream(lvk_hip_stab* st)
{
    if (!st) return nullptr;
    return (void*)((st->overlap && st->s.stabilize_output && st->remap_stream) ? st->remap_stream : st->ctx->stream);
}

// The context's staging slots carry an event "the kernel that read this slot is done", recorded on whatever stream launched that kernel --
// also on a bulk stream that is about to go away.  An event whose stream has been destroyed cannot be waited for any more
// (hipEventSynchronize fails), so before a stream of this stabilizer dies the slots' events move to the context's own stream (everything
// on the dying stream has completed: it was synchronised).
static void rehome_stage_events(lvk_hip_ctx* ctx)
{
    for (int i = 0; i < lvk_hip_ctx::kStageSlots; i++) if (ctx->stage_done[i]) (void)hipEventRecord(ctx->stage_done[i], ctx->stream);
}

static int stab_detach_bulk_stream(lvk_hip_stab* st)
{
    lvk_hip_ctx* ctx = st->ctx;
    if (!st->remap_stream) return LVK_HIP_OK;
    (void)hipStreamSynchronize(st->remap_stream);
    rehome_stage_events(ctx);
    // Events of this stabilizer that were recorded on the stream that goes away: everything on it has completed, so nothing has to wait for
    // them any more -- and an event whose stream has been destroyed must not be waited for at all.  The per-slot "remap has read this pool
    // slot" events are disarmed; the events a later push waits on unconditionally (remap_done of a pending release, ingest_done) are
    // re-recorded on the context's own stream.
    std::fill(st->slot_read_armed.begin(), st->slot_read_armed.end(), (char)0);
    for (int i = 0; i < 2; i++) if (st->remap_done[i]) (void)hipEventRecord(st->remap_done[i], ctx->stream);
    if (st->ingest_done) (void)hipEventRecord(st->ingest_done, ctx->stream);
    st->remap_wait = nullptr;
    auto& aux = ctx->aux_streams;
    aux.erase(std::remove(aux.begin(), aux.end(), st->remap_stream), aux.end());
    if (st->remap_stream_owned) LVK_HIP_CHECK(ctx, hipStreamDestroy(st->remap_stream));
    st->remap_stream = nullptr; st->remap_stream_owned = false;
    return LVK_HIP_OK;
}

static int stab_set_overlap(lvk_hip_stab* st, bool enable, lvk_hip_ctx* bulk)
{
    lvk_hip_ctx* ctx = st->ctx;
    LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (st->remap_stream) LVK_HIP_CHECK(ctx, hipStreamSynchronize(st->remap_stream));
    // both streams are idle: a frame whose remap was pending is free again
    if (st->pending_release) { if (st->queue_kind == 1) st->orphaned.push_back(st->pending_release); st->pending_release = nullptr; st->pending_slot = -1; }
    st->caller_wait_pending = false;
    if (enable)
    {
        hipStream_t want = bulk ? bulk->stream : nullptr;
        if (bulk) LVK_HIP_REQUIRE(ctx, bulk != ctx && bulk->device == ctx->device && bulk->stream != ctx->stream);
        if (st->remap_stream && (bulk ? st->remap_stream != want : !st->remap_stream_owned))
        { const int rc = stab_detach_bulk_stream(st); if (rc != LVK_HIP_OK) return rc; }
        if (!st->remap_stream)
        {
            if (bulk) { st->remap_stream = want; st->remap_stream_owned = false; }
            else
            {
                // lowest priority: the bulk kernels of this stream (remap, 4:2:0 conversion) fill every CU; the tracker's small,
                // latency-bound kernels on the main stream should get the wave slots they free first
                int prio_least = 0, prio_greatest = 0;
                LVK_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
                LVK_HIP_CHECK(ctx, hipStreamCreateWithPriority(&st->remap_stream, hipStreamNonBlocking, prio_least));
                st->remap_stream_owned = true;
            }
            ctx->aux_streams.push_back(st->remap_stream);
        }
        for (int i = 0; i < 2; i++)
            if (!st->remap_done[i]) LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&st->remap_done[i], hipEventDisableTiming));
    }
    else if (st->remap_stream && !st->remap_stream_owned)
    {
        // a caller-owned bulk stream is let go of when the overlap ends (lvk_hip.h: NULL = overlap off): the caller may destroy that context
        // now, and nothing here -- lvk_hip_sync through aux_streams, configure(), destroy -- touches its stream again
        const int rc = stab_detach_bulk_stream(st); if (rc != LVK_HIP_OK) return rc;
    }
    st->overlap = enable;
    return LVK_HIP_OK;
}

int lvk_hip_stab_set_overlap(lvk_hip_stab* st, int enable)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    return stab_set_overlap(st, enable != 0, nullptr);
}

// Overlap mode on a stream the CALLER owns: the bulk kernels run on `bulk`'s stream (NULL: overlap off).  For hosts whose output frames
// outlive the stabilizer or are consumed by stream-ordered work of their own: the frames then belong to `bulk` (the C++ facade does this).
int lvk_hip_stab_set_bulk_context(lvk_hip_stab* st, lvk_hip_ctx* bulk)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    return stab_set_overlap(st, bulk != nullptr, bulk);
}

// Per-stage GPU time measured with HIP events on the launch stream.  enable != 0 starts (and resets) the
// accumulation; lvk_hip_stab_get_profile synchronises the stream and reports, per stage, the summed milliseconds
// and the number of timed launches (stage ids: LVK_STAGE_*).
int lvk_hip_stab_set_profiling(lvk_hip_stab* st, int enable)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    const int rc = st->prof_collect();
    st->profiling = enable != 0;
    st->prof_mask = enable == 1 ? ~0u : ((unsigned)enable & 0xffffu) >> 1;   // 1: every stage; otherwise (1 << (stage + 1)) bits
    st->prof_every = std::max(1u, ((unsigned)enable >> 16) & 0xffu);        // bits 16..23: sample one push in N (0 / 1 = every push)
    st->prof_tick = 0;
    for (int i = 0; i < LVK_STAGE_COUNT; i++) { st->prof_ms[i] = 0; st->prof_n[i] = 0; }
    return rc;
}

int lvk_hip_stab_get_profile(lvk_hip_stab* st, double total_ms[LVK_STAGE_COUNT], long long launches[LVK_STAGE_COUNT])
{
    if (!st || !total_ms || !launches) return LVK_HIP_ERR_ARG;
    const int rc = st->prof_collect();
    for (int i = 0; i < LVK_STAGE_COUNT; i++) { total_ms[i] = st->prof_ms[i]; launches[i] = st->prof_n[i]; }
    return rc;
}

int lvk_hip_stab_configure(lvk_hip_stab* st, const lvk_stab_settings* settings)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    LVK_HIP_REQUIRE(st->ctx, settings);
    st->finish_post();
    return st->configure(*settings);
}

// lvk::col::{RED, GREEN, BLUE}[format] (Functions/Drawing.hpp:27-71)
static void overlay_colours(int format, double red[3], double green[3], double blue[3])
{
    const double R[3][3] = {{0, 0, 255}, {255, 0, 0}, {76, 84, 255}}, G[3][3] = {{0, 255, 0}, {0, 255, 0}, {149, 43, 21}},
                 B[3][3] = {{255, 0, 0}, {0, 0, 255}, {29, 255, 107}};
    const int k = format == LVK_FORMAT_YUV ? 2 : (format == LVK_FORMAT_RGB || format == LVK_FORMAT_RGBA ? 1 : 0);
    for (int i = 0; i < 3; i++) { red[i] = R[k][i]; green[i] = G[k][i]; blue[i] = B[k][i]; }
}

// StabilizationFilter::draw_trackers (StabilizationFilter.cpp:163-175): crosses (size 7, thickness 4 -- FrameTracker.cpp:498-503
// passes a literal 4, not its `thickness` argument) at the tracked features, coloured lerp(RED, GREEN, trust), into the newest
// queued frame (the caller's borrowed buffer).
int lvk_hip_stab_draw_trackers(lvk_hip_stab* st)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    LVK_HIP_REQUIRE(st->ctx, !st->queue.empty());                                           // StreamBuffer::newest: !is_empty()
    st->finish_post();
    const QueuedFrame& f = st->queue.back();
    double r[3], g[3], b[3];
    overlay_colours(f.format, r, g, b);
    uint8_t col[3];
    for (int i = 0; i < 3; i++) col[i] = (uint8_t)(r[i] + (double)st->trust * (g[i] - r[i]));      // Math.tpp:124-129, Drawing.tpp:184-189
    std::vector<float> pts(st->tracked.size() * 2);
    for (size_t i = 0; i < st->tracked.size(); i++) { pts[2 * i] = st->tracked[i].x; pts[2 * i + 1] = st->tracked[i].y; }
    const float sx = (float)f.cols / (float)st->tracker_s.detection_width, sy = (float)f.rows / (float)st->tracker_s.detection_height;
    return lvk_launch_draw_crosses(st->ctx, st->ctx->stream, const_cast<void*>(f.d_ptr), f.step, f.rows, f.cols, pts.data(), (int)st->tracked.size(),
                                   sx, sy, col, 7, 4);
}

// StabilizationFilter::draw_motion_mesh (:179-188): BLUE grid of motion_resolution - 1 cells, thickness 1
int lvk_hip_stab_draw_motion_mesh(lvk_hip_stab* st)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    LVK_HIP_REQUIRE(st->ctx, !st->queue.empty());
    const QueuedFrame& f = st->queue.back();
    double r[3], g[3], b[3];
    overlay_colours(f.format, r, g, b);
    const uint8_t col[3] = {(uint8_t)b[0], (uint8_t)b[1], (uint8_t)b[2]};
    return lvk_launch_draw_grid(st->ctx, st->ctx->stream, const_cast<void*>(f.d_ptr), f.step, f.rows, f.cols, st->s.motion_width - 1, st->s.motion_height - 1, col, 1);
}

int lvk_hip_stab_restart(lvk_hip_stab* st);
int lvk_hip_stab_set_lens(lvk_hip_stab* st, const lvk_camera_params* params)
{
    if (!st) return LVK_HIP_ERR_ARG;
    if (params && (params->fx == 0.0 || params->fy == 0.0)) return st->fail(LVK_HIP_ERR_ARG, "camera profile with zero focal length");
    st->lens = params != nullptr;
    if (params) st->lens_params = *params;
    st->lens_rows = st->lens_cols = 0;
    return lvk_hip_stab_restart(st);
}

int lvk_hip_stab_restart(lvk_hip_stab* st)          // StabilizationFilter::restart (StabilizationFilter.cpp:139-144)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    // the queue's frames go back to their owners: nothing on the bulk stream may still be reading them
    if (st->remap_stream) LVK_HIP_CHECK(st->ctx, hipStreamSynchronize(st->remap_stream));
    // host entry points: the emitted frame whose download has not been handed to the copy engine yet still goes out (*produced was
    // reported); frames that were announced and never pushed are forgotten -- a restart is where a caller seeks or switches sources, and a
    // stale announcement would otherwise refuse every later push ("another frame has been announced") or, matched by pointer identity,
    // feed a reused buffer's pre-restart upload to the tracker
    { int hrc; if ((hrc = st->flush_download(true)) != LVK_HIP_OK || (hrc = st->cancel_lookahead()) != LVK_HIP_OK) return hrc; }
    st->scene_quality = 1.0f;
    st->queue.clear(); st->queue_kind = 0;
    st->pending_release = nullptr; st->pending_slot = -1;
    st->orphaned.clear();                          // restart(): every borrowed frame is the caller's again
    st->reset_context();
    return LVK_HIP_OK;
}

int lvk_hip_stab_reset_context(lvk_hip_stab* st)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    st->reset_context();
    return LVK_HIP_OK;
}

int lvk_hip_stab_ready(const lvk_hip_stab* st) { return st && st->queue.size() == st->queue_capacity ? 1 : 0; }
int lvk_hip_stab_frame_delay(const lvk_hip_stab* st) { return st ? st->s.predictive_samples : 0; }

} // extern "C"

// StabilizationFilter::filter (StabilizationFilter.cpp:69-135).  (luma, luma_step, luma_pix): where the tracker reads the
// luma of this frame from -- the packed frame itself (pix 3) or, on the YUV420 path, the caller's planar Y (pix 1).
// planes of a 4:2:0 output: when given, a warped frame leaves through the fused remap + egress kernel instead of d_out
struct OutPlanes420 { void* y; int y_step; void* u; int u_step; void* v; int v_step; int nv12; bool used; };

static int push_impl(lvk_hip_stab* st, const void* d_frame, int step, int rows, int cols, uint64_t timestamp, int format,
                     const void* luma, int luma_step, int luma_pix,
                     void* d_out, int out_step, int* produced, uint64_t* out_timestamp, const void** released, OutPlanes420* o420 = nullptr)
{
    lvk_hip_ctx* ctx = st->ctx;
    if (produced) *produced = 0;
    if (released) *released = nullptr;
    LVK_HIP_REQUIRE(ctx, d_frame && rows > 0 && cols > 0 && step >= 3 * cols);           // !input.empty()
    if (!st->buffers_ok) return ctx->fail(LVK_HIP_ERR_RUNTIME, "the last configure() failed while allocating the tracker's buffers: configure again");
    st->caller_runs_free = st->caller_free_running_now() || st->host_free_running_hint;
    // 3-channel VideoFrame formats (VideoFrame.cpp:170-306): YUV tracks channel 0, BGR / RGB track cvtColor(..2GRAY); the remap
    // runs the YUV or the RGB EASU program by the frame's format (Image.cpp:36-41).  GRAY / 4-channel frames are not on this path.
    LVK_HIP_REQUIRE(ctx, format == LVK_FORMAT_YUV || format == LVK_FORMAT_BGR || format == LVK_FORMAT_RGB);
    const int luma_channel = format == LVK_FORMAT_YUV ? 0 : (format == LVK_FORMAT_BGR ? -1 : -2);
    LVK_HIP_REQUIRE(ctx, luma_pix == 3 || format == LVK_FORMAT_YUV);
    const QueuedFrame in{d_frame, step, rows, cols, timestamp, format};
    { const int lrc = st->ensure_lens(rows, cols); if (lrc != LVK_HIP_OK) return lrc; }
    static const WarpMeshF identity_mesh(2, 2);
    const uint8_t bg[3] = {(uint8_t)st->s.background[0], (uint8_t)st->s.background[1], (uint8_t)st->s.background[2]};

    auto enqueue = [&]() {
        if (st->queue.size() == st->queue_capacity) { if (released) *released = st->queue.front().d_ptr; st->queue.pop_front(); }
        st->queue.push_back(in);
    };
    auto emit = [&](const WarpMeshF* mesh) -> int {
        const QueuedFrame f = st->queue.front();
        st->queue.pop_front();
        LVK_HIP_REQUIRE(ctx, d_out != nullptr && out_step >= 3 * f.cols);
        int rc = LVK_HIP_OK;
        // overlap mode: the delayed frame was pushed >= 1 push ago and the tracker has synchronised the main stream since,
        // so the remap may run on its own stream concurrently with the next frame's tracking
        const bool side = st->overlap && mesh && st->s.stabilize_output;
        hipStream_t rs = side ? st->remap_stream : ctx->stream;
        // The persistent grid (4 remap blocks per CU) leaves room for the tracker's blocks of the NEXT frame; 5 or 6 starve them (8 450 /
        // 7 700 instead of 8 780 frames/s).  That only matters to a caller that runs free: one that waits for every frame (the previous
        // push ended long ago and the bulk stream is idle) gets the full grid -- the remap then has the GPU to itself (p50 latency -6 %).
        static const int pinned = [] { const char* e = std::getenv("LVK_HIP_CO_BLOCKS"); return e ? std::atoi(e) : 0; }();      // experiments: blocks per CU, always
        const bool persistent = side && (pinned != 0 || st->caller_runs_free);
        // a remap whose stores cross the host link (lvk_hip_stab_push_yuv420_host) is bound by the link, not by the chip: ONE block per CU
        // for a free-running caller -- measured (two upload streams at the time) 2 800 frames/s against 2 560 with the 4 blocks per CU of a device-resident stream (the
        // stores of more blocks only fill the link's write queue sooner, which stalls the tracker's kernels), 2 450 with one per two CUs
        ctx->co_blocks_per_cu = pinned != 0 ? pinned : ((persistent && st->host_direct_now) ? 1 : 0);
        if (side && (rc = st->bulk_stream_sees_caller_work()) != LVK_HIP_OK) return rc;
        if (st->remap_wait) { const hipEvent_t e = st->remap_wait; st->remap_wait = nullptr; LVK_HIP_CHECK(ctx, hipStreamWaitEvent(rs, e, 0)); }
        st->trace.mark(HostTrace::EMIT_WAITS);
        const int pe = st->prof_begin(LVK_STAGE_REMAP, rs);
        if (st->lens && (f.rows != st->lens_rows || f.cols != st->lens_cols)) return ctx->fail(LVK_HIP_ERR_ARG, "frame size changed while a lens profile is set");
        if (mesh && o420 && o420->y)
        {
            rc = lvk_launch_warpmesh_apply_420(ctx, rs, f.d_ptr, f.step, f.rows, f.cols, o420->y, o420->y_step, o420->u, o420->u_step, o420->v, o420->v_step,
                                               o420->nv12, mesh->off.data(), mesh->rows, mesh->cols, bg, st->lens ? &st->lens_args : nullptr, persistent);
            o420->used = true;
        }
        else if (mesh) rc = lvk_launch_warpmesh_apply_lens(ctx, rs, f.d_ptr, f.step, f.rows, f.cols, d_out, out_step, mesh->off.data(), mesh->rows, mesh->cols, bg,
                                                      f.format == LVK_FORMAT_YUV ? 1 : 0, st->lens ? &st->lens_args : nullptr, persistent);
        else
        {
            hipError_t e = hipMemcpy2DAsync(d_out, out_step, f.d_ptr, f.step, (size_t)f.cols * 3, f.rows, hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) rc = ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
        }
        st->prof_end(pe, rs);
        st->trace.mark(HostTrace::EMIT_KERNEL);
        if (rc != LVK_HIP_OK) return rc;
        if (produced) *produced = 1;
        if (out_timestamp) *out_timestamp = f.ts;                                         // WarpMesh.cpp:221-222
        if (side && st->pool_frames)
        {
            // 4:2:0 path in overlap mode: the slot is next written by an ingest -- on this same stream, i.e. after the remap that is
            // reading it now (stream order is all the protection it needs), or on the tracking stream behind this event
            if (st->tracker_ingest_capable)
            {
                const int si = st->slot_index(f.d_ptr);
                if (si >= 0)
                {
                    if (!st->slot_read_done[(size_t)si]) { hipError_t e = hipEventCreateWithFlags(&st->slot_read_done[(size_t)si], hipEventDisableTiming); if (e != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
                    hipError_t e = hipEventRecord(st->slot_read_done[(size_t)si], rs);
                    if (e != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
                    st->slot_read_armed[(size_t)si] = 1;
                }
            }
            if (released) *released = f.d_ptr;
        }
        else if (side)
        {
            // the frame stays borrowed until its remap has finished: hand back the previous one instead
            const int slot = st->remap_slot; st->remap_slot ^= 1;
            hipError_t e = hipEventRecord(st->remap_done[slot], rs);
            if (e != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
            if (st->pending_release)
            {
                if ((e = hipEventSynchronize(st->remap_done[st->pending_slot])) != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
                if (released) *released = st->pending_release;
            }
            st->pending_release = f.d_ptr; st->pending_slot = slot;
        }
        else if (released) *released = f.d_ptr;
        return LVK_HIP_OK;
    };

    if (!st->s.stabilize_output)                                                          // StabilizationFilter.cpp:77-95
    {
        enqueue();
        if (st->queue.size() != st->queue_capacity) return LVK_HIP_OK;
        return emit(st->s.crop_to_stable_region ? &st->smoother.scene_crop() : st->lens ? &identity_mesh : nullptr);
    }

    st->trace.mark(HostTrace::ENTER);
    WarpMeshF motion(st->s.motion_height, st->s.motion_width);                            // m_NullMotion
    WarpMeshF est; bool have = false;
    if (st->post_error) { st->post_error = false; return st->fail(LVK_HIP_ERR_RUNTIME, "GPU-side fast_filter disagrees with the host's"); }
    int rc = st->track(in, luma, luma_step, luma_pix, luma_channel, est, have);
    if (rc != LVK_HIP_OK) { st->finish_post(); return rc; }
    if (have) motion = est;

    // quality assurance (StabilizationFilter.cpp:101-115)
    const float tq = st->tracking_stability;
    st->scene_quality = st->scene_quality + QA_UPDATE_RATE * (tq - st->scene_quality);
    if (tq < st->s.min_tracking_quality) st->trust = 0.0f;
    else if (st->scene_quality < st->s.min_scene_quality) st->trust = step_toward(st->trust, 0.0f, QA_BLEND_STEP);
    else st->trust = step_toward(st->trust, 1.0f, QA_BLEND_STEP);
    motion.scale(st->trust);
    st->last_motion = motion;

    enqueue();
    WarpMeshF correction = st->smoother.next(motion);
    if (st->queue.size() != st->queue_capacity) return LVK_HIP_OK;                          // !ready(): output.release()
    if (st->s.crop_to_stable_region) correction += st->smoother.scene_crop();
    st->last_correction = correction;
    st->trace.mark(HostTrace::SMOOTH);
    const int erc = emit(&correction);
    st->trace.mark(HostTrace::REMAP_LAUNCH);
    return erc;                                                                             // finish_post(): at the next push, or when the lists are read
}

// a free-running caller: the bulk stream still busy with the previous remap, or this push beginning within 15 us of the previous one's return
// (a caller that waits for its frames synchronises and reads back in between: at least a remap's duration)
bool lvk_hip_stab::caller_free_running_now()
{
    bulk_busy_at_push = false;
    if (overlap && remap_stream)
    {
        const hipError_t q = hipStreamQuery(remap_stream);
        if (q != hipSuccess) (void)hipGetLastError();
        bulk_busy_at_push = q == hipErrorNotReady;
    }
    return bulk_busy_at_push || (last_push_end.time_since_epoch().count() != 0 && std::chrono::steady_clock::now() - last_push_end < std::chrono::microseconds(15));
}

int lvk_hip_stab::ensure_hostio(int rows, int cols)
{
    HostIO& h = hostio;
    if (h.rows == rows && h.cols == cols && h.up) return LVK_HIP_OK;
    LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (remap_stream) LVK_HIP_CHECK(ctx, hipStreamSynchronize(remap_stream));
    // (LVK_HIP_HOST_SINK=copy: the last emitted frame of the old size may not have been handed to the copy engine yet -- it was reported as
    //  produced, so it goes out before its staging planes are freed)
    { const int frc = flush_download(true); if (frc != LVK_HIP_OK) return frc; }
    free_hostio();
    const size_t bytes = (size_t)rows * cols + 2 * (size_t)((rows + 1) / 2) * ((cols + 1) / 2);
    for (auto& p : h.d_in) LVK_HIP_CHECK(ctx, hipMalloc(&p, bytes));
    for (auto& p : h.d_out) LVK_HIP_CHECK(ctx, hipMalloc(&p, bytes));
    // The streams that exist are the streams that are used: every stream of the process is a queue the runtime maps onto its few hardware
    // queues, and a transfer stream that lands on the hardware queue of the caller's stream stalls the tracker's kernels behind its copies
    // (measured, 4K free running with look-ahead: 3 140-3 190 frames/s with ONE upload stream, 2 700-2 850 with two, 2 040-2 130 with a third
    // side stream next to them).  The second upload stream (chroma of a frame pushed without look-ahead) and the download streams
    // (LVK_HIP_HOST_SINK=copy) are made on first use.  [The mechanism was narrowed down with scripts/sdma_interference_probe.py and the timelines
    // G-H of profiles/r03_host_feed_timeline.txt: whichever kernel comes first on the tracking stream after a look-ahead upload has started
    // -- the downscale, the flow kernel, a 5 KB copy, even a kernel that only stores its arguments -- ends ~220 us after that upload began.]
    { const int rcs = host_stream(h.up); if (rcs != LVK_HIP_OK) return rcs; }
    ctx->sync_hooks.emplace_back((void*)this, [this]() { return flush_download(true); });          // lvk_hip_sync() covers the transfers
    for (int i = 0; i < HostIO::K_IN; i++)
    {
        LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&h.y_done[i], hipEventDisableTiming));
        LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&h.c_done[i], hipEventDisableTiming));
    }
    for (int i = 0; i < HostIO::K_OUT; i++)
    {
        LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&h.out_ready[i], hipEventDisableTiming));
        LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&h.down_done[i], hipEventDisableTiming));
        h.down_armed[i] = false;
    }
    h.rows = rows; h.cols = cols; h.in_next = h.out_next = 0; h.last_down = -1; h.ahead.clear();
    return LVK_HIP_OK;
}

void lvk_hip_stab::free_hostio()
{
    HostIO& h = hostio;
    auto& aux = ctx->aux_streams;
    { auto& hooks = ctx->sync_hooks; hooks.erase(std::remove_if(hooks.begin(), hooks.end(), [this](const auto& kv) { return kv.first == (void*)this; }), hooks.end()); }
    h.pending.valid = false;
    for (hipStream_t s : {h.up, h.down, h.up2, h.down2})
        if (s) { (void)hipStreamSynchronize(s); aux.erase(std::remove(aux.begin(), aux.end(), s), aux.end()); (void)hipStreamDestroy(s); }
    h.up = h.down = h.up2 = h.down2 = nullptr;
    for (auto& p : h.d_in) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto& p : h.d_out) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto* arr : {h.y_done, h.c_done}) for (int i = 0; i < HostIO::K_IN; i++) if (arr[i]) { (void)hipEventDestroy(arr[i]); arr[i] = nullptr; }
    for (auto* arr : {h.out_ready, h.down_done}) for (int i = 0; i < HostIO::K_OUT; i++) if (arr[i]) { (void)hipEventDestroy(arr[i]); arr[i] = nullptr; }
    h.rows = h.cols = 0;
}

// The uploads of one host frame into staging slot k, on the upload stream.
//   * pushed now (the caller waits for this frame): luma, event, chroma, event -- the tracker starts on the luma plane while the chroma planes
//     are still on the link;
//   * announced ahead (the link is the bottleneck, not this frame's latency): ONE copy when the planes are contiguous.  A copy engine
//     that has to wait for anything but its own previous copy -- here: the event between the two copies -- is restarted by the
//     runtime's signal handler 60-80 us late (timeline in profiles/r03_host_feed_timeline.txt): 341 us of link time per frame instead of 265.
int lvk_hip_stab::host_upload(const void* h_y, int y_step, const void* h_u, int u_step, const void* h_v, int v_step, int nv12, int rows, int cols, int k, bool ahead)
{
    HostIO& io = hostio;
    const int crows = rows / 2, ccols = nv12 ? cols : cols / 2;
    uint8_t* d_y = (uint8_t*)io.d_in[k];
    uint8_t* d_u = d_y + (size_t)rows * cols;
    uint8_t* d_v = nv12 ? d_u : d_u + (size_t)crows * ccols;
    int rc;
    // (the staging slot is free: the kernels that read it -- downscale, conversion -- were complete when the push that used it returned)
    auto copy_plane = [&](void* dst, int dpitch, const void* src, int spitch, int width, int height, hipStream_t s) -> hipError_t {
        if (spitch == width && dpitch == width) return hipMemcpyAsync(dst, src, (size_t)width * height, hipMemcpyHostToDevice, s);
        return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, s);
    };
    const bool contiguous = y_step == cols && u_step == ccols && (const uint8_t*)h_u == (const uint8_t*)h_y + (size_t)rows * cols &&
                            (nv12 || (v_step == ccols && (const uint8_t*)h_v == (const uint8_t*)h_u + (size_t)crows * ccols));
    if (ahead && contiguous)
    {
        const size_t bytes = (size_t)rows * cols + (size_t)(nv12 ? 1 : 2) * crows * ccols;
        // ONE upload stream.  (hipMemcpyAsync blocks the host while an earlier copy of the same stream is still in flight, which two alternating
        // streams avoid -- LVK_HIP_HOST_UP2=1 --, but the second stream costs more than that wait: see ensure_hostio.)
        if (host_up2 && (k & 1)) { if ((rc = host_stream(io.up2)) != LVK_HIP_OK) return rc; }
        hipStream_t us = (host_up2 && (k & 1)) ? io.up2 : io.up;
        if (host_h2d_blocks > 0 && ((uintptr_t)h_y & 15) == 0) { if ((rc = lvk_launch_copy_bytes(ctx, us, d_y, h_y, bytes, host_h2d_blocks)) != LVK_HIP_OK) return rc; }
        else LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_y, h_y, bytes, hipMemcpyHostToDevice, us));
        LVK_HIP_CHECK(ctx, hipEventRecord(io.c_done[k], us));
        io.y_is_c[k] = true;
        return LVK_HIP_OK;
    }
    io.y_is_c[k] = false;
    hipStream_t cs = io.up;                                                          // luma and chroma of a frame pushed now: one stream, in order
    const bool kernel_up = host_h2d_blocks > 0 && y_step == cols && ((uintptr_t)h_y & 15) == 0 && ((uintptr_t)h_u & 15) == 0 && ((size_t)rows * cols) % 16 == 0;
    if (kernel_up) { if ((rc = lvk_launch_copy_bytes(ctx, io.up, d_y, h_y, (size_t)rows * cols, host_h2d_blocks)) != LVK_HIP_OK) return rc; }
    else LVK_HIP_CHECK(ctx, copy_plane(d_y, cols, h_y, y_step, cols, rows, io.up));
    LVK_HIP_CHECK(ctx, hipEventRecord(io.y_done[k], io.up));
    if (!nv12 && u_step == ccols && v_step == ccols && (const uint8_t*)h_v == (const uint8_t*)h_u + (size_t)crows * ccols)
    {
        if (kernel_up) { if ((rc = lvk_launch_copy_bytes(ctx, cs, d_u, h_u, 2 * (size_t)crows * ccols, host_h2d_blocks)) != LVK_HIP_OK) return rc; }
        else LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_u, h_u, 2 * (size_t)crows * ccols, hipMemcpyHostToDevice, cs));      // U | V contiguous: one copy
    }
    else
    {
        LVK_HIP_CHECK(ctx, copy_plane(d_u, ccols, h_u, u_step, ccols, crows, cs));
        if (!nv12) LVK_HIP_CHECK(ctx, copy_plane(d_v, ccols, h_v, v_step, ccols, crows, cs));
    }
    LVK_HIP_CHECK(ctx, hipEventRecord(io.c_done[k], cs));
    return LVK_HIP_OK;
}

// Deferred download (LVK_HIP_HOST_SINK=copy): the D2H copy of an emitted frame is handed to the runtime only once the remap that wrote the
// device planes is KNOWN to be complete, on a stream with nothing pending -- a copy that has to wait for a kernel is performed by the
// runtime with a blit kernel (which saturates the link's write queue and stalls every other kernel), an unencumbered one by a copy engine.
// wait = false: only if the remap has finished (polled at the start and at the end of the following push); true: wait for it.
int lvk_hip_stab::flush_download(bool wait)
{
    HostIO& io = hostio;
    if (!io.pending.valid) return LVK_HIP_OK;
    const int j = io.pending.slot;
    const hipError_t q = hipEventQuery(io.out_ready[j]);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); if (!wait) return LVK_HIP_OK; LVK_HIP_CHECK(ctx, hipEventSynchronize(io.out_ready[j])); }
    else if (q != hipSuccess) return fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(q));
    const int rows = io.rows, cols = io.cols, nv12 = io.pending.nv12, crows = rows / 2, ccols = nv12 ? cols : cols / 2;
    uint8_t* o_y = (uint8_t*)io.d_out[j]; uint8_t* o_u = o_y + (size_t)rows * cols; uint8_t* o_v = nv12 ? o_u : o_u + (size_t)crows * ccols;
    { int rcs; if ((rcs = host_stream(io.down)) != LVK_HIP_OK || (rcs = host_stream(io.down2)) != LVK_HIP_OK) return rcs; }
    hipStream_t ds = (j & 1) ? io.down2 : io.down;           // (hipMemcpyAsync blocks the host while an earlier copy of the same stream is in flight)
    auto copy_plane = [&](void* dst, int dpitch, const void* src, int spitch, int width, int height) -> hipError_t {
        if (spitch == width && dpitch == width) return hipMemcpyAsync(dst, src, (size_t)width * height, hipMemcpyDeviceToHost, ds);
        return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, ds);
    };
    const auto& p = io.pending;
    {
        // a destination the previous download (on the other stream) may still be writing: order behind it
        const uint8_t* lo = (const uint8_t*)p.y; const uint8_t* hi = lo + (size_t)p.ys * rows;
        if (io.last_down >= 0 && io.last_down != j && io.down_armed[io.last_down] && io.last_dst_lo < hi && lo < io.last_dst_hi)
            LVK_HIP_CHECK(ctx, hipStreamWaitEvent(ds, io.down_done[io.last_down], 0));
        io.last_dst_lo = lo; io.last_dst_hi = hi;
    }
    const bool contiguous = p.ys == cols && p.us == ccols && (uint8_t*)p.u == (uint8_t*)p.y + (size_t)rows * cols &&
                            (nv12 || (p.vs == ccols && (uint8_t*)p.v == (uint8_t*)p.u + (size_t)crows * ccols));
    if (contiguous) LVK_HIP_CHECK(ctx, hipMemcpyAsync(p.y, o_y, (size_t)rows * cols + (size_t)(nv12 ? 1 : 2) * crows * ccols, hipMemcpyDeviceToHost, ds));
    else
    {
        LVK_HIP_CHECK(ctx, copy_plane(p.y, p.ys, o_y, cols, cols, rows));
        LVK_HIP_CHECK(ctx, copy_plane(p.u, p.us, o_u, ccols, ccols, crows));
        if (!nv12) LVK_HIP_CHECK(ctx, copy_plane(p.v, p.vs, o_v, ccols, ccols, crows));
    }
    LVK_HIP_CHECK(ctx, hipEventRecord(io.down_done[j], ds));
    io.down_armed[j] = true; io.last_down = j;
    io.pending.valid = false;
    return LVK_HIP_OK;
}

// Announced frames that will not be pushed (the caller stopped, seeked or restarted): their uploads are waited for -- the staging slots are
// rewritten by the next upload on the same stream anyway, but the caller's planes must not be read after this returns -- and forgotten.
int lvk_hip_stab::cancel_lookahead()
{
    HostIO& io = hostio;
    for (hipStream_t us : {io.up, io.up2}) if (us) LVK_HIP_CHECK(ctx, hipStreamSynchronize(us));
    io.ahead.clear();
    for (hipEvent_t& e : ingest_wait) e = nullptr;
    return LVK_HIP_OK;
}

int lvk_hip_stab::ensure_pool(int rows, int cols)
{
    const size_t want = (size_t)s.predictive_samples + 4;
    if (rows == pool_rows && cols == pool_cols && pool_all.size() >= want) return LVK_HIP_OK;
    if (rows != pool_rows || cols != pool_cols)
    {
        // new geometry: queued pool frames of the old size stay valid until they are emitted, so only grow lazily
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (remap_stream) LVK_HIP_CHECK(ctx, hipStreamSynchronize(remap_stream));
        // A resolution change in the middle of a 4:2:0 stream: the frames still queued live in pool slots of the old geometry, and the
        // caller's output planes have the new one -- they are DROPPED (the delay builds up again, `produced` stays 0 for predictive_samples
        // pushes); tracker and path smoother carry on, as in the reference, which would also still emit those frames at their old size
        // (StabilizationFilter.cpp:118-131 keeps whole VideoFrames in its queue).
        if (!queue.empty() && !pool_all.empty()) { queue.clear(); pending_release = nullptr; pending_slot = -1; }
        free_pool();
        pool_rows = rows; pool_cols = cols;
        LVK_HIP_CHECK(ctx, hipMalloc(&pool_out, (size_t)rows * cols * 3));
    }
    while (pool_all.size() < want)
    {
        void* p = nullptr;
        LVK_HIP_CHECK(ctx, hipMalloc(&p, (size_t)rows * cols * 3));
        pool_all.push_back(p); pool_free.push_back(p); slot_read_done.push_back(nullptr); slot_read_armed.push_back(0);
    }
    return LVK_HIP_OK;
}

void lvk_hip_stab::free_pool()
{
    for (void* p : pool_all) (void)hipFree(p);
    for (hipEvent_t e : slot_read_done) if (e) (void)hipEventDestroy(e);
    pool_all.clear(); pool_free.clear(); slot_read_done.clear(); slot_read_armed.clear();
    if (pool_out) { (void)hipFree(pool_out); pool_out = nullptr; }
    pool_rows = pool_cols = 0;
}

extern "C" {

int lvk_hip_stab_push(lvk_hip_stab* st, const void* d_frame, int step, int rows, int cols, uint64_t timestamp, int format,
                      void* d_out, int out_step, int* produced, uint64_t* out_timestamp, const void** released)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    st->trace.begin();
    st->prof_tick++;
    st->push_seq++;
    struct AnnouncementEnds { lvk_hip_stab* s; ~AnnouncementEnds() { s->ahead_announced = lvk_hip_stab::LumaAhead(); } } announcement_ends{st};
    if (st->queue.empty()) st->queue_kind = 0;
    if (st->queue_kind == 2) return st->fail(LVK_HIP_ERR_ARG, "frames of lvk_hip_stab_push_yuv420 are still queued: restart() before switching to lvk_hip_stab_push");
    st->queue_kind = 1;
    st->pool_frames = false;
    int rc = st->mark_caller_work();
    if (rc != LVK_HIP_OK) return rc;
    rc = push_impl(st, d_frame, step, rows, cols, timestamp, format, d_frame, step, 3, d_out, out_step, produced, out_timestamp, released);
    if (released && !*released && !st->orphaned.empty()) { *released = st->orphaned.front(); st->orphaned.pop_front(); }
    st->trace.mark(HostTrace::EXIT);
    st->last_push_end = std::chrono::steady_clock::now();
    return rc;
}

// The OBS asynchronous path in one call: I4XXIngest / NV12Ingest::to_ocl -> StabilizationFilter::filter -> ::to_obs
// (Modules/OBS-Plugin/Interop/VisionFilter.cpp:151-212, FrameIngest.cpp:494-602).  Planar (or NV12) 4:2:0 in, 4:2:0 out;
// the packed 8UC3 frames the filter works on live in an internal pool (predictive_samples + 4 frames).  The input planes
// are consumed before the call returns; the output planes are complete after lvk_hip_sync().
int lvk_hip_stab_push_yuv420(lvk_hip_stab* st, const void* d_y, int y_step, const void* d_u, int u_step, const void* d_v, int v_step, int nv12,
                             int rows, int cols, uint64_t timestamp,
                             void* o_y, int oy_step, void* o_u, int ou_step, void* o_v, int ov_step,
                             int* produced, uint64_t* out_timestamp)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    lvk_hip_ctx* ctx = st->ctx;
    st->trace.begin();
    st->prof_tick++;
    st->push_seq++;
    struct AnnouncementEnds { lvk_hip_stab* s; ~AnnouncementEnds() { s->ahead_announced = lvk_hip_stab::LumaAhead(); } } announcement_ends{st};
    if (produced) *produced = 0;
    if (st->queue.empty()) st->queue_kind = 0;
    if (st->queue_kind == 1) return st->fail(LVK_HIP_ERR_ARG, "borrowed frames of lvk_hip_stab_push are still queued: restart() before switching to lvk_hip_stab_push_yuv420");
    st->queue_kind = 2;
    int rc = st->ensure_pool(rows, cols);
    if (rc != LVK_HIP_OK) return rc;
    if ((rc = st->mark_caller_work()) != LVK_HIP_OK) return rc;
    if (st->pool_free.empty())
    {
        // frames dropped by restart() / a shrinking queue never came back through *released: reclaim them
        for (void* p : st->pool_all)
        {
            bool used = (p == st->pending_release);
            for (const QueuedFrame& q : st->queue) used = used || (q.d_ptr == p);
            if (!used) st->pool_free.push_back(p);
        }
        LVK_HIP_REQUIRE(ctx, !st->pool_free.empty());
    }
    void* slot = st->pool_free.front(); st->pool_free.pop_front();
    // The packed frame is only read by the remap `predictive_samples` pushes later (the tracker reads the luma plane itself), so in
    // overlap mode the conversion runs on the remap stream, off the tracker's critical path; same-stream order protects the slot.
    const bool side_ingest = st->overlap && st->s.stabilize_output;
    // a delayed frame (its remap is launched after later synchronisations of the tracking stream) may be converted on either stream: track() decides
    st->tracker_ingest_capable = side_ingest && st->queue_capacity > 1;
    st->ingest_on_tracker = false;
    int pe = 0;
    auto do_ingest = [=]() -> int {
        const bool on_tracker = side_ingest && st->ingest_on_tracker;
        hipStream_t is = (side_ingest && !on_tracker) ? st->remap_stream : ctx->stream;
        if (side_ingest && !on_tracker) { const int w = st->bulk_stream_sees_caller_work(); if (w != LVK_HIP_OK) return w; }
        if (on_tracker)
        {
            const int si = st->slot_index(slot);
            if (si >= 0 && st->slot_read_armed[(size_t)si]) { st->slot_read_armed[(size_t)si] = 0; LVK_HIP_CHECK(ctx, hipStreamWaitEvent(is, st->slot_read_done[(size_t)si], 0)); }
        }
        // host-resident frames: the planes are still arriving on the upload stream
        for (hipEvent_t& e : st->ingest_wait) if (e) { LVK_HIP_CHECK(ctx, hipStreamWaitEvent(is, e, 0)); e = nullptr; }
        const int pi = st->prof_begin(LVK_STAGE_INGEST, is);
        const int r = lvk_launch_ingest_yuv420(ctx, is, d_y, y_step, d_u, u_step, d_v, v_step, nv12, rows, cols, slot, 3 * cols);
        st->prof_end(pi, is);
        if (r != LVK_HIP_OK) return r;
        if (side_ingest)
        {
            if (!st->ingest_done) LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&st->ingest_done, hipEventDisableTiming));
            LVK_HIP_CHECK(ctx, hipEventRecord(st->ingest_done, is));
        }
        return LVK_HIP_OK;
    };
    // Overlap mode: the conversion is not on the tracker's critical path (it runs behind the previous remap on the bulk stream), so its
    // launch and its event record wait until the tracker's kernels are on their way -- track() calls it after its last launch.
    // (8.05k -> 8.17k frames/s, p50 latency -7 us.)
    if (side_ingest) st->deferred_ingest = do_ingest;
    else { rc = do_ingest(); if (rc != LVK_HIP_OK) { st->pool_free.push_back(slot); return rc; } }
    int prod = 0; const void* released = nullptr;
    st->pool_frames = side_ingest;
    OutPlanes420 o420{o_y, oy_step, o_u, ou_step, o_v, ov_step, nv12, false};
    if (!(o_y && o_u && (nv12 || o_v))) o420.y = nullptr;
    rc = push_impl(st, slot, 3 * cols, rows, cols, timestamp, LVK_FORMAT_YUV, d_y, y_step, 1, st->pool_out, 3 * cols, &prod, out_timestamp, &released, &o420);
    if (st->deferred_ingest) { const int r2 = st->run_deferred_ingest(); if (rc == LVK_HIP_OK) rc = r2; }      // (track() returned before its launches)
    if (released) st->pool_free.push_back(const_cast<void*>(released));
    if (side_ingest)
    {
        // (the next frame's pyramid is on its way behind this chain: nothing will shadow the list bookkeeping at the start of the next push --
        //  track() does it behind the two launches that are no longer there --, so it runs here, while the conversion is still running anyway)
        if (st->ahead_built_for == st->push_seq + 1 && !st->early_post_off) st->finish_post();
        st->trace.mark(HostTrace::EXIT_PRE);
        // contract: the caller's planes are consumed when the call returns (the conversion started ~a tracking pass ago).  An event, not
        // hipStreamSynchronize: synchronising the bulk stream itself costs ~10 us of host time even when it is idle (measured).
        LVK_HIP_CHECK(ctx, hipEventSynchronize(st->ingest_done));
        st->trace.mark(HostTrace::EXIT_WAIT);
    }
    // same contract without a tracker pass (delay-only mode, stabilize_output off): nothing has synchronised behind the conversion yet
    else if (!st->s.stabilize_output) LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (rc != LVK_HIP_OK) return rc;
    if (prod && o420.used) { if (produced) *produced = 1; }                // the fused remap + egress kernel has written the planes
    else if (prod)
    {
        // the emitted frame has the geometry of the pool (all pooled frames share it)
        LVK_HIP_REQUIRE(ctx, o_y && o_u && (nv12 || o_v));
        hipStream_t es = (st->overlap && st->s.stabilize_output) ? st->remap_stream : ctx->stream;
        if (st->remap_wait) { const hipEvent_t e = st->remap_wait; st->remap_wait = nullptr; LVK_HIP_CHECK(ctx, hipStreamWaitEvent(es, e, 0)); }
        pe = st->prof_begin(LVK_STAGE_EGRESS, es);
        rc = lvk_launch_egress_yuv420(ctx, es, st->pool_out, 3 * cols, rows, cols, o_y, oy_step, o_u, ou_step, o_v, ov_step, nv12);
        st->prof_end(pe, es);
        if (rc != LVK_HIP_OK) return rc;
        if (produced) *produced = 1;
    }
    st->trace.mark(HostTrace::EXIT);
    st->last_push_end = std::chrono::steady_clock::now();
    return LVK_HIP_OK;
}

// Look-ahead for DEVICE-resident frames (a caller that has the next frame in HBM already: the reader thread of VideoFilter::stream runs ahead of
// the filter thread, Filters/VideoFilter.cpp:62-209; a transcoder with its clip resident).  Announce frame n + 1, THEN push frame n: the push
// puts the downscale and the pyramid of frame n + 1 on the tracking stream behind its own chain, where the GPU runs them while the host has its
// turn (results, path smoother, remap launch), and the push of frame n + 1 starts at the optical flow.  Only the luma is read ahead (the Y plane /
// channel 0 of a packed YUV frame / the grey value of BGR, RGB); it must not change between this call and the return of the push that carries it.
// The announcement holds for the very next push only: a push that carries other planes, an other geometry or that does not track (the first
// frame, a restart, stabilize_output off) simply works as if nothing had been announced.  Same pixels either way.
static int stab_announce(lvk_hip_stab* st, const void* luma, int step, int pix, int channel, int rows, int cols)
{
    st->ahead_announced.luma = luma; st->ahead_announced.step = step; st->ahead_announced.pix = pix; st->ahead_announced.channel = channel;
    st->ahead_announced.rows = rows; st->ahead_announced.cols = cols;
    return LVK_HIP_OK;
}

int lvk_hip_stab_prefetch(lvk_hip_stab* st, const void* d_frame, int step, int rows, int cols, int format)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_hip_ctx* ctx = st->ctx;
    LVK_HIP_REQUIRE(ctx, d_frame && rows > 0 && cols > 0 && step >= 3 * cols);
    LVK_HIP_REQUIRE(ctx, format == LVK_FORMAT_YUV || format == LVK_FORMAT_BGR || format == LVK_FORMAT_RGB);
    return stab_announce(st, d_frame, step, 3, format == LVK_FORMAT_YUV ? 0 : (format == LVK_FORMAT_BGR ? -1 : -2), rows, cols);
}

int lvk_hip_stab_prefetch_yuv420(lvk_hip_stab* st, const void* d_y, int y_step, const void* d_u, int u_step, const void* d_v, int v_step, int nv12, int rows, int cols)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_hip_ctx* ctx = st->ctx;
    LVK_HIP_REQUIRE(ctx, d_y && d_u && (nv12 || d_v) && rows > 0 && cols > 0 && rows % 2 == 0 && cols % 2 == 0);
    LVK_HIP_REQUIRE(ctx, y_step >= cols && u_step >= (nv12 ? cols : cols / 2) && (nv12 || v_step >= cols / 2));
    return stab_announce(st, d_y, y_step, 1, 0, rows, cols);
}

long long lvk_hip_stab_lookahead_frames(lvk_hip_stab* st) { return st ? (long long)st->lookahead_frames : 0; }

// Look-ahead for streaming callers (the reader thread of VideoFilter::stream uploads frames ahead of the filter thread,
// Filters/VideoFilter.cpp:62-209): starts the upload of the planes that the NEXT lvk_hip_stab_push_yuv420_host call will push, so that the
// link is busy with frame n + 1 while frame n is tracked: announce frame n + 1, THEN push frame n.  Announced frames are pushed in order; at
// most two may be outstanding.  The planes stay the caller's until their push has returned.
int lvk_hip_stab_prefetch_yuv420_host(lvk_hip_stab* st, const void* h_y, int y_step, const void* h_u, int u_step, const void* h_v, int v_step, int nv12, int rows, int cols)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    lvk_hip_ctx* ctx = st->ctx;
    LVK_HIP_REQUIRE(ctx, h_y && h_u && (nv12 || h_v) && rows > 0 && cols > 0 && rows % 2 == 0 && cols % 2 == 0);
    LVK_HIP_REQUIRE(ctx, y_step >= cols && u_step >= (nv12 ? cols : cols / 2) && (nv12 || v_step >= cols / 2));
    int rc;
    if ((rc = st->require_pinned_planes(h_y, y_step, h_u, u_step, h_v, v_step, nv12, rows, cols, "lvk_hip_stab_prefetch_yuv420_host")) != LVK_HIP_OK) return rc;
    if ((rc = st->ensure_hostio(rows, cols)) != LVK_HIP_OK) return rc;
    lvk_hip_stab::HostIO& io = st->hostio;
    // two staging slots: the frame being pushed and the one on the link -- at most two announced frames that have not been pushed yet
    LVK_HIP_REQUIRE(ctx, io.ahead.size() < (size_t)lvk_hip_stab::HostIO::K_IN);
    const int k = io.in_next; io.in_next = (k + 1) % lvk_hip_stab::HostIO::K_IN;
    if ((rc = st->host_upload(h_y, y_step, h_u, u_step, h_v, v_step, nv12, rows, cols, k, true)) != LVK_HIP_OK) return rc;
    io.ahead.push_back({k, {h_y, h_u, nv12 ? h_u : h_v}, rows, cols, nv12 ? 1 : 0});
    return LVK_HIP_OK;
}

// Forget the announced frames that have not been pushed (lvk_hip_stab_restart does the same): for a caller that announced frame n + 1 and
// then stops, seeks or switches to other buffers.  Returns once the uploads no longer read the caller's planes.
int lvk_hip_stab_prefetch_cancel(lvk_hip_stab* st)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    st->forget_device_lookahead();
    return st->cancel_lookahead();
}

// Host-resident frames: FrameIngest::upload_planes -> to_ocl -> StabilizationFilter::filter -> to_obs -> download_planes in one call
// (Modules/OBS-Plugin/Interop/FrameIngest.cpp:415-474,494-602, VisionFilter.cpp:151-212) -- SURVEY.md section 8d's metric ("p99 ms/frame
// including H2D of the input and D2H of the output when frames are host-resident").  h_* / oh_*: planes in PINNED host memory
// (lvk_hip_host_malloc, hipHostMalloc, hipHostRegister).  What the link gives (profiles/r03_pcie_probe.txt): 55 GB/s one way, 46.8 GB/s
// each way with ONE copy-engine stream per direction at once, 31 with two per direction -- so:
//   in:  one upload stream; the luma plane goes first and the tracker's stream waits for IT only (downscale, pyramid, flow and the motion
//        estimate run while the chroma planes are still on the link); the 4:2:0 conversion waits for both.  Planes that are contiguous in
//        host memory (the OBS frame layout, FrameIngest.cpp:441-453 "uploads are done in bulk") travel as one copy each.
//   out: a caller that waits for every frame gets the planes written by the remap kernel ITSELF into the pinned host planes (zero copy:
//        the stores go over the link as they are produced -- no remap -> download serialisation, ~0.1 ms less per frame); a caller that
//        runs free gets remap -> device planes -> one download on the download stream behind an event (a copy engine both ways is the
//        faster pair when the link is saturated: 46.8 vs 43 GB/s).  Same pixels either way.
// Input planes are consumed when the call returns; output planes are complete after lvk_hip_sync().
int lvk_hip_stab_push_yuv420_host(lvk_hip_stab* st, const void* h_y, int y_step, const void* h_u, int u_step, const void* h_v, int v_step, int nv12,
                                  int rows, int cols, uint64_t timestamp,
                                  void* oh_y, int oy_step, void* oh_u, int ou_step, void* oh_v, int ov_step,
                                  int* produced, uint64_t* out_timestamp)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    lvk_hip_ctx* ctx = st->ctx;
    if (produced) *produced = 0;
    LVK_HIP_REQUIRE(ctx, h_y && h_u && (nv12 || h_v) && rows > 0 && cols > 0 && rows % 2 == 0 && cols % 2 == 0);
    LVK_HIP_REQUIRE(ctx, y_step >= cols && u_step >= (nv12 ? cols : cols / 2) && (nv12 || v_step >= cols / 2));
    int rc;
    if ((rc = st->require_pinned_planes(h_y, y_step, h_u, u_step, h_v, v_step, nv12, rows, cols, "lvk_hip_stab_push_yuv420_host")) != LVK_HIP_OK) return rc;
    if (oh_y && oh_u && (nv12 || oh_v))
    {
        LVK_HIP_REQUIRE(ctx, oy_step >= cols && ou_step >= (nv12 ? cols : cols / 2) && (nv12 || ov_step >= cols / 2));
        if ((rc = st->require_pinned_planes(oh_y, oy_step, oh_u, ou_step, oh_v, ov_step, nv12, rows, cols, "lvk_hip_stab_push_yuv420_host (output)")) != LVK_HIP_OK) return rc;
    }
    if ((rc = st->ensure_hostio(rows, cols)) != LVK_HIP_OK) return rc;
    lvk_hip_stab::HostIO& io = st->hostio;
    if ((rc = st->flush_download(false)) != LVK_HIP_OK) return rc;
    auto tr_last = std::chrono::steady_clock::now();
    auto tr_mark = [&](int k) { if (!st->trace.on) return; const auto now = std::chrono::steady_clock::now(); st->host_trace_acc[k] += std::chrono::duration<double, std::micro>(now - tr_last).count(); tr_last = now; };
    const int crows = rows / 2, ccols = nv12 ? cols : cols / 2;                     // chroma plane geometry (bytes per row)
    int k;
    if (!io.ahead.empty())
    {
        // its upload has been under way since the look-ahead call; look-ahead frames are pushed in the order they were announced
        const auto a = io.ahead.front();
        if (!(a.key[0] == h_y && a.key[1] == h_u && a.key[2] == (nv12 ? h_u : h_v) && a.rows == rows && a.cols == cols && a.nv12 == (nv12 ? 1 : 0)))
            return st->fail(LVK_HIP_ERR_ARG, "lvk_hip_stab_push_yuv420_host: another frame has been announced (lvk_hip_stab_prefetch_yuv420_host) and not pushed yet -- "
                                             "announced frames are pushed in the order announced, and a frame pushed while announcements are outstanding must be the oldest of them");
        io.ahead.pop_front();
        k = a.slot;
    }
    else
    {
        k = io.in_next; io.in_next = (k + 1) % lvk_hip_stab::HostIO::K_IN;
        if ((rc = st->host_upload(h_y, y_step, h_u, u_step, h_v, v_step, nv12, rows, cols, k, false)) != LVK_HIP_OK) return rc;
    }
    uint8_t* d_y = (uint8_t*)io.d_in[k];
    uint8_t* d_u = d_y + (size_t)rows * cols;
    uint8_t* d_v = nv12 ? d_u : d_u + (size_t)crows * ccols;
    tr_mark(0);
    // what this call hands to the push it wraps (events to wait for, the sink hints) never outlives it, whichever way it returns
    struct ClearHooks
    {
        lvk_hip_stab* s;
        ~ClearHooks() { s->remap_wait = nullptr; s->ingest_wait[0] = s->ingest_wait[1] = nullptr; s->host_free_running_hint = false; s->host_direct_now = false; }
    } clear_hooks{st};
    LVK_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, io.y_is_c[k] ? io.c_done[k] : io.y_done[k], 0));           // the tracker needs the luma plane only
    st->ingest_wait[0] = io.y_is_c[k] ? nullptr : io.y_done[k]; st->ingest_wait[1] = io.c_done[k];

    // where the output planes are written: by the remap kernel itself, straight into the pinned host planes (measured, 4K, free running:
    // 2 800 frames/s against 2 490 for remap -> device planes -> download, whose D2H copy the runtime performs with a blit KERNEL that
    // saturates the link's write queue and stalls every other kernel's memory traffic while it runs -- timelines under profiles/).
    // LVK_HIP_HOST_SINK=copy keeps the download route for comparison.
    const bool have_out = oh_y && oh_u && (nv12 || oh_v);
    st->host_free_running_hint = st->caller_free_running_now() ||
                                 (io.last_end.time_since_epoch().count() != 0 && std::chrono::steady_clock::now() - io.last_end < std::chrono::microseconds(15));
    const bool direct = have_out && st->host_sink_mode != 2;
    const int j = io.out_next;
    uint8_t* o_y = nullptr; uint8_t* o_u = nullptr; uint8_t* o_v = nullptr;
    int oys = oy_step, ous = ou_step, ovs = ov_step;
    if (have_out && !direct)
    {
        o_y = (uint8_t*)io.d_out[j]; o_u = o_y + (size_t)rows * cols; o_v = nv12 ? o_u : o_u + (size_t)crows * ccols;
        oys = cols; ous = ccols; ovs = ccols;
        if (io.pending.valid && io.pending.slot == j) { if ((rc = st->flush_download(true)) != LVK_HIP_OK) return rc; }
        if (io.down_armed[j]) st->remap_wait = io.down_done[j];                      // the download that last read this slot
    }
    else if (have_out)
    {
        o_y = (uint8_t*)oh_y; o_u = (uint8_t*)oh_u; o_v = (uint8_t*)oh_v;
        // a download of an earlier frame may still be writing the caller's (possibly the same) host planes: the kernel's stores follow it
        if ((rc = st->flush_download(true)) != LVK_HIP_OK) return rc;
        if (io.last_down >= 0 && io.down_armed[io.last_down]) st->remap_wait = io.down_done[io.last_down];
    }
    int prod = 0;
    tr_mark(1);
    st->host_direct_now = direct;
    rc = lvk_hip_stab_push_yuv420(st, d_y, cols, d_u, ccols, d_v, ccols, nv12, rows, cols, timestamp, o_y, oys, o_u, ous, o_v, ovs, &prod, out_timestamp);
    tr_mark(2);
    // "consumed on return": the conversion (which waited for both uploads) has finished in every mode by now; the event costs nothing then
    LVK_HIP_CHECK(ctx, hipEventSynchronize(io.c_done[k]));
    tr_mark(3);
    if (rc != LVK_HIP_OK) return rc;
    if ((rc = st->flush_download(false)) != LVK_HIP_OK) return rc;                  // the previous frame's remap has usually finished by now
    if (prod && have_out && !direct)
    {
        if ((rc = st->flush_download(true)) != LVK_HIP_OK) return rc;               // (one deferred download at a time)
        io.out_next = (j + 1) % lvk_hip_stab::HostIO::K_OUT;
        hipStream_t os = (hipStream_t)lvk_hip_stab_output_stream(st);
        LVK_HIP_CHECK(ctx, hipEventRecord(io.out_ready[j], os));
        io.pending.valid = true; io.pending.slot = j; io.pending.y = oh_y; io.pending.u = oh_u; io.pending.v = oh_v;
        io.pending.ys = oy_step; io.pending.us = ou_step; io.pending.vs = ov_step; io.pending.nv12 = nv12 ? 1 : 0;
        io.down_armed[j] = false;
    }
    if (produced) *produced = prod;
    tr_mark(4); st->host_trace_n++;
    io.last_end = std::chrono::steady_clock::now();
    return LVK_HIP_OK;
}

// Pinned host memory for the planes of lvk_hip_stab_push_yuv420_host (what obs_source_frame buffers would be registered as)
int lvk_hip_host_malloc(lvk_hip_ctx* ctx, size_t bytes, void** h_ptr)
{
    if (!ctx || !h_ptr) return LVK_HIP_ERR_ARG;
    LVK_HIP_REQUIRE(ctx, bytes > 0);
    LVK_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    LVK_HIP_CHECK(ctx, hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    return LVK_HIP_OK;
}

int lvk_hip_host_free(lvk_hip_ctx* ctx, void* h_ptr)
{
    if (!ctx) return LVK_HIP_ERR_ARG;
    if (h_ptr) LVK_HIP_CHECK(ctx, hipHostFree(h_ptr));
    return LVK_HIP_OK;
}

int lvk_hip_stab_get_stats(const lvk_hip_stab* st, lvk_stab_stats* o)
{
    if (!st || !o) return LVK_HIP_ERR_ARG;
    const_cast<lvk_hip_stab*>(st)->finish_post();
    o->tracking_stability = st->tracking_stability; o->scene_quality = st->scene_quality; o->trust = st->trust;
    o->distribution = st->last_distribution; o->n_detected = st->last_detected; o->n_matched = st->last_matched;
    o->n_tracked = (int)st->tracked.size(); o->frame_delay = st->s.predictive_samples;
    o->smoothing_factor = st->smoother.smoothing_factor();
    for (int i = 0; i < 9; i++) o->homography[i] = st->last_H[i];
    return LVK_HIP_OK;
}

// How many frames ran the detector so far, and where their corners went through the suppression grid: inside the chain on the device
// (k_fast_insert) or in the host loop between two halves of it (grids / regions the kernel does not cover, LVK_HIP_HOST_GRID=1).
int lvk_hip_stab_detector_frames(const lvk_hip_stab* st, long long* on_device, long long* on_host)
{
    if (!st || !on_device || !on_host) return LVK_HIP_ERR_ARG;
    *on_device = st->device_grid_frames; *on_host = st->host_grid_frames;
    return LVK_HIP_OK;
}

int lvk_hip_stab_get_meshes(const lvk_hip_stab* st, float* motion, float* correction, int cap_floats)
{
    if (!st || !motion || !correction) return LVK_HIP_ERR_ARG;
    const int n = (int)st->last_motion.off.size();
    if (n > cap_floats) return LVK_HIP_ERR_ARG;
    std::memcpy(motion, st->last_motion.off.data(), n * sizeof(float));
    if ((int)st->last_correction.off.size() == n) std::memcpy(correction, st->last_correction.off.data(), n * sizeof(float));
    return n;
}

int lvk_hip_stab_get_features(const lvk_hip_stab* st, float* xy_resp_age, int cap)
{
    if (!st || !xy_resp_age) return LVK_HIP_ERR_ARG;
    const_cast<lvk_hip_stab*>(st)->finish_post();
    const int n = std::min(cap, (int)st->tracked.size());
    for (int i = 0; i < n; i++)
    {
        xy_resp_age[4 * i] = st->tracked[i].x; xy_resp_age[4 * i + 1] = st->tracked[i].y;
        xy_resp_age[4 * i + 2] = st->tracked[i].response; xy_resp_age[4 * i + 3] = (float)st->tracked[i].age;
    }
    return (int)st->tracked.size();
}

// StabilizationFilter::stable_region (StabilizationFilter.cpp:199-205): margins * frame size -> cv::Rect (rounded)
int lvk_hip_stab_stable_region(const lvk_hip_stab* st, int rows, int cols, int rect[4])
{
    if (!st || !rect) return LVK_HIP_ERR_ARG;
    float m[4]; st->smoother.margins(m);
    rect[0] = lvkh::cv_round(m[0] * (float)cols); rect[1] = lvkh::cv_round(m[1] * (float)rows);
    rect[2] = lvkh::cv_round(m[2] * (float)cols); rect[3] = lvkh::cv_round(m[3] * (float)rows);
    return LVK_HIP_OK;
}

} // extern "C"
