// Allocation and control of one stabilization filter: buffers, configure / restart / reset, overlap mode, profiling, debug overlays, taps.
// Reference: StabilizationFilter::{configure,restart,ready,reset_context,frame_delay,stable_region,draw_trackers,draw_motion_mesh}
// (Filters/StabilizationFilter.cpp:42-65,139-206), FrameTracker::{configure,restart} (Vision/FrameTracker.cpp:57-104).
#include "stab_state.hpp"

using namespace lvkstab;

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
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_fast_out, (size_t)fast_regions * fast_cap * sizeof(uint32_t), hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_fast_counts, fast_regions * sizeof(int), hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_regions, fast_regions * sizeof(FastRegion), hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_pts, n * sizeof(float2), hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_matched, n * sizeof(float2), hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_p1, 2 * n * sizeof(float2), hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_status, n, hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_H, 9 * sizeof(double), hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_ninl, sizeof(int), hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_mask, n, hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_und, 2 * n * sizeof(float2), hipHostMallocCoherent));
    LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_count, sizeof(int), hipHostMallocCoherent));
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
        LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_occ, ((grid.capacity() + 31) / 32 + 1) * sizeof(uint32_t), hipHostMallocCoherent));
        LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_new_kp, std::max<size_t>(grid.capacity(), 1) * sizeof(uint32_t), hipHostMallocCoherent));
        LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_insert, 8 * sizeof(int), hipHostMallocCoherent));
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
        { std::lock_guard<std::mutex> alock(st->ctx->aux_mutex); aux.erase(std::remove(aux.begin(), aux.end(), st->remap_stream), aux.end()); }
        if (st->remap_stream_owned) (void)hipStreamDestroy(st->remap_stream);
    }
    for (int i = 0; i < 2; i++) if (st->remap_done[i]) (void)hipEventDestroy(st->remap_done[i]);
    if (st->ingest_done) (void)hipEventDestroy(st->ingest_done);
    if (st->chain_done) (void)hipEventDestroy(st->chain_done);
    if (st->caller_ready) (void)hipEventDestroy(st->caller_ready);
    if (st->ahead_read_done) (void)hipEventDestroy(st->ahead_read_done);
    if (st->h_chain_flag) (void)hipHostFree(st->h_chain_flag);
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
    { std::lock_guard<std::mutex> alock(ctx->aux_mutex); aux.erase(std::remove(aux.begin(), aux.end(), st->remap_stream), aux.end()); }
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
            { std::lock_guard<std::mutex> alock(ctx->aux_mutex); ctx->aux_streams.push_back(st->remap_stream); }
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
    lvk_device_guard device_guard(st->ctx);
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
    lvk_device_guard device_guard(st->ctx);
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
    { int hrc; if ((hrc = st->flush_download(true)) != LVK_HIP_OK || (hrc = st->cancel_lookahead()) != LVK_HIP_OK || (hrc = st->finish_device_lookahead_reads()) != LVK_HIP_OK) return hrc; }
    st->scene_quality = 1.0f;
    st->queue.clear(); st->queue_kind = 0;
    st->pending_release = nullptr; st->pending_slot = -1;
    st->orphaned.clear();                          // restart(): every borrowed frame is the caller's again
    { const int src = st->sweep_retired(); if (src != LVK_HIP_OK) return src; }      // pool slots of an earlier frame size whose frames were still queued
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

// ---- taps (tests / HUD)
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

// Which schedule the pushes took (LVK_SCHED_*): see include/lvk_hip.h
int lvk_hip_stab_schedule_counters(lvk_hip_stab* st, long long out[LVK_SCHED_COUNT], int reset)
{
    if (!st || !out) return LVK_HIP_ERR_ARG;
    for (int i = 0; i < LVK_SCHED_COUNT; i++) { out[i] = st->sched[i]; if (reset) st->sched[i] = 0; }
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
