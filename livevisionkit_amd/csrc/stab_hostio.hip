// Frames in HOST memory either side of the 4:2:0 path: staging planes, transfer streams, deferred downloads, upload look-ahead.
// Reference: FrameIngest::upload_planes / download_planes around to_ocl -> filter -> to_obs (Modules/OBS-Plugin/Interop/FrameIngest.cpp:415-474,
// 494-602, VisionFilter.cpp:151-212).
#include "stab_state.hpp"

using namespace lvkstab;

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
    for (hipStream_t s : {h.up, h.down, h.down2})
        if (s)
        {
            (void)hipStreamSynchronize(s);
            { std::lock_guard<std::mutex> alock(ctx->aux_mutex); aux.erase(std::remove(aux.begin(), aux.end(), s), aux.end()); }
            (void)hipStreamDestroy(s);
        }
    h.up = h.down = h.down2 = nullptr;
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
        // streams would avoid, but a second stream costs more than that wait: see ensure_hostio; 3 140-3 190 against 2 700-2 850 frames/s.)
        hipStream_t us = io.up;
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_y, h_y, bytes, hipMemcpyHostToDevice, us));
        LVK_HIP_CHECK(ctx, hipEventRecord(io.c_done[k], us));
        io.y_is_c[k] = true;
        return LVK_HIP_OK;
    }
    io.y_is_c[k] = false;
    hipStream_t cs = io.up;                                                          // luma and chroma of a frame pushed now: one stream, in order
    LVK_HIP_CHECK(ctx, copy_plane(d_y, cols, h_y, y_step, cols, rows, io.up));
    LVK_HIP_CHECK(ctx, hipEventRecord(io.y_done[k], io.up));
    if (!nv12 && u_step == ccols && v_step == ccols && (const uint8_t*)h_v == (const uint8_t*)h_u + (size_t)crows * ccols)
    {
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_u, h_u, 2 * (size_t)crows * ccols, hipMemcpyHostToDevice, cs));      // U | V contiguous: one copy
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
    if (io.up) LVK_HIP_CHECK(ctx, hipStreamSynchronize(io.up));
    io.ahead.clear();
    for (hipEvent_t& e : ingest_wait) e = nullptr;
    return LVK_HIP_OK;
}

extern "C" {

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
                                  void* oh_y, int oy_step, void* oh_u, int ou_step, void* oh_v, int ov_step, int o_rows,
                                  int* produced, uint64_t* out_timestamp, lvk_frame_info* emitted)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    lvk_hip_ctx* ctx = st->ctx;
    if (produced) *produced = 0;
    LVK_HIP_REQUIRE(ctx, h_y && h_u && (nv12 || h_v) && rows > 0 && cols > 0 && rows % 2 == 0 && cols % 2 == 0);
    LVK_HIP_REQUIRE(ctx, y_step >= cols && u_step >= (nv12 ? cols : cols / 2) && (nv12 || v_step >= cols / 2));
    int rc;
    if ((rc = st->require_pinned_planes(h_y, y_step, h_u, u_step, h_v, v_step, nv12, rows, cols, "lvk_hip_stab_push_yuv420_host")) != LVK_HIP_OK) return rc;
    // the frame this push emits is the DELAYED one, at its own size (frames queued before a resize leave at the old size): the output planes are
    // checked -- pitch, rows, pinned over their whole extent -- against THAT geometry
    QueuedFrame due{};
    const bool will_emit = st->next_output(QueuedFrame{nullptr, 3 * cols, rows, cols, timestamp, LVK_FORMAT_YUV}, &due);
    const int erows = will_emit ? due.rows : rows, ecols = will_emit ? due.cols : cols;
    // (planes given to a push that emits nothing are held to the incoming frame's geometry: pageable memory is refused whenever it is seen)
    if (oh_y && oh_u && (nv12 || oh_v))
    {
        if (!(oy_step >= ecols && ou_step >= (nv12 ? ecols : ecols / 2) && (nv12 || ov_step >= ecols / 2) && o_rows >= erows))
            return st->fail(LVK_HIP_ERR_ARG, "the output planes do not hold the frame this push emits: " + std::to_string(ecols) + " x " + std::to_string(erows) +
                                             " (the DELAYED frame's own size -- lvk_hip_stab_next_output); nothing was queued");
        if ((rc = st->require_pinned_planes(oh_y, oy_step, oh_u, ou_step, oh_v, ov_step, nv12, erows, ecols, "lvk_hip_stab_push_yuv420_host (output)")) != LVK_HIP_OK) return rc;
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
    // (a frame of an EARLIER size -- the staging planes of the download route have the new one -- always leaves through the kernel's own stores)
    const bool direct = have_out && (st->host_sink_mode != 2 || erows != rows || ecols != cols);
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
    rc = lvk_hip_stab_push_yuv420(st, d_y, cols, d_u, ccols, d_v, ccols, nv12, rows, cols, timestamp, o_y, oys, o_u, ous, o_v, ovs, direct ? o_rows : rows, &prod, out_timestamp, emitted);
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
    lvk_device_guard device_guard(ctx);
    LVK_HIP_CHECK(ctx, hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    return LVK_HIP_OK;
}

int lvk_hip_host_free(lvk_hip_ctx* ctx, void* h_ptr)
{
    if (!ctx) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(ctx);
    if (h_ptr) LVK_HIP_CHECK(ctx, hipHostFree(h_ptr));
    return LVK_HIP_OK;
}

} // extern "C"
