// Device-resident frames announced one push ahead: the announced luma's downscale + pyramid go behind the current push's chain.
// Reference: the reader thread of VideoFilter::stream runs ahead of the filter thread (Filters/VideoFilter.cpp:62-209).
#include "stab_state.hpp"

using namespace lvkstab;

// The announced frame's downscale + pyramid, put on the tracking stream behind the chain of the push that is under way (called by track()
// after its last launch).  `P` (the previous frame's pyramid) has been read for the last time by the flow kernel of this push; at the next
// push it is `C`.  The event behind the two launches is what lvk_hip_stab_prefetch_cancel / _restart wait for before they hand the announced
// luma plane back to the caller.
int lvk_hip_stab::launch_build_ahead(DevicePyramid& P, int cur_w, int cur_h)
{
    const LumaAhead a = ahead_announced;
    ahead_announced = LumaAhead();
    int rc;
    int pe = prof_begin(LVK_STAGE_DOWNSCALE);
    if ((rc = lvk_launch_luma_area_resize(ctx, a.luma, a.step, a.pix, a.channel, a.rows, a.cols, const_cast<uint8_t*>(P.args.lv[0].img), P.args.lv[0].step, cur_h, cur_w)) != LVK_HIP_OK) return rc;
    prof_end(pe);
    pe = prof_begin(LVK_STAGE_PYRAMID);
    if ((rc = P.build(ctx)) != LVK_HIP_OK) return rc;
    prof_end(pe);
    if (!ahead_read_done) LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&ahead_read_done, hipEventDisableTiming));
    LVK_HIP_CHECK(ctx, hipEventRecord(ahead_read_done, ctx->stream));
    ahead_read_armed = true;
    ahead_built = a; ahead_built_for = push_seq + 1;
    return LVK_HIP_OK;
}

extern "C" {

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

// Forget the announced frames that have not been pushed (lvk_hip_stab_restart does the same): for a caller that announced frame n + 1 and
// then stops, seeks or switches to other buffers.  Returns once nothing reads the announced planes any more: the uploads of host planes have
// completed, and a pyramid that was being built ahead from a device-resident luma plane has read it.
int lvk_hip_stab_prefetch_cancel(lvk_hip_stab* st)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    st->forget_device_lookahead();
    { const int rc = st->finish_device_lookahead_reads(); if (rc != LVK_HIP_OK) return rc; }
    return st->cancel_lookahead();
}

} // extern "C"
