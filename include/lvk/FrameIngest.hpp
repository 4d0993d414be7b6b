// lvk::FrameIngest -- the OBS plugin's Interop/FrameIngest (reference: Modules/OBS-Plugin/Interop/FrameIngest.hpp:28-228, FrameIngest.cpp) over the
// C-ABI's lvk_hip_ingest_obs / lvk_hip_egress_obs: the step either side of every filter in the plugin's asynchronous path,
//
//     auto ingest = lvk::FrameIngest::Select(frame->format);        // FrameIngest.cpp:36-75
//     ingest->upload_obs_frame(frame, video_frame);                  // :92-102   obs_source_frame -> VideoFrame (packed 8UC3 on the device)
//     filter.apply(std::move(video_frame), video_frame);
//     ingest->download_ocl_frame(video_frame, frame);                // :106-117  and back
//
// The reference has one subclass per family of formats (I4XXIngest, NV12Ingest, P422Ingest, P444Ingest, DirectIngest); here the family is a switch
// inside the library and this class only moves the planes.  libobs is not a dependency of this header: the member templates take ANY struct with
// obs_source_frame's members -- data[], linesize[], width, height, timestamp, format -- so obs_source_frame itself fits, and so does a test's stand-in.
//
// Transfers: like the reference (one bulk cv::Mat::copyTo(UMat) of the planes' contiguous span, blocking, FrameIngest.cpp:362-474) the upload has
// finished with the host planes when it returns, and the download has filled them.  Hosts that own pinned planes and want the transfers scheduled
// around the filter use lvk_hip_stab_push_yuv420_host instead (HostFrame420).
#pragma once

#include "LiveVisionKit.hpp"

#include <memory>

namespace lvk {

class FrameIngest
{
public:
    // nullptr for a format the plugin does not convert (FrameIngest.cpp:71-75 returns an empty pointer)
    static std::unique_ptr<FrameIngest> Select(int obs_format, const std::shared_ptr<hip::Context>& ctx = nullptr)
    {
        const int f = lvk_hip_obs_frame_format(obs_format);
        if (f < 0) return nullptr;
        if (f == LVK_FORMAT_GRAY) return nullptr;             // Y800: a one-channel VideoFrame, which this facade's VideoFrame (8UC3) does not model; the C-ABI converts it
        return std::unique_ptr<FrameIngest>(new FrameIngest(obs_format, static_cast<VideoFrame::Format>(f), ctx));
    }

    int obs_format() const { return m_OBSFormat; }                         // :128-131
    VideoFrame::Format ocl_format() const { return m_OCLFormat; }          // :121-124

    template <class ObsFrame>
    static bool test_obs_frame(const ObsFrame* frame)                       // :135-142
    {
        return frame != nullptr && frame->data[0] != nullptr && frame->width > 0 && frame->height > 0 && frame->format != 0;
    }

    template <class ObsFrame>
    void upload_obs_frame(const ObsFrame* src, VideoFrame& dst)
    {
        LVK_HIP_ASSERT(test_obs_frame(src) && (int)src->format == m_OBSFormat);
        const int rows = (int)src->height, cols = (int)src->width;
        Plane pl[3]; const int n = planes(rows, cols, pl);
        hip::ContextLock lock(m_ctx->mutex());
        size_t off[3], total = 0;
        for (int i = 0; i < n; i++)
        {
            LVK_HIP_ASSERT(src->data[i] != nullptr);
            pl[i].step = src->linesize[i] != 0 ? (int)src->linesize[i] : pl[i].width_bytes;
            LVK_HIP_ASSERT(pl[i].step >= pl[i].width_bytes);
            off[i] = total; total += ((size_t)pl[i].step * pl[i].rows + 255) & ~(size_t)255;
        }
        stage(total);
        const void* d_planes[3] = {nullptr, nullptr, nullptr}; int steps[3] = {0, 0, 0};
        for (int i = 0; i < n; i++)
        {
            uint8_t* d = static_cast<uint8_t*>(m_stage.get()) + off[i];
            m_ctx->check(lvk_hip_upload(m_ctx->get(), d, src->data[i], (size_t)pl[i].step * pl[i].rows), "FrameIngest::upload_obs_frame");
            d_planes[i] = d; steps[i] = pl[i].step;
        }
        dst.create({cols, rows}, CV_8UC3, m_ctx);
        m_ctx->check(lvk_hip_ingest_obs(m_ctx->get(), m_OBSFormat, d_planes, steps, rows, cols, dst.device_ptr(), (int)dst.step), "FrameIngest::upload_obs_frame");
        m_ctx->check(lvk_hip_sync(m_ctx->get()), "FrameIngest::upload_obs_frame");       // the host planes are the caller's again
        dst.timestamp = src->timestamp;                                                  // :99-101
        dst.format = m_OCLFormat;
    }

    template <class ObsFrame>
    void download_ocl_frame(const VideoFrame& src, ObsFrame* dst)
    {
        LVK_HIP_ASSERT(test_obs_frame(dst) && (int)dst->format == m_OBSFormat);
        LVK_HIP_ASSERT(src.has_known_format() && !src.empty());
        // (the reference converts a frame of another known format first, viewAsFormat, :113; the filters of this facade keep the format they were given)
        LVK_HIP_ASSERT(src.format == m_OCLFormat);
        const int rows = (int)dst->height, cols = (int)dst->width;
        LVK_HIP_ASSERT(rows == src.rows && cols == src.cols);
        Plane pl[3]; const int n = planes(rows, cols, pl);
        hip::ContextLock lock(m_ctx->mutex());
        if (src.context() && src.context() != m_ctx) m_ctx->wait_for(*src.context());
        size_t off[3], total = 0;
        for (int i = 0; i < n; i++)
        {
            LVK_HIP_ASSERT(dst->data[i] != nullptr);
            pl[i].step = pl[i].width_bytes;                                              // tight on the device; the host pitch is honoured row by row below
            off[i] = total; total += ((size_t)pl[i].step * pl[i].rows + 255) & ~(size_t)255;
        }
        stage(total);
        void* d_planes[3] = {nullptr, nullptr, nullptr}; int steps[3] = {0, 0, 0};
        for (int i = 0; i < n; i++) { d_planes[i] = static_cast<uint8_t*>(m_stage.get()) + off[i]; steps[i] = pl[i].step; }
        m_ctx->check(lvk_hip_egress_obs(m_ctx->get(), m_OBSFormat, src.device_ptr(), (int)src.step, rows, cols, d_planes, steps), "FrameIngest::download_ocl_frame");
        for (int i = 0; i < n; i++)
        {
            const int host_step = dst->linesize[i] != 0 ? (int)dst->linesize[i] : pl[i].width_bytes;
            LVK_HIP_ASSERT(host_step >= pl[i].written_bytes);
            const uint8_t* d = static_cast<const uint8_t*>(d_planes[i]);
            if (host_step == pl[i].width_bytes && pl[i].written_bytes == pl[i].width_bytes)
                m_ctx->check(lvk_hip_download(m_ctx->get(), dst->data[i], d, (size_t)pl[i].step * pl[i].rows), "FrameIngest::download_ocl_frame");
            else if (pl[i].linear)                                                       // DirectIngest's 4-byte formats: rows * cols * 3 bytes of the stream (:751-753)
                m_ctx->check(lvk_hip_download(m_ctx->get(), dst->data[i], d, (size_t)pl[i].written_bytes * pl[i].rows), "FrameIngest::download_ocl_frame");
            else
                for (int r = 0; r < pl[i].rows; r++)
                    m_ctx->check(lvk_hip_download(m_ctx->get(), dst->data[i] + (size_t)r * host_step, d + (size_t)r * pl[i].step, (size_t)pl[i].written_bytes),
                                 "FrameIngest::download_ocl_frame");
        }
        m_ctx->check(lvk_hip_sync(m_ctx->get()), "FrameIngest::download_ocl_frame");
        dst->timestamp = src.timestamp;                                                  // :116
    }

    const std::shared_ptr<hip::Context>& context() const { return m_ctx; }

private:
    struct Plane { int rows = 0, width_bytes = 0, written_bytes = 0, step = 0; bool linear = false; };

    FrameIngest(int obs_format, VideoFrame::Format ocl_format, const std::shared_ptr<hip::Context>& ctx)
        : m_OBSFormat(obs_format), m_OCLFormat(ocl_format), m_ctx(ctx ? ctx : hip::shared_context()) {}

    // the planes FrameIngest moves for one frame (the alpha planes of I40A / I42A / YUVA stay where they are)
    int planes(int rows, int cols, Plane pl[3]) const
    {
        auto set = [&](int i, int r, int w) { pl[i].rows = r; pl[i].width_bytes = w; pl[i].written_bytes = w; };
        switch (m_OBSFormat)
        {
        case LVK_VIDEO_FORMAT_I420: case LVK_VIDEO_FORMAT_I40A: set(0, rows, cols); set(1, rows / 2, cols / 2); set(2, rows / 2, cols / 2); return 3;
        case LVK_VIDEO_FORMAT_NV12: set(0, rows, cols); set(1, rows / 2, cols); return 2;
        case LVK_VIDEO_FORMAT_I422: case LVK_VIDEO_FORMAT_I42A: set(0, rows, cols); set(1, rows, cols / 2); set(2, rows, cols / 2); return 3;
        case LVK_VIDEO_FORMAT_I444: case LVK_VIDEO_FORMAT_YUVA: set(0, rows, cols); set(1, rows, cols); set(2, rows, cols); return 3;
        case LVK_VIDEO_FORMAT_YUY2: case LVK_VIDEO_FORMAT_YVYU: case LVK_VIDEO_FORMAT_UYVY: set(0, rows, 2 * cols); return 1;
        case LVK_VIDEO_FORMAT_AYUV: set(0, rows, 4 * cols); return 1;
        case LVK_VIDEO_FORMAT_BGR3: set(0, rows, 3 * cols); return 1;
        case LVK_VIDEO_FORMAT_RGBA: case LVK_VIDEO_FORMAT_BGRA: case LVK_VIDEO_FORMAT_BGRX:
            set(0, rows, 4 * cols); pl[0].written_bytes = 3 * cols; pl[0].linear = true; return 1;
        }
        LVK_HIP_ASSERT(false && "format not supported");
        return 0;
    }

    void stage(size_t bytes)
    {
        if (bytes <= m_stage_bytes && m_stage) return;
        void* p = nullptr;
        m_ctx->check(lvk_hip_malloc(m_ctx->get(), bytes, &p), "FrameIngest::stage");
        auto c = m_ctx;
        m_stage = std::shared_ptr<void>(p, [c](void* q) { lvk_hip_free(c->get(), q); });       // m_ImportBuffer / m_ExportBuffer of the reference
        m_stage_bytes = bytes;
    }

    int m_OBSFormat;
    VideoFrame::Format m_OCLFormat;
    std::shared_ptr<hip::Context> m_ctx;
    std::shared_ptr<void> m_stage;
    size_t m_stage_bytes = 0;
};

} // namespace lvk
