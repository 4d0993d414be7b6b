// C++ facade of the MI355X-native stabilization path: the lvk::VideoFilter / lvk::StabilizationFilter API of
// LiveVisionKit (reference: LiveVisionKit/Filters/VideoFilter.hpp:32-61, Filters/StabilizationFilter.hpp:28-78,
// Utility/Configurable.hpp:26-44, Utility/Unique.hpp, Timing/Stopwatch.hpp, Data/VideoFrame.hpp:25-81) as a
// header-only wrapper over the C-ABI of lvk_hip.h.  Same class names, member names, settings fields and defaults,
// so the call sites of Modules/OBS-Plugin/Sources/Stabilisation/VSFilter.cpp compile against it unchanged
// (tests/cpp/plugin_conformance.cpp reproduces them).
//
// Differences that a HIP build implies:
//   * lvk::VideoFrame owns a HIP device buffer (the reference's VideoFrame is a cv::UMat, i.e. an OpenCL buffer).
//   * errors go through lvk::context::assert_handler exactly as LVK_ASSERT does (Directives.hpp:37-52).
#pragma once

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <iostream>
#include <mutex>
#include <thread>
#include <initializer_list>
#include <memory>
#include <numeric>
#include <string>
#include <utility>
#include <vector>

#include "../lvk_hip.h"
#include "cv_min.hpp"

namespace lvk {

// ---------------------------------------------------------------------------------------------- Directives.hpp
namespace context {
inline std::function<void(std::string, std::string, std::string)> assert_handler =
    [](std::string file, std::string function, std::string assertion) {
        std::cerr << "Failed Assert " << file << "@" << function << "(..) ... " << assertion << std::endl;
        std::abort();
    };
}
#define LVK_HIP_ASSERT(cond) do { if (!(cond)) lvk::context::assert_handler("LiveVisionKit.hpp", __func__, #cond); } while (0)

// ---------------------------------------------------------------------------------------------- hip context
namespace hip {
// One HIP context (stream + staging) per object that needs one; replaces OpenCV's implicit OpenCL queue.
class Context
{
public:
    explicit Context(int device = 0)
    {
        if (lvk_hip_ctx_create(device, &m_ctx) != LVK_HIP_OK)
            lvk::context::assert_handler("LiveVisionKit.hpp", "Context", std::string("lvk_hip_ctx_create: ") + lvk_hip_last_error(nullptr));
    }
    // a context that enqueues on an existing hipStream_t (e.g. the stream a filter produces its outputs on)
    Context(int device, void* hip_stream)
    {
        if (lvk_hip_ctx_create_on_stream(device, hip_stream, &m_ctx) != LVK_HIP_OK)
            lvk::context::assert_handler("LiveVisionKit.hpp", "Context", std::string("lvk_hip_ctx_create_on_stream: ") + lvk_hip_last_error(nullptr));
    }
    ~Context() { lvk_hip_ctx_destroy(m_ctx); }
    // The C-ABI's rule is "one context is driven by one host thread at a time" (lvk_hip.h).  Objects of this facade may SHARE a context
    // across threads (two filters constructed on one context, a frame handed to another thread): every facade call that drives a
    // context holds its mutex for the duration of the call.  Recursive: facade calls nest (a filter's apply creates frames).
    std::recursive_mutex& mutex() const { return m_mutex; }
    // everything enqueued so far on `producer` happens before what this context enqueues from now on (GPU-side, no host wait)
    void wait_for(const Context& producer) const
    {
        if (producer.m_ctx == m_ctx) return;
        std::scoped_lock lock(m_mutex, producer.m_mutex);         // both, deadlock-free whatever the order two threads name them in
        check(lvk_hip_ctx_wait(m_ctx, producer.m_ctx), "Context::wait_for");
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    lvk_hip_ctx* get() const { return m_ctx; }
    void check(int rc, const char* what) const
    {
        if (rc < 0) lvk::context::assert_handler("LiveVisionKit.hpp", what, lvk_hip_last_error(m_ctx));
    }
private:
    lvk_hip_ctx* m_ctx = nullptr;
    mutable std::recursive_mutex m_mutex;
};
using ContextLock = std::lock_guard<std::recursive_mutex>;
inline std::shared_ptr<Context> shared_context(int device = 0)
{
    static thread_local std::weak_ptr<Context> cached;
    auto c = cached.lock();
    if (!c) { c = std::make_shared<Context>(device); cached = c; }
    return c;
}
} // namespace hip

// ---------------------------------------------------------------------------------------------- Data/VideoFrame.hpp
struct VideoFrame
{
    enum Format { BGR, BGRA, RGB, RGBA, YUV, GRAY, UNKNOWN };

    uint64_t timestamp = 0;
    Format format = UNKNOWN;
    int cols = 0, rows = 0;
    int& width = cols; int& height = rows;
    size_t step = 0;

    VideoFrame() = default;
    explicit VideoFrame(uint64_t ts) : timestamp(ts) {}
    VideoFrame(const VideoFrame& o) : timestamp(o.timestamp), format(o.format), cols(o.cols), rows(o.rows), step(o.step), m_buf(o.m_buf), m_ctx(o.m_ctx) {}
    VideoFrame(VideoFrame&& o) noexcept : timestamp(o.timestamp), format(o.format), cols(o.cols), rows(o.rows), step(o.step), m_buf(std::move(o.m_buf)), m_ctx(std::move(o.m_ctx)) { o.cols = o.rows = 0; o.step = 0; }
    VideoFrame& operator=(const VideoFrame& o) { timestamp = o.timestamp; format = o.format; cols = o.cols; rows = o.rows; step = o.step; m_buf = o.m_buf; m_ctx = o.m_ctx; return *this; }
    VideoFrame& operator=(VideoFrame&& o) noexcept
    {
        timestamp = o.timestamp; format = o.format; cols = o.cols; rows = o.rows; step = o.step; m_buf = std::move(o.m_buf); m_ctx = std::move(o.m_ctx);
        o.cols = o.rows = 0; o.step = 0; return *this;
    }
    virtual ~VideoFrame() = default;

    // cv::UMat subset
    bool empty() const { return !m_buf || cols == 0 || rows == 0; }
    cv::Size size() const { return {cols, rows}; }
    int type() const { return CV_8UC3; }
    void release() { m_buf.reset(); cols = rows = 0; step = 0; }
    void create(const cv::Size& sz, int /*type = CV_8UC3*/, const std::shared_ptr<hip::Context>& ctx = nullptr)
    {
        if (m_buf && m_buf.use_count() == 1 && cols == sz.width && rows == sz.height && (!ctx || ctx == m_ctx)) return;
        // lvk_hip_malloc / lvk_hip_free pool by size (like cv::UMat's OpenCL buffer pool): no hipMalloc / hipFree in steady state.
        // A frame's buffer is written on its context's stream; a filter that reads it on another stream fences it back before dropping it.
        m_ctx = ctx ? ctx : (m_ctx ? m_ctx : hip::shared_context());
        void* p = nullptr;
        step = (size_t)sz.width * 3;
        m_ctx->check(lvk_hip_malloc(m_ctx->get(), step * (size_t)sz.height, &p), "VideoFrame::create");
        auto c = m_ctx;
        m_buf = std::shared_ptr<void>(p, [c](void* q) { lvk_hip_free(c->get(), q); });
        cols = sz.width; rows = sz.height;
    }
    void* device_ptr() const { return m_buf.get(); }
    bool has_known_format() const { return format != UNKNOWN; }

    // host <-> device helpers (reference: cv::UMat::copyTo / getMat); packed 8UC3, tight rows
    void upload(const uint8_t* host, int rows_, int cols_, Format fmt, uint64_t ts, const std::shared_ptr<hip::Context>& ctx = nullptr)
    {
        create({cols_, rows_}, CV_8UC3, ctx);
        hip::ContextLock lock(m_ctx->mutex());
        m_ctx->check(lvk_hip_upload(m_ctx->get(), m_buf.get(), host, step * (size_t)rows), "VideoFrame::upload");
        format = fmt; timestamp = ts;
    }
    void download(uint8_t* host) const
    {
        hip::ContextLock lock(m_ctx->mutex());
        m_ctx->check(lvk_hip_download(m_ctx->get(), host, m_buf.get(), step * (size_t)rows), "VideoFrame::download");
        m_ctx->check(lvk_hip_sync(m_ctx->get()), "VideoFrame::download");
    }
    const std::shared_ptr<hip::Context>& context() const { return m_ctx; }
    std::shared_ptr<void> buffer() const { return m_buf; }
    VideoFrame clone() const                               // cv::UMat::clone: a device copy with the same metadata
    {
        VideoFrame c;
        if (empty()) return c;
        c.create(size(), CV_8UC3, m_ctx);
        hip::ContextLock lock(m_ctx->mutex());
        m_ctx->check(lvk_hip_upscale(m_ctx->get(), m_buf.get(), (int)step, rows, cols, c.m_buf.get(), (int)c.step, rows, cols, 1), "VideoFrame::clone");
        c.timestamp = timestamp; c.format = format;
        return c;
    }

private:
    std::shared_ptr<void> m_buf;
    std::shared_ptr<hip::Context> m_ctx;
};
typedef VideoFrame Frame;

// The plugin's wire format on the device: I420 (y, u, v planes) or NV12 (y + interleaved uv), what I4XXIngest / NV12Ingest move between
// OBS and the filter chain (Modules/OBS-Plugin/Interop/FrameIngest.cpp:494-602).  StabilizationFilter::apply takes it directly: the
// 4:2:0 -> 4:4:4 conversion, the filter and the 4:4:4 -> 4:2:0 conversion run as one push (lvk_hip_stab_push_yuv420).
struct VideoFrame420
{
    uint64_t timestamp = 0;
    bool nv12 = false;
    int cols = 0, rows = 0;

    bool empty() const { return !m_buf || cols == 0 || rows == 0; }
    void release() { m_buf.reset(); cols = rows = 0; }
    void create(const cv::Size& sz, bool nv12_, const std::shared_ptr<hip::Context>& ctx = nullptr)
    {
        LVK_HIP_ASSERT(sz.width > 0 && sz.height > 0 && sz.width % 2 == 0 && sz.height % 2 == 0);
        if (m_buf && m_buf.use_count() == 1 && cols == sz.width && rows == sz.height && nv12 == nv12_ && (!ctx || ctx == m_ctx)) return;
        m_ctx = ctx ? ctx : (m_ctx ? m_ctx : hip::shared_context());
        void* p = nullptr;
        m_ctx->check(lvk_hip_malloc(m_ctx->get(), (size_t)sz.width * sz.height * 3 / 2, &p), "VideoFrame420::create");
        auto c = m_ctx;
        m_buf = std::shared_ptr<void>(p, [c](void* q) { lvk_hip_free(c->get(), q); });
        cols = sz.width; rows = sz.height; nv12 = nv12_;
    }
    // one allocation: Y, then U and V (I420) or the interleaved UV plane (NV12)
    uint8_t* y() const { return static_cast<uint8_t*>(m_buf.get()); }
    uint8_t* u() const { return y() + (size_t)cols * rows; }
    uint8_t* v() const { return nv12 ? u() : u() + (size_t)(cols / 2) * (rows / 2); }
    int y_step() const { return cols; }
    int uv_step() const { return nv12 ? cols : cols / 2; }
    void upload(const uint8_t* host, int rows_, int cols_, bool nv12_, uint64_t ts, const std::shared_ptr<hip::Context>& ctx = nullptr)
    {
        create({cols_, rows_}, nv12_, ctx);
        hip::ContextLock lock(m_ctx->mutex());
        m_ctx->check(lvk_hip_upload(m_ctx->get(), m_buf.get(), host, (size_t)cols * rows * 3 / 2), "VideoFrame420::upload");
        timestamp = ts;
    }
    void download(uint8_t* host) const
    {
        hip::ContextLock lock(m_ctx->mutex());
        m_ctx->check(lvk_hip_download(m_ctx->get(), host, m_buf.get(), (size_t)cols * rows * 3 / 2), "VideoFrame420::download");
        m_ctx->check(lvk_hip_sync(m_ctx->get()), "VideoFrame420::download");
    }
    const std::shared_ptr<hip::Context>& context() const { return m_ctx; }
private:
    std::shared_ptr<void> m_buf;
    std::shared_ptr<hip::Context> m_ctx;
};

// The same wire format in PINNED HOST memory: what OBS hands the plugin (obs_source_frame planes) before FrameIngest::upload_planes and
// after download_planes (Modules/OBS-Plugin/Interop/FrameIngest.cpp:415-474).  StabilizationFilter::apply takes it directly
// (lvk_hip_stab_push_yuv420_host): the library schedules the uploads, the conversion, the filter and the way back.
struct HostFrame420
{
    uint64_t timestamp = 0;
    bool nv12 = false;
    int cols = 0, rows = 0;

    bool empty() const { return !m_buf || cols == 0 || rows == 0; }
    void release() { m_buf.reset(); cols = rows = 0; }
    void create(const cv::Size& sz, bool nv12_, const std::shared_ptr<hip::Context>& ctx = nullptr)
    {
        LVK_HIP_ASSERT(sz.width > 0 && sz.height > 0 && sz.width % 2 == 0 && sz.height % 2 == 0);
        if (m_buf && m_buf.use_count() == 1 && cols == sz.width && rows == sz.height && (!ctx || ctx == m_ctx)) { nv12 = nv12_; return; }
        m_ctx = ctx ? ctx : (m_ctx ? m_ctx : hip::shared_context());
        void* p = nullptr;
        m_ctx->check(lvk_hip_host_malloc(m_ctx->get(), (size_t)sz.width * sz.height * 3 / 2, &p), "HostFrame420::create");
        auto c = m_ctx;
        m_buf = std::shared_ptr<void>(p, [c](void* q) { lvk_hip_host_free(c->get(), q); });
        cols = sz.width; rows = sz.height; nv12 = nv12_;
    }
    uint8_t* y() const { return static_cast<uint8_t*>(m_buf.get()); }
    uint8_t* u() const { return y() + (size_t)cols * rows; }
    uint8_t* v() const { return nv12 ? u() : u() + (size_t)(cols / 2) * (rows / 2); }
    int y_step() const { return cols; }
    int uv_step() const { return nv12 ? cols : cols / 2; }
    size_t bytes() const { return (size_t)cols * rows * 3 / 2; }
    bool unique() const { return m_buf && m_buf.use_count() == 1; }
    // an output of StabilizationFilter::apply is complete once its filter's context is idle
    void wait() const { if (m_ctx) { hip::ContextLock lock(m_ctx->mutex()); m_ctx->check(lvk_hip_sync(m_ctx->get()), "HostFrame420::wait"); } }
    const std::shared_ptr<hip::Context>& context() const { return m_ctx; }
private:
    std::shared_ptr<void> m_buf;
    std::shared_ptr<hip::Context> m_ctx;
};

// ---------------------------------------------------------------------------------------------- Timing
class Time
{
public:
    Time() = default;
    explicit Time(double seconds) : m_s(seconds) {}
    double seconds() const { return m_s; }
    double milliseconds() const { return m_s * 1e3; }
    double microseconds() const { return m_s * 1e6; }
private:
    double m_s = 0.0;
};

class Stopwatch
{
public:
    explicit Stopwatch(size_t history = 1) : m_cap(history ? history : 1) {}
    void start() { m_t0 = clock::now(); m_running = true; }
    Time stop()
    {
        const double s = std::chrono::duration<double>(clock::now() - m_t0).count();
        m_running = false;
        m_hist.push_back(s);
        while (m_hist.size() > m_cap) m_hist.pop_front();
        return Time(s);
    }
    Time average() const { return Time(m_hist.empty() ? 0.0 : std::accumulate(m_hist.begin(), m_hist.end(), 0.0) / (double)m_hist.size()); }
    Time deviation() const
    {
        if (m_hist.empty()) return Time(0.0);
        const double a = average().seconds();
        double v = 0; for (double s : m_hist) v += (s - a) * (s - a);
        return Time(std::sqrt(v / (double)m_hist.size()));
    }
    void set_history_size(size_t n) { m_cap = n ? n : 1; while (m_hist.size() > m_cap) m_hist.pop_front(); }
    bool is_running() const { return m_running; }
private:
    using clock = std::chrono::steady_clock;
    clock::time_point m_t0{};
    bool m_running = false;
    size_t m_cap;
    std::deque<double> m_hist;
};

// ---------------------------------------------------------------------------------------------- Utility
template <typename Scope> class Unique
{
public:
    Unique() : m_UID(next()++) {}
    Unique(const Unique&) : m_UID(next()++) {}
    Unique(Unique&& o) noexcept : m_UID(o.m_UID) {}
    virtual ~Unique() = default;
    uint64_t uid() const { return m_UID; }
private:
    static uint64_t& next() { static uint64_t n = 1; return n; }
    uint64_t m_UID;
};

template <typename T> class Configurable
{
public:
    explicit Configurable(const T& settings = {}) : m_Settings(settings) {}
    virtual ~Configurable() = default;
    void configure_default() { configure(T{}); }
    virtual void configure(const T& settings) = 0;
    void reconfigure(const std::function<void(T&)>& updater) { T s = m_Settings; updater(s); configure(s); }
    const T& settings() const { return m_Settings; }
protected:
    T m_Settings;
};

// ---------------------------------------------------------------------------------------------- settings (same names & defaults)
struct FeatureDetectorSettings                       // Vision/FeatureDetector.hpp:28-37
{
    cv::Size detection_resolution = {256, 256};
    cv::Size detection_regions = {2, 2};
    bool force_detection = false;
    float max_feature_density = 0.20f;
    float min_feature_density = 0.05f;
    float accumulation_rate = 2.0f;
};

struct FrameTrackerSettings : public FeatureDetectorSettings   // Vision/FrameTracker.hpp:31-44
{
    cv::Size motion_resolution = {16, 16};
    bool track_local_motions = true;
    float temporal_smoothing = 1.0f;
    float local_smoothing = 20.0f;
    size_t min_motion_samples = 75;
    float acceptance_threshold = 8.0f;
    float uniformity_threshold = 0.20f;
};

struct PathSmootherSettings                          // Vision/PathSmoother.hpp:29-39
{
    size_t predictive_samples = 10;
    cv::Size motion_resolution = {2, 2};
    cv::Size2f corrective_limits = {0.1f, 0.1f};
    float smoothing_steps = 20.0f;
    float response_rate = 0.04f;
};

struct StabilizationFilterSettings : public FrameTrackerSettings, public PathSmootherSettings   // Filters/StabilizationFilter.hpp:28-39
{
    cv::Size motion_resolution = {2, 2};
    cv::Scalar background_colour = {255, 0, 255};
    bool crop_to_stable_region = false;
    bool stabilize_output = true;
    float min_scene_quality = 0.8f;
    float min_tracking_quality = 0.3f;
};

} // namespace lvk

#ifndef LVK_WITH_OPENCV
// Stand-in for the one cv::VideoCapture use on the path (VideoFilter::stream, Filters/VideoFilter.cpp:62-209): a pull source
// of frames.  Subclass it (file reader, camera, synthetic generator).
namespace cv {
enum { CAP_PROP_POS_MSEC = 0 };
class VideoCapture
{
public:
    virtual ~VideoCapture() = default;
    virtual bool isOpened() const = 0;
    virtual bool read(lvk::VideoFrame& frame) = 0;             // false at the end of the stream
    virtual double get(int /*prop*/) const { return 0.0; }     // CAP_PROP_POS_MSEC of the frame just read
};
} // namespace cv
#endif

namespace lvk {

namespace detail {
// bounded hand-off queue between two pipeline stages of VideoFilter::stream
template <typename T>
class StageQueue
{
public:
    explicit StageQueue(size_t capacity) : m_Capacity(capacity) {}
    bool push(T&& v)                                            // false once the consumer has gone away
    {
        std::unique_lock<std::mutex> lock(m_Mutex);
        m_Space.wait(lock, [&] { return m_Items.size() < m_Capacity || m_Abandoned; });
        if (m_Abandoned) return false;
        m_Items.push_back(std::move(v));
        m_Ready.notify_one();
        return true;
    }
    bool pop(T& v)                                              // false when drained and the producer has finished
    {
        std::unique_lock<std::mutex> lock(m_Mutex);
        m_Ready.wait(lock, [&] { return !m_Items.empty() || m_Finished; });
        if (m_Items.empty()) return false;
        v = std::move(m_Items.front()); m_Items.pop_front();
        m_Space.notify_one();
        return true;
    }
    void finish() { std::lock_guard<std::mutex> lock(m_Mutex); m_Finished = true; m_Ready.notify_all(); }
    void abandon() { std::lock_guard<std::mutex> lock(m_Mutex); m_Abandoned = true; m_Items.clear(); m_Space.notify_all(); }
private:
    const size_t m_Capacity;
    std::mutex m_Mutex;
    std::condition_variable m_Space, m_Ready;
    std::deque<T> m_Items;
    bool m_Finished = false, m_Abandoned = false;
};
} // namespace detail

// ---------------------------------------------------------------------------------------------- Filters/VideoFilter.hpp
class VideoFilter : public Unique<VideoFilter>
{
public:
    explicit VideoFilter(const std::string& filter_name = "Identity Filter") : m_Alias(filter_name + " (" + std::to_string(this->uid()) + ")") {}
    virtual ~VideoFilter() = default;
    const std::string& alias() const { return m_Alias; }

    void apply(VideoFrame&& input, VideoFrame& output, const bool profile = false)    // VideoFilter.cpp:46-51
    {
        sync_gpu(profile); m_FrameTimer.start();
        filter(std::move(input), output);
        sync_gpu(profile); m_FrameTimer.stop();
    }
    void apply(const VideoFrame& input, VideoFrame& output, const bool profile = false) { apply(Frame(input), output, profile); }

    // VideoFilter.cpp:62-209: reader thread -> filter thread -> callback on the calling thread, at most 15 frames buffered between
    // stages; frames the filter holds back (empty output) are skipped; callback returning true ends the stream early.
    // Each stage works on its own HIP stream (thread-local context), so a frame is completed (stream sync) before it is handed on.
    void stream(cv::VideoCapture& input, const std::function<bool(Frame&)>& callback, const bool profile = false)
    {
        LVK_HIP_ASSERT(input.isOpened());
        constexpr size_t max_buffer_frames = 15;
        detail::StageQueue<Frame> input_queue(max_buffer_frames), output_queue(max_buffer_frames);

        std::thread input_thread([&] {
            Frame read_frame;
            while (input.read(read_frame))
            {
                // VideoFilter.cpp:81-85: "Assume the input frame is BGR" (what cv::VideoCapture decodes to) and stamp it with the stream
                // position -- both ALWAYS overwritten.  A capture that delivers frames in another 3-channel format (a raw YUV reader)
                // says so through stream_keeps_frame_format(true); the timestamp is the stream position either way.
                if (!m_StreamKeepsFormat || !read_frame.has_known_format()) read_frame.format = VideoFrame::BGR;
                const double stream_position = std::max(0.0, input.get(cv::CAP_PROP_POS_MSEC));
                read_frame.timestamp = static_cast<uint64_t>(stream_position * 1.0e6);              // Time::Milliseconds(..).nanoseconds()
                if (read_frame.context())
                {
                    hip::ContextLock lock(read_frame.context()->mutex());
                    read_frame.context()->check(lvk_hip_sync(read_frame.context()->get()), "VideoFilter::stream");
                }
                if (!input_queue.push(std::move(read_frame))) break;
                read_frame = Frame();
            }
            input_queue.finish();
        });
        std::thread filter_thread([&] {
            Frame input_frame, filtered_frame;
            while (input_queue.pop(input_frame))
            {
                this->apply(std::move(input_frame), filtered_frame, profile);
                if (filtered_frame.empty()) continue;
                sync_gpu(true);                                                                    // complete before it changes threads
                if (!output_queue.push(std::move(filtered_frame))) break;
                filtered_frame = Frame();
            }
            output_queue.finish();
        });
        Frame output_frame;
        while (output_queue.pop(output_frame))
            if (callback(output_frame))
            {
                // terminated by the user: starve both stages, as the reference does
                input_queue.abandon();
                output_queue.abandon();
                break;
            }
        input_thread.join();
        filter_thread.join();
    }
    // (no reference counterpart) stream(): keep the format a capture has set on its frames instead of assuming BGR
    void stream_keeps_frame_format(const bool keep) { m_StreamKeepsFormat = keep; }
    void set_timing_samples(const size_t samples) { LVK_HIP_ASSERT(samples >= 1); m_FrameTimer.set_history_size(samples); }
    const Stopwatch& timings() const { return m_FrameTimer; }

protected:
    virtual void filter(VideoFrame&& input, VideoFrame& output) { output = std::move(input); }
    virtual void sync_gpu(bool /*trigger*/) {}
private:
    Stopwatch m_FrameTimer;
    const std::string m_Alias;
    bool m_StreamKeepsFormat = false;
};
typedef VideoFilter IdentityFilter;

// ---------------------------------------------------------------------------------------------- Filters/StabilizationFilter.hpp
// Camera profile of the plugin's lens-correction filter (Modules/OBS-Plugin/Sources/Tools/CCTool.cpp:120-153)
struct CameraParameters { double fx = 0, fy = 0, cx = 0, cy = 0, k1 = 0, k2 = 0, p1 = 0, p2 = 0, k3 = 0; };

class StabilizationFilter final : public VideoFilter, public Configurable<StabilizationFilterSettings>
{
public:
    explicit StabilizationFilter(const StabilizationFilterSettings& settings = {}, int device = 0)
        : VideoFilter("Stabilization Filter"), m_Device(device), m_Ctx(std::make_shared<hip::Context>(device))
    {
        configure(settings);
    }
    // On the caller's context: the filter then shares the stream its input frames are produced on (no cross-stream fences needed).
    StabilizationFilter(const StabilizationFilterSettings& settings, const std::shared_ptr<hip::Context>& context, int device = 0)
        : VideoFilter("Stabilization Filter"), m_Device(device), m_Ctx(context)
    {
        configure(settings);
    }
    ~StabilizationFilter() override { hip::ContextLock lock(m_Ctx->mutex()); lvk_hip_stab_destroy(m_Stab); }
    StabilizationFilter(const StabilizationFilter&) = delete;
    StabilizationFilter& operator=(const StabilizationFilter&) = delete;

    void configure(const StabilizationFilterSettings& settings) override          // StabilizationFilter.cpp:42-65
    {
        hip::ContextLock lock(m_Ctx->mutex());
        const lvk_stab_settings pod = to_pod(settings);
        if (!m_Stab) m_Ctx->check(lvk_hip_stab_create(m_Ctx->get(), &pod, &m_Stab), "StabilizationFilter::configure");
        else m_Ctx->check(lvk_hip_stab_configure(m_Stab, &pod), "StabilizationFilter::configure");
        m_Settings = settings;
        static_cast<PathSmootherSettings&>(m_Settings).motion_resolution = settings.motion_resolution;
        static_cast<FrameTrackerSettings&>(m_Settings).motion_resolution = settings.motion_resolution;
        refresh_output_context();
    }

    void restart() { hip::ContextLock lock(m_Ctx->mutex()); m_Ctx->check(lvk_hip_stab_restart(m_Stab), "restart"); m_Held.clear(); }
    bool ready() const { hip::ContextLock lock(m_Ctx->mutex()); return lvk_hip_stab_ready(m_Stab) != 0; }
    void reset_context() { hip::ContextLock lock(m_Ctx->mutex()); m_Ctx->check(lvk_hip_stab_reset_context(m_Stab), "reset_context"); }
    void draw_trackers() { hip::ContextLock lock(m_Ctx->mutex()); m_Ctx->check(lvk_hip_stab_draw_trackers(m_Stab), "draw_trackers"); }          // StabilizationFilter.cpp:163-175
    void draw_motion_mesh() { hip::ContextLock lock(m_Ctx->mutex()); m_Ctx->check(lvk_hip_stab_draw_motion_mesh(m_Stab), "draw_motion_mesh"); }    // StabilizationFilter.cpp:179-188
    size_t frame_delay() const { hip::ContextLock lock(m_Ctx->mutex()); return (size_t)lvk_hip_stab_frame_delay(m_Stab); }
    cv::Rect stable_region() const
    {
        int r[4] = {0, 0, 0, 0};
        hip::ContextLock lock(m_Ctx->mutex());
        lvk_hip_stab_stable_region(m_Stab, m_LastRows, m_LastCols, r);
        return {r[0], r[1], r[2], r[3]};
    }
    const std::shared_ptr<hip::Context>& context() const { return m_Ctx; }

    // ---- MI355X additions (no reference counterpart; the reference API above is unchanged) ----------------------------------
    // Overlap mode: the bulk kernels (4:2:0 conversion, output remap) run on a second stream next to the next frame's tracking
    // (lvk_hip_stab_set_overlap).  Output frames then belong to a context on that stream: whatever consumes them through
    // frame.context() (ScalingFilter, download, clone) is ordered behind the remap that writes them.
    void set_overlap(const bool enable)
    {
        // the bulk stream belongs to a context of the facade, so that output frames stay valid after the filter is gone
        hip::ContextLock lock(m_Ctx->mutex());
        if (enable && !m_BulkCtx) m_BulkCtx = std::make_shared<hip::Context>(m_Device);
        m_Ctx->check(lvk_hip_stab_set_bulk_context(m_Stab, enable ? m_BulkCtx->get() : nullptr), "set_overlap");
        m_Overlap = enable;
        refresh_output_context();
    }
    // Fused lens pre-warp (BASELINE config 5): frames are pushed RAW, the lens correction of the plugin's LCFilter
    // (Modules/OBS-Plugin/Sources/Enhancement/LCFilter.cpp:133-192) is composed into the stabilizing remap.  Restarts the filter.
    void set_lens(const CameraParameters& p)
    {
        const lvk_camera_params c{p.fx, p.fy, p.cx, p.cy, p.k1, p.k2, p.p1, p.p2, p.k3};
        hip::ContextLock lock(m_Ctx->mutex());
        m_Ctx->check(lvk_hip_stab_set_lens(m_Stab, &c), "set_lens"); m_Held.clear();
    }
    void clear_lens() { hip::ContextLock lock(m_Ctx->mutex()); m_Ctx->check(lvk_hip_stab_set_lens(m_Stab, nullptr), "clear_lens"); m_Held.clear(); }

    // The OBS asynchronous path in one call (VisionFilter.cpp:151-212 = to_ocl -> filter -> to_obs): 4:2:0 planes in, 4:2:0 planes out.
    // `output` is released while the delay builds, exactly like the packed overload.
    using VideoFilter::apply;
    void apply(const VideoFrame420& input, VideoFrame420& output, const bool profile = false)
    {
        LVK_HIP_ASSERT(!input.empty());
        if (input.context() != m_Ctx) m_Ctx->wait_for(*input.context());      // (takes both contexts' locks: before ours is held)
        bool give_back = false;
        {
        hip::ContextLock lock(m_Ctx->mutex());
        sync_gpu(profile);
        m_LastRows = input.rows; m_LastCols = input.cols;
        // the output has the size of the DELAYED frame (a resized source still gets its queued frames at their own size, as in the reference)
        VideoFrame420 result;
        lvk_frame_info due{input.rows, input.cols, LVK_FORMAT_YUV};
        if (lvk_hip_stab_next_output(m_Stab, input.rows, input.cols, LVK_FORMAT_YUV, &due) != 1) due = lvk_frame_info{input.rows, input.cols, LVK_FORMAT_YUV};
        result.create({due.cols, due.rows}, input.nv12, m_OutCtx);
        int produced = 0; uint64_t ts = 0;
        m_Ctx->check(lvk_hip_stab_push_yuv420(m_Stab, input.y(), input.y_step(), input.u(), input.uv_step(), input.v(), input.uv_step(), input.nv12 ? 1 : 0,
                                              input.rows, input.cols, input.timestamp,
                                              result.y(), result.y_step(), result.u(), result.uv_step(), result.v(), result.uv_step(), result.rows, &produced, &ts, nullptr),
                     "StabilizationFilter::apply(4:2:0)");
        if (produced) { result.timestamp = ts; output = std::move(result); }
        else output.release();
        sync_gpu(profile);
        // the planes are consumed when the push returns only in overlap mode; otherwise their conversion is merely enqueued on our stream
        give_back = !m_Overlap && input.context() != m_Ctx;
        }
        // (both contexts' locks, taken together: ours is no longer held -- two filters on different contexts that feed each other 4:2:0 frames
        //  from two threads would otherwise each keep their own mutex while waiting for the other's, round-4 ADVICE)
        if (give_back) input.context()->wait_for(*m_Ctx);
    }

    // Frames in pinned host memory: upload_planes -> to_ocl -> filter -> to_obs -> download_planes (FrameIngest.cpp:415-474,494-602) as one
    // push.  `input` is consumed when the call returns; `output` (released while the delay builds) is complete after output.wait() or a
    // profiled call.  A streaming caller announces the following frame with prefetch() before applying the current one, so that the link
    // carries frame n + 1 while frame n is tracked (what the reader thread of VideoFilter::stream does for device frames).
    void apply(const HostFrame420& input, HostFrame420& output, const bool profile = false)
    {
        LVK_HIP_ASSERT(!input.empty());
        hip::ContextLock lock(m_Ctx->mutex());
        sync_gpu(profile);
        m_LastRows = input.rows; m_LastCols = input.cols;
        // pinned planes are expensive to allocate: outputs come from a pool and return to it when the caller drops them
        // (sized like the DELAYED frame: a resized source still gets its queued frames at their own size)
        lvk_frame_info due{input.rows, input.cols, LVK_FORMAT_YUV};
        if (lvk_hip_stab_next_output(m_Stab, input.rows, input.cols, LVK_FORMAT_YUV, &due) != 1) due = lvk_frame_info{input.rows, input.cols, LVK_FORMAT_YUV};
        HostFrame420* slot = nullptr;
        for (auto& f : m_HostPool) if (f.unique() && f.cols == due.cols && f.rows == due.rows) { slot = &f; break; }
        if (!slot) { if (m_HostPool.size() >= 8) m_HostPool.erase(m_HostPool.begin()); m_HostPool.emplace_back(); slot = &m_HostPool.back(); }
        slot->create({due.cols, due.rows}, input.nv12, m_Ctx);
        HostFrame420 result = *slot;
        int produced = 0; uint64_t ts = 0;
        m_Ctx->check(lvk_hip_stab_push_yuv420_host(m_Stab, input.y(), input.y_step(), input.u(), input.uv_step(), input.v(), input.uv_step(), input.nv12 ? 1 : 0,
                                                   input.rows, input.cols, input.timestamp,
                                                   result.y(), result.y_step(), result.u(), result.uv_step(), result.v(), result.uv_step(), result.rows, &produced, &ts, nullptr),
                     "StabilizationFilter::apply(host 4:2:0)");
        if (produced) { result.timestamp = ts; output = std::move(result); }
        else output.release();
        sync_gpu(profile);
    }
    void prefetch(const HostFrame420& next)
    {
        LVK_HIP_ASSERT(!next.empty());
        hip::ContextLock lock(m_Ctx->mutex());
        m_Ctx->check(lvk_hip_stab_prefetch_yuv420_host(m_Stab, next.y(), next.y_step(), next.u(), next.uv_step(), next.v(), next.uv_step(), next.nv12 ? 1 : 0,
                                                       next.rows, next.cols), "StabilizationFilter::prefetch");
    }

    // Device-resident planes known one frame ahead (a reader that runs ahead of the filter, as VideoFilter::stream's does): announce frame
    // n + 1, then apply frame n -- its downscale and pyramid run behind frame n's chain (lvk_hip_stab_prefetch_yuv420).  The planes must not
    // change until their apply() has returned; an announcement that the next apply() does not match is ignored.  Same pixels either way.
    void prefetch(const VideoFrame420& next)
    {
        LVK_HIP_ASSERT(!next.empty());
        if (next.context() != m_Ctx) m_Ctx->wait_for(*next.context());      // what has produced the planes so far is ahead of our stream
        hip::ContextLock lock(m_Ctx->mutex());
        m_Ctx->check(lvk_hip_stab_prefetch_yuv420(m_Stab, next.y(), next.y_step(), next.u(), next.uv_step(), next.v(), next.uv_step(), next.nv12 ? 1 : 0,
                                                  next.rows, next.cols), "StabilizationFilter::prefetch(4:2:0)");
    }

    // a frame was announced and will not be applied (the source ended, seeked or switched buffers)
    void cancel_prefetch() { hip::ContextLock lock(m_Ctx->mutex()); m_Ctx->check(lvk_hip_stab_prefetch_cancel(m_Stab), "StabilizationFilter::cancel_prefetch"); }

private:
    void filter(VideoFrame&& input, VideoFrame& output) override                    // StabilizationFilter.cpp:69-135
    {
        LVK_HIP_ASSERT(input.has_known_format());
        LVK_HIP_ASSERT(!input.empty());
        m_LastRows = input.rows; m_LastCols = input.cols;
        VideoFrame in = std::move(input);                  // input and output may alias the same object (VSFilter.cpp:358,363)
        // the frame was written on its own context's stream (an upload, an upstream filter): order our stream behind it
        if (in.context() && in.context() != m_Ctx) m_Ctx->wait_for(*in.context());
        std::shared_ptr<hip::Context> give_back;           // owner of a released frame that must still wait for our last read of it
        {
        hip::ContextLock lock(m_Ctx->mutex());
        // dst has the size of the DELAYED frame (the queue holds whole frames, StabilizationFilter.cpp:118-131; WarpMesh::apply creates dst from
        // the delayed source, WarpMesh.cpp:183-223 -> Image.cpp:53,116): a source that is resized in the middle of a stream -- VSFilter.cpp:352-364
        // does not restart its filter -- still gets its queued frames at their own size.  Pooled: a no-op in steady state.
        VideoFrame result;
        lvk_frame_info due{0, 0, 0}, emitted{0, 0, 0};
        const int will_emit = lvk_hip_stab_next_output(m_Stab, in.rows, in.cols, (int)in.format, &due);
        if (will_emit == 1) result.create({due.cols, due.rows}, CV_8UC3, m_OutCtx);
        int produced = 0; uint64_t ts = 0; const void* released = nullptr;
        m_Held.push_back({in.buffer(), in.context()});     // keep the borrowed device buffer alive while it is queued
        const int rc = lvk_hip_stab_push(m_Stab, in.device_ptr(), (int)in.step, in.rows, in.cols, in.timestamp, (int)in.format,
                                         result.device_ptr(), (int)result.step, result.rows, &produced, &ts, &released, &emitted);
        if (rc == LVK_HIP_ERR_ARG) m_Held.pop_back();      // (a refused push has queued nothing; after any other error the frame may be queued: keep it alive)
        m_Ctx->check(rc, "StabilizationFilter::filter");
        if (released)
            for (auto it = m_Held.begin(); it != m_Held.end(); ++it)
                if (it->buffer.get() == released)
                {
                    // hand the buffer back to its owner: its context's stream must not reuse it before our last read (the remap just
                    // enqueued) has run.  In overlap mode the library reports a frame only after that remap has finished.
                    if (!m_Overlap && it->ctx && it->ctx != m_Ctx) give_back = it->ctx;
                    m_Held.erase(it);
                    break;
                }
        if (produced) { result.timestamp = ts; result.format = (VideoFrame::Format)emitted.format; output = std::move(result); }
        else output.release();
        }
        if (give_back) give_back->wait_for(*m_Ctx);        // (both contexts' locks: ours is no longer held)
    }
    void sync_gpu(bool trigger) override { if (trigger) { hip::ContextLock lock(m_Ctx->mutex()); m_Ctx->check(lvk_hip_sync(m_Ctx->get()), "sync_gpu"); } }
    void refresh_output_context()
    {
        // outputs are produced on the bulk stream while overlap is on and the output is being stabilized (lvk_hip_stab_output_stream)
        m_OutCtx = (m_BulkCtx && lvk_hip_stab_output_stream(m_Stab) == lvk_hip_stream(m_BulkCtx->get())) ? m_BulkCtx : m_Ctx;
    }

    static lvk_stab_settings to_pod(const StabilizationFilterSettings& s)
    {
        lvk_stab_settings p;
        p.detection_width = s.detection_resolution.width; p.detection_height = s.detection_resolution.height;
        p.detection_regions_x = s.detection_regions.width; p.detection_regions_y = s.detection_regions.height;
        p.force_detection = s.force_detection ? 1 : 0;
        p.max_feature_density = s.max_feature_density; p.min_feature_density = s.min_feature_density; p.accumulation_rate = s.accumulation_rate;
        p.track_local_motions = s.track_local_motions ? 1 : 0; p.temporal_smoothing = s.temporal_smoothing; p.local_smoothing = s.local_smoothing;
        p.min_motion_samples = (int)s.min_motion_samples; p.acceptance_threshold = s.acceptance_threshold; p.uniformity_threshold = s.uniformity_threshold;
        p.predictive_samples = (int)s.predictive_samples; p.corrective_limit_x = s.corrective_limits.width; p.corrective_limit_y = s.corrective_limits.height;
        p.smoothing_steps = s.smoothing_steps; p.response_rate = s.response_rate;
        p.motion_width = s.motion_resolution.width; p.motion_height = s.motion_resolution.height;
        p.background[0] = (float)s.background_colour[0]; p.background[1] = (float)s.background_colour[1]; p.background[2] = (float)s.background_colour[2];
        p.crop_to_stable_region = s.crop_to_stable_region ? 1 : 0; p.stabilize_output = s.stabilize_output ? 1 : 0;
        p.min_scene_quality = s.min_scene_quality; p.min_tracking_quality = s.min_tracking_quality;
        return p;
    }

    struct HeldFrame { std::shared_ptr<void> buffer; std::shared_ptr<hip::Context> ctx; };
    int m_Device = 0;
    std::vector<HostFrame420> m_HostPool;
    std::shared_ptr<hip::Context> m_Ctx, m_BulkCtx, m_OutCtx;      // m_OutCtx: the context (stream) output frames are produced on
    lvk_hip_stab* m_Stab = nullptr;
    bool m_Overlap = false;
    std::deque<HeldFrame> m_Held;
    int m_LastRows = 0, m_LastCols = 0;
};

// ---------------------------------------------------------------------------------------------- file input
// A raw 4:2:0 clip as a cv::VideoCapture: the reference's harness streams a FILE through its filters (VideoFilter::stream,
// Filters/VideoFilter.cpp:62-209; the CLI opens a cv::VideoCapture on a path, Modules/VideoEditor/VideoProcessor.cpp:148-230).  Frames of
// a fixed size back to back, I420 (Y, U, V planes) or NV12 (Y, interleaved UV); `fps` gives CAP_PROP_POS_MSEC (frame k: k * 1000 / fps),
// which stream() turns into the frame's timestamp.  Delivered as
//   * BGR (default) -- what cv::VideoCapture decodes to and what stream() assumes (VideoFilter.cpp:81-82): the conversion is OpenCV 4.8's
//     8-bit YUV420 -> BGR (ITU-R BT.601, 20-bit fixed point, nearest chroma; imgproc/src/color_yuv.simd.hpp) on the host, as a decoder's;
//   * YUV -- packed 4:4:4 through the library's own 4:2:0 ingest (lvk_hip_ingest_yuv420 = the plugin's I4XXIngest / NV12Ingest::to_ocl),
//     with VideoFilter::stream_keeps_frame_format(true) on the filter;
// and as the raw planes in pinned host memory (read(HostFrame420&)) for StabilizationFilter::apply(const HostFrame420&, ..).
#ifndef LVK_WITH_OPENCV                                          // (with the real cv::VideoCapture the file is opened through OpenCV's backends)
class RawYuvCapture : public cv::VideoCapture
{
public:
    enum class Deliver { BGR, YUV };
    RawYuvCapture(const std::string& path, const int cols, const int rows, const double fps, const bool nv12 = false,
                  const Deliver deliver = Deliver::BGR, const std::shared_ptr<hip::Context>& ctx = nullptr)
        : m_Cols(cols), m_Rows(rows), m_Fps(fps), m_NV12(nv12), m_Deliver(deliver), m_Ctx(ctx)
    {
        LVK_HIP_ASSERT(cols > 0 && rows > 0 && cols % 2 == 0 && rows % 2 == 0 && fps > 0.0);
        m_File = std::fopen(path.c_str(), "rb");
    }
    ~RawYuvCapture() override { if (m_File) std::fclose(m_File); }
    RawYuvCapture(const RawYuvCapture&) = delete;
    RawYuvCapture& operator=(const RawYuvCapture&) = delete;

    bool isOpened() const override { return m_File != nullptr; }
    double get(int prop) const override { return (prop == cv::CAP_PROP_POS_MSEC && m_Index > 0) ? (double)(m_Index - 1) * 1000.0 / m_Fps : 0.0; }
    size_t frames_read() const { return m_Index; }
    size_t frame_bytes() const { return (size_t)m_Cols * m_Rows * 3 / 2; }

    bool read(HostFrame420& frame)                              // the planes as stored, in pinned host memory
    {
        if (!m_File) return false;
        frame.create({m_Cols, m_Rows}, m_NV12, m_Ctx);
        if (std::fread(frame.y(), 1, frame_bytes(), m_File) != frame_bytes()) return false;
        frame.timestamp = (uint64_t)((double)m_Index * 1000.0 / m_Fps * 1.0e6);
        m_Index++;
        return true;
    }

    bool read(VideoFrame& frame) override                       // packed 8UC3
    {
        if (!read(m_Stage)) return false;
        if (m_Deliver == Deliver::BGR)
        {
            m_Packed.resize((size_t)m_Cols * m_Rows * 3);
            to_bgr(m_Stage, m_Packed.data());
            frame.upload(m_Packed.data(), m_Rows, m_Cols, VideoFrame::BGR, m_Stage.timestamp, m_Ctx);
            // (upload enqueues a copy out of m_Packed: complete before the next read overwrites it)
            hip::ContextLock lock(frame.context()->mutex());
            frame.context()->check(lvk_hip_sync(frame.context()->get()), "RawYuvCapture::read");
            return true;
        }
        frame.create({m_Cols, m_Rows}, CV_8UC3, m_Ctx);
        const auto& ctx = frame.context();
        m_Planes.create({m_Cols, m_Rows}, m_NV12, ctx);
        hip::ContextLock lock(ctx->mutex());
        ctx->check(lvk_hip_upload(ctx->get(), m_Planes.y(), m_Stage.y(), frame_bytes()), "RawYuvCapture::read");
        ctx->check(lvk_hip_ingest_yuv420(ctx->get(), m_Planes.y(), m_Planes.y_step(), m_Planes.u(), m_Planes.uv_step(), m_Planes.v(), m_Planes.uv_step(),
                                         m_NV12 ? 1 : 0, m_Rows, m_Cols, frame.device_ptr(), (int)frame.step), "RawYuvCapture::read");
        ctx->check(lvk_hip_sync(ctx->get()), "RawYuvCapture::read");                   // the staging planes are reused by the next read
        frame.format = VideoFrame::YUV; frame.timestamp = m_Stage.timestamp;
        return true;
    }

private:
    static uint8_t sat8(const int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    void to_bgr(const HostFrame420& f, uint8_t* dst) const
    {
        constexpr int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SHIFT = 20;
        const uint8_t* yp = f.y(); const uint8_t* up = f.u(); const uint8_t* vp = f.v();
        const int cstep = f.uv_step(), cpix = m_NV12 ? 2 : 1;
        for (int y = 0; y < m_Rows; y++)
            for (int x = 0; x < m_Cols; x++)
            {
                const size_t ci = (size_t)(y / 2) * cstep + (size_t)(x / 2) * cpix;
                const int u = (int)up[ci] - 128, v = (int)(m_NV12 ? up[ci + 1] : vp[ci]) - 128;
                const int yy = std::max(0, (int)yp[(size_t)y * m_Cols + x] - 16) * CY;
                uint8_t* p = dst + ((size_t)y * m_Cols + x) * 3;
                p[0] = sat8((yy + (1 << (SHIFT - 1)) + CUB * u) >> SHIFT);
                p[1] = sat8((yy + (1 << (SHIFT - 1)) + CVG * v + CUG * u) >> SHIFT);
                p[2] = sat8((yy + (1 << (SHIFT - 1)) + CVR * v) >> SHIFT);
            }
    }

    int m_Cols, m_Rows; double m_Fps; bool m_NV12; Deliver m_Deliver;
    std::shared_ptr<hip::Context> m_Ctx;
    std::FILE* m_File = nullptr;
    size_t m_Index = 0;
    HostFrame420 m_Stage;
    VideoFrame420 m_Planes;
    std::vector<uint8_t> m_Packed;
};
#endif

// ---------------------------------------------------------------------------------------------- Functions/Image.hpp, Filters/ScalingFilter.hpp
// lvk::upscale (Image.cpp:155-202): EASU upsampling to `size`; size == src.size() copies.
inline void upscale(const VideoFrame& src, VideoFrame& dst, const cv::Size& size, const bool yuv = true)
{
    LVK_HIP_ASSERT(size.width >= src.cols && size.height >= src.rows);
    LVK_HIP_ASSERT(!src.empty());
    const auto& ctx = src.context();
    VideoFrame out;                                        // dst may be the object src refers to
    out.create(size, CV_8UC3, ctx);
    hip::ContextLock lock(ctx->mutex());
    ctx->check(lvk_hip_upscale(ctx->get(), src.device_ptr(), (int)src.step, src.rows, src.cols,
                               out.device_ptr(), (int)out.step, out.rows, out.cols, yuv ? 1 : 0), "upscale");
    out.timestamp = src.timestamp; out.format = src.format;
    dst = std::move(out);
}

// lvk::sharpen (Image.cpp:206-233): RCAS.  The reference's ScalingFilter passes the same frame as src and dst, where its kernel races
// neighbour reads against writes; here the result is always that of distinct buffers (src == dst goes through a fresh frame).
inline void sharpen(const VideoFrame& src, VideoFrame& dst, const float sharpness = 0.7f)
{
    LVK_HIP_ASSERT(sharpness >= 0.0f && sharpness <= 1.0f);
    LVK_HIP_ASSERT(!src.empty());
    const auto& ctx = src.context();
    VideoFrame out;
    out.create(src.size(), CV_8UC3, ctx);
    hip::ContextLock lock(ctx->mutex());
    ctx->check(lvk_hip_sharpen(ctx->get(), src.device_ptr(), (int)src.step, src.rows, src.cols, out.device_ptr(), (int)out.step, sharpness), "sharpen");
    out.timestamp = src.timestamp; out.format = src.format;
    dst = std::move(out);
}

struct ScalingFilterSettings                         // Filters/ScalingFilter.hpp:27-32
{
    cv::Size output_size = {1920, 1080};
    float sharpness = 0.8f;
    bool yuv_input = true;
};

class ScalingFilter final : public VideoFilter, public Configurable<ScalingFilterSettings>   // Filters/ScalingFilter.cpp:27-59
{
public:
    explicit ScalingFilter(const ScalingFilterSettings& settings = {}) : VideoFilter("Scaling Filter") { configure(settings); }
    explicit ScalingFilter(const cv::Size& output_size, const float sharpness = 0.8f) : VideoFilter("Scaling Filter")
    {
        ScalingFilterSettings settings; settings.output_size = output_size; settings.sharpness = sharpness;
        configure(settings);
    }
    void configure(const ScalingFilterSettings& settings) override
    {
        LVK_HIP_ASSERT(settings.sharpness >= 0.0f && settings.sharpness <= 1.0f);
        LVK_HIP_ASSERT(settings.output_size.width > 0);
        LVK_HIP_ASSERT(settings.output_size.height > 0);
        m_Settings = settings;
    }
private:
    void filter(VideoFrame&& input, VideoFrame& output) override
    {
        LVK_HIP_ASSERT(!input.empty());
        VideoFrame in = std::move(input), scaled;
        m_Ctx = in.context();
        lvk::upscale(in, scaled, m_Settings.output_size, m_Settings.yuv_input);
        lvk::sharpen(scaled, output, m_Settings.sharpness);
        output.timestamp = in.timestamp;
    }
    void sync_gpu(bool trigger) override { if (trigger && m_Ctx) { hip::ContextLock lock(m_Ctx->mutex()); m_Ctx->check(lvk_hip_sync(m_Ctx->get()), "sync_gpu"); } }
    std::shared_ptr<hip::Context> m_Ctx;
};

// ---------------------------------------------------------------------------------------------- Filters/CompositeFilter.hpp
struct CompositeFilterSettings                       // Filters/CompositeFilter.hpp:28-34
{
    std::vector<std::shared_ptr<lvk::VideoFilter>> filter_chain;
    bool save_outputs = false;
};

// A chain of filters run back to back on one frame (CompositeFilter.cpp:28-190): a disabled filter is skipped, an empty intermediate
// frame (a filter that is still filling its delay) ends the pass with an empty output, `save_outputs` keeps every stage's result.
class CompositeFilter final : public VideoFilter, public Configurable<CompositeFilterSettings>
{
public:
    explicit CompositeFilter(const CompositeFilterSettings& settings = {}) : VideoFilter("Composite Filter") { configure(settings); }
    CompositeFilter(const std::initializer_list<std::shared_ptr<lvk::VideoFilter>>& filter_chain, const CompositeFilterSettings& settings = {})
        : VideoFilter("Composite Filter")
    {
        CompositeFilterSettings chained; chained.filter_chain = filter_chain; chained.save_outputs = settings.save_outputs;
        configure(chained);
    }
    void configure(const CompositeFilterSettings& settings) override
    {
        m_Settings = settings;
        m_FilterOutputs.resize(settings.filter_chain.size());
        m_FilterRunState.assign(settings.filter_chain.size(), true);
    }
    const std::vector<std::shared_ptr<lvk::VideoFilter>>& filters() const { return m_Settings.filter_chain; }
    std::shared_ptr<lvk::VideoFilter> filters(const size_t index) { LVK_HIP_ASSERT(index < m_Settings.filter_chain.size()); return m_Settings.filter_chain[index]; }
    const std::vector<Frame>& outputs() const { return m_FilterOutputs; }
    const VideoFrame& outputs(const size_t index) { LVK_HIP_ASSERT(index < m_FilterOutputs.size()); return m_FilterOutputs[index]; }
    bool is_filter_enabled(const size_t index) { LVK_HIP_ASSERT(index < m_FilterRunState.size()); return m_FilterRunState[index]; }
    void disable_filter(const size_t index) { LVK_HIP_ASSERT(index < m_FilterRunState.size()); m_FilterRunState[index] = false; }
    void enable_filter(const size_t index) { LVK_HIP_ASSERT(index < m_FilterRunState.size()); m_FilterRunState[index] = true; }
    void enable_all_filters() { m_FilterRunState.assign(m_FilterRunState.size(), true); }
    size_t filter_count() const { return m_Settings.filter_chain.size(); }

private:
    void filter(VideoFrame&& input, VideoFrame& output) override
    {
        LVK_HIP_ASSERT(!input.empty());
        VideoFrame current = std::move(input);
        for (size_t i = 0; i < m_Settings.filter_chain.size(); i++)
        {
            if (!m_FilterRunState[i]) continue;
            if (current.empty()) break;                                    // a stage upstream has nothing to hand on yet
            m_Settings.filter_chain[i]->apply(std::move(current), m_FilterOutputs[i]);
            current = m_Settings.save_outputs ? m_FilterOutputs[i].clone() : std::move(m_FilterOutputs[i]);
        }
        output = std::move(current);
    }
    std::vector<bool> m_FilterRunState;
    std::vector<Frame> m_FilterOutputs;
};

// north-star aliases (BASELINE.json names from another LVK snapshot; SURVEY.md name mapping)
using PathStabilizerSettings = PathSmootherSettings;
using GridDetectorSettings = FeatureDetectorSettings;

} // namespace lvk

// Math/Homography.hpp, Math/WarpMesh.hpp, the lvk::remap launchers of Functions/Image.hpp
#include "WarpMesh.hpp"
