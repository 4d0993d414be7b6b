// API-conformance translation unit: reproduces, against include/lvk/LiveVisionKit.hpp, the way the reference's
// callers use the stabilizer (libobs is not available here, so the plugin itself cannot be compiled):
//   Modules/OBS-Plugin/Sources/Stabilisation/VSFilter.hpp:54        StabilizationFilter m_Filter (by value)
//   Modules/OBS-Plugin/Sources/Stabilisation/VSFilter.cpp:235-294   reconfigure(lambda) with every settings field
//   .../VSFilter.cpp:304,327-333,347                                 frame_delay(), settings(), set_timing_samples(30)
//   .../VSFilter.cpp:352-364                                         apply(std::move(frame), frame[, true]), draw_*()
//   .../VSFilter.cpp:368-383                                         timings().average()/deviation().milliseconds(), stable_region()
//   Modules/VideoEditor/FilterParser.tpp:51-64                       make_shared<StabilizationFilter>() + Configurable<>::configure
// With -DRUN_ON_GPU it also runs a short synthetic stream (needs a GPU).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "lvk/LiveVisionKit.hpp"

namespace lvk {

class VSFilterLike
{
public:
    VSFilterLike() { m_Filter.set_timing_samples(30); }

    void configure(bool field_subsystem, bool strict, float crop_x, float crop_y, int samples, bool apply_crop, bool disabled, bool test_mode)
    {
        m_TestMode = test_mode;
        m_Filter.reconfigure([&](StabilizationFilterSettings& stab_settings) {
            stab_settings.crop_to_stable_region = apply_crop && !m_TestMode;
            stab_settings.stabilize_output = !disabled;
            stab_settings.corrective_limits.height = crop_y;
            stab_settings.corrective_limits.width = crop_x;
            stab_settings.predictive_samples = samples;
            stab_settings.background_colour[0] = 105.0f;
            stab_settings.background_colour[1] = 212.0f;
            stab_settings.background_colour[2] = 235.0f;
            if (field_subsystem)
            {
                stab_settings.detection_resolution = {480, 270};
                stab_settings.acceptance_threshold = 10.0f;
                stab_settings.track_local_motions = true;
                stab_settings.motion_resolution = {16, 16};
                stab_settings.detection_regions = {2, 2};
                stab_settings.max_feature_density = 0.12f;
                stab_settings.min_feature_density = 0.06f;
                stab_settings.accumulation_rate = 3.0f;
            }
            else
            {
                stab_settings.detection_resolution = {480, 270};
                stab_settings.acceptance_threshold = 3.0f;
                stab_settings.track_local_motions = false;
                stab_settings.motion_resolution = {2, 2};
                stab_settings.detection_regions = {2, 1};
                stab_settings.max_feature_density = 0.12f;
                stab_settings.min_feature_density = 0.04f;
                stab_settings.accumulation_rate = 3.0f;
            }
            if (strict) { stab_settings.min_scene_quality = 0.95f; stab_settings.min_tracking_quality = 0.35f; }
            else { stab_settings.min_scene_quality = 0.40f; stab_settings.min_tracking_quality = 0.20f; }
        });
        const auto new_stream_delay = static_cast<int>((1000.0f / 60.0f) * static_cast<float>(m_Filter.frame_delay()));
        std::printf("delay %d ms, predictive %zu, crop (%.1f%%, %.1f%%), crop_out %d, disabled %d\n", new_stream_delay,
                    m_Filter.settings().predictive_samples, m_Filter.settings().corrective_limits.width * 100.0f,
                    m_Filter.settings().corrective_limits.height * 100.0f, (int)m_Filter.settings().crop_to_stable_region,
                    (int)!m_Filter.settings().stabilize_output);
    }

    void filter(Frame& frame)
    {
        if (m_TestMode)
        {
            m_Filter.apply(std::move(frame), frame, true);
            m_Filter.draw_motion_mesh();
            m_Filter.draw_trackers();
            const double frame_time_ms = m_Filter.timings().average().milliseconds();
            const double deviation_ms = m_Filter.timings().deviation().milliseconds();
            const auto& crop_region = m_Filter.stable_region();
            const cv::Point text_at = crop_region.tl() + cv::Point(5, 40);
            (void)frame_time_ms; (void)deviation_ms; (void)text_at;
        }
        else m_Filter.apply(std::move(frame), frame);
    }

private:
    StabilizationFilter m_Filter;      // held BY VALUE, as VSFilter.hpp:54
    bool m_TestMode = false;
};

} // namespace lvk

#ifdef RUN_ON_GPU
namespace {

// The settings tests/test_golden.py runs the golden clip with: library defaults, then the OBS preset `name` with
// predictive_samples 3, tracking at 320 x 180 and the relaxed QA thresholds (oracle_lib.preset(name, **STAB_OVER)).
lvk::StabilizationFilterSettings golden_settings(const std::string& name)
{
    lvk::StabilizationFilterSettings s;
    s.detection_resolution = {320, 180};
    s.max_feature_density = 0.12f; s.accumulation_rate = 3.0f;
    s.min_scene_quality = 0.4f; s.min_tracking_quality = 0.2f;
    s.corrective_limits = {0.05f, 0.05f};
    s.crop_to_stable_region = true;
    s.background_colour = {105, 212, 235};
    s.predictive_samples = 3;
    if (name == "field")
    {
        s.acceptance_threshold = 10.0f; s.track_local_motions = true; s.motion_resolution = {16, 16};
        s.detection_regions = {2, 2}; s.min_feature_density = 0.06f;
    }
    else
    {
        s.acceptance_threshold = 3.0f; s.track_local_motions = false; s.motion_resolution = {2, 2};
        s.detection_regions = {2, 1}; s.min_feature_density = 0.04f;
    }
    return s;
}

// --golden <clip.raw> <n> <rows> <cols> <out.raw>: every frame of the packed YUV clip through lvk::StabilizationFilter::apply (both
// presets, then the 4:2:0 overload with overlap on); emitted frames are appended to out.raw for the Python side to compare with
// tests/golden/stabilizer.npz / the oracle.
int run_golden(const char* clip_path, int n, int rows, int cols, const char* out_path)
{
    std::vector<uint8_t> clip((size_t)n * rows * cols * 3);
    FILE* f = std::fopen(clip_path, "rb");
    if (!f || std::fread(clip.data(), 1, clip.size(), f) != clip.size()) { std::printf("golden: cannot read the clip\n"); return 1; }
    std::fclose(f);
    FILE* out = std::fopen(out_path, "wb");
    if (!out) return 1;
    std::vector<uint8_t> host((size_t)rows * cols * 3);
    for (const char* name : {"homography", "field"})
    {
        lvk::StabilizationFilter filter;                        // library defaults first, then configure: the order the golden run uses
        filter.configure(golden_settings(name));
        int emitted = 0;
        for (int i = 0; i < n; i++)
        {
            lvk::Frame frame;
            frame.upload(clip.data() + (size_t)i * rows * cols * 3, rows, cols, lvk::VideoFrame::YUV, 1000 + i);
            filter.apply(std::move(frame), frame);
            if (frame.empty()) continue;
            if (frame.timestamp != (uint64_t)(1000 + i - 3)) { std::printf("golden: bad timestamp\n"); return 1; }
            frame.download(host.data());
            std::fwrite(host.data(), 1, host.size(), out);
            emitted++;
        }
        std::printf("golden %s: %d frames\n", name, emitted);
    }
    // The same clip through VideoFilter::stream (Filters/VideoFilter.cpp:62-209): reader thread -> filter thread -> this thread's callback,
    // frames crossing threads (and HIP streams) twice.  Appended to out.raw like the apply() runs: the Python side holds them to the same
    // sha-256s -- stream() is pixel-checked, not only counted.
    struct ClipCapture : cv::VideoCapture
    {
        const std::vector<uint8_t>& clip; int n, rows, cols, i = 0;
        ClipCapture(const std::vector<uint8_t>& c, int n_, int r, int co) : clip(c), n(n_), rows(r), cols(co) {}
        bool isOpened() const override { return true; }
        double get(int) const override { return (double)(i - 1); }          // frame k sits at k ms: stream() stamps it k * 1e6 ns (VideoFilter.cpp:84-85)
        bool read(lvk::VideoFrame& f) override
        {
            if (i >= n) return false;
            f.upload(clip.data() + (size_t)i * rows * cols * 3, rows, cols, lvk::VideoFrame::YUV, 1000 + i);
            i++;
            return true;
        }
    };
    for (const char* name : {"homography", "field"})
    {
        lvk::StabilizationFilter filter;
        filter.configure(golden_settings(name));
        filter.stream_keeps_frame_format(true);                 // the golden clip is packed YUV (the reference's reader would call it BGR)
        ClipCapture cap(clip, n, rows, cols);
        int emitted = 0; bool ok = true;
        filter.stream(cap, [&](lvk::Frame& frame) {
            ok = ok && frame.timestamp == (uint64_t)emitted * 1000000ull && frame.format == lvk::VideoFrame::YUV;
            frame.download(host.data());
            std::fwrite(host.data(), 1, host.size(), out);
            emitted++;
            return false;
        });
        if (!ok) { std::printf("golden stream: bad timestamp\n"); return 1; }
        std::printf("golden stream %s: %d frames\n", name, emitted);
    }
    std::fclose(out);
    return 0;
}

// --resize <in.bin> <out.bin>: a packed stream whose frame size (and format) changes in the middle -- an OBS source that is resized;
// VSFilter.cpp:352-364 keeps calling apply(std::move(frame), frame) on the same filter.  The queue holds whole frames
// (StabilizationFilter.cpp:118-131) and dst is created from the DELAYED source (WarpMesh.cpp:183-223 -> Image.cpp:53,116): every frame
// leaves, at its own size and format.  Records in / out: int32 rows, cols, format; uint64 timestamp; rows * cols * 3 bytes.  Two passes
// (plain, overlap); the Python side holds every emitted frame to the oracle's frame of the same timestamp.
int run_resize(const char* in_path, const char* out_path)
{
    struct Rec { int32_t rows, cols, format; uint64_t ts; std::vector<uint8_t> px; };
    std::vector<Rec> recs;
    FILE* f = std::fopen(in_path, "rb");
    if (!f) { std::printf("resize: cannot read the clip\n"); return 1; }
    for (;;)
    {
        Rec r; int32_t head[3];
        if (std::fread(head, sizeof(int32_t), 3, f) != 3) break;
        r.rows = head[0]; r.cols = head[1]; r.format = head[2];
        if (std::fread(&r.ts, sizeof(uint64_t), 1, f) != 1) return 1;
        r.px.resize((size_t)r.rows * r.cols * 3);
        if (std::fread(r.px.data(), 1, r.px.size(), f) != r.px.size()) return 1;
        recs.push_back(std::move(r));
    }
    std::fclose(f);
    FILE* out = std::fopen(out_path, "wb");
    if (!out) return 1;
    std::vector<uint8_t> host;
    for (int pass = 0; pass < 2; pass++)
    {
        lvk::StabilizationFilter filter;
        filter.configure(golden_settings("homography"));
        if (pass == 1) filter.set_overlap(true);
        int emitted = 0;
        for (const Rec& r : recs)
        {
            lvk::Frame frame;
            frame.upload(r.px.data(), r.rows, r.cols, (lvk::VideoFrame::Format)r.format, r.ts);
            filter.apply(std::move(frame), frame);                       // VSFilter.cpp:358,363: input and output are the same object
            if (frame.empty()) continue;
            const int32_t head[3] = {frame.rows, frame.cols, (int32_t)frame.format};
            host.resize((size_t)frame.rows * frame.cols * 3);
            frame.download(host.data());
            std::fwrite(head, sizeof(int32_t), 3, out); std::fwrite(&frame.timestamp, sizeof(uint64_t), 1, out);
            std::fwrite(host.data(), 1, host.size(), out);
            emitted++;
        }
        std::printf("resize pass %d: %d of %zu frames emitted\n", pass, emitted, recs.size());
    }
    std::fclose(out);
    return 0;
}

// --resize420 <in.bin> <out.bin>: the same for the plugin's 4:2:0 wire format -- apply(const VideoFrame420&, VideoFrame420&) on device planes (overlap on)
// and apply(const HostFrame420&, HostFrame420&) on pinned host planes -- over an I420 stream whose size changes twice.  Records in / out: int32 rows,
// cols; uint64 timestamp; rows * cols * 3 / 2 bytes (Y, U, V).  Every frame must leave, at its own size.
int run_resize420(const char* in_path, const char* out_path)
{
    struct Rec { int32_t rows, cols; uint64_t ts; std::vector<uint8_t> px; };
    std::vector<Rec> recs;
    FILE* f = std::fopen(in_path, "rb");
    if (!f) { std::printf("resize420: cannot read the clip\n"); return 1; }
    for (;;)
    {
        Rec r; int32_t head[2];
        if (std::fread(head, sizeof(int32_t), 2, f) != 2) break;
        r.rows = head[0]; r.cols = head[1];
        if (std::fread(&r.ts, sizeof(uint64_t), 1, f) != 1) return 1;
        r.px.resize((size_t)r.rows * r.cols * 3 / 2);
        if (std::fread(r.px.data(), 1, r.px.size(), f) != r.px.size()) return 1;
        recs.push_back(std::move(r));
    }
    std::fclose(f);
    FILE* out = std::fopen(out_path, "wb");
    if (!out) return 1;
    std::vector<uint8_t> host;
    auto emit = [&](int rows, int cols, uint64_t ts, const uint8_t* px) {
        const int32_t head[2] = {rows, cols};
        std::fwrite(head, sizeof(int32_t), 2, out); std::fwrite(&ts, sizeof(uint64_t), 1, out);
        std::fwrite(px, 1, (size_t)rows * cols * 3 / 2, out);
    };
    for (int pass = 0; pass < 2; pass++)
    {
        lvk::StabilizationFilter filter;
        filter.configure(golden_settings("homography"));
        filter.set_overlap(true);
        int emitted = 0;
        for (const Rec& r : recs)
        {
            if (pass == 0)
            {
                lvk::VideoFrame420 in, res;
                in.upload(r.px.data(), r.rows, r.cols, false, r.ts);
                filter.apply(in, res);
                if (res.empty()) continue;
                host.resize((size_t)res.rows * res.cols * 3 / 2);
                res.download(host.data());
                emit(res.rows, res.cols, res.timestamp, host.data());
            }
            else
            {
                lvk::HostFrame420 in, res;
                in.create({r.cols, r.rows}, false);
                std::memcpy(in.y(), r.px.data(), r.px.size()); in.timestamp = r.ts;
                filter.apply(in, res, true);
                if (res.empty()) continue;
                res.wait();
                emit(res.rows, res.cols, res.timestamp, res.y());
            }
            emitted++;
        }
        std::printf("resize420 pass %d: %d of %zu frames emitted\n", pass, emitted, recs.size());
    }
    std::fclose(out);
    return 0;
}

// A chain whose stages run on DIFFERENT streams with nothing but the facade's fences between them: asynchronous upload on the
// thread's context -> ScalingFilter (same context) -> StabilizationFilter (its own context; overlap: a third stream) -> ScalingFilter
// on the output frame's context -> download.  Compared with the same chain run with a full synchronisation after every stage.
int run_chain_race_check()
{
    const int rows = 270, cols = 480, n = 14;
    std::vector<std::vector<uint8_t>> frames(n, std::vector<uint8_t>((size_t)rows * cols * 3));
    for (int i = 0; i < n; i++)
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
            {
                uint8_t* p = &frames[i][((size_t)y * cols + x) * 3];
                const int xs = x + 2 * (i % 3), ys = y + (i % 2);
                p[0] = (uint8_t)((((xs / 14) + (ys / 14)) % 2) ? 205 : 35 + (xs * 5 + ys * 9) % 31); p[1] = (uint8_t)(100 + (x >> 3)); p[2] = (uint8_t)(90 + (y >> 2));
            }
    auto run = [&](bool synchronous, bool overlap, std::vector<std::vector<uint8_t>>& outs) {
        lvk::StabilizationFilterSettings st; st.predictive_samples = 3; st.detection_resolution = {480, 270}; st.track_local_motions = false;
        auto pre = std::make_shared<lvk::ScalingFilter>(cv::Size(960, 540), 0.6f);
        auto stab = std::make_shared<lvk::StabilizationFilter>(st);
        stab->set_overlap(overlap);
        auto post = std::make_shared<lvk::ScalingFilter>(cv::Size(1280, 720), 0.3f);
        lvk::CompositeFilter chain({pre, stab, post});
        for (int i = 0; i < n; i++)
        {
            lvk::Frame frame;
            frame.upload(frames[i].data(), rows, cols, lvk::VideoFrame::YUV, i);
            if (synchronous)
            {
                // every stage completed (device idle) before the next one starts
                frame.context()->check(lvk_hip_sync(frame.context()->get()), "chain");
                pre->apply(std::move(frame), frame, true);
                stab->apply(std::move(frame), frame, true);
                if (!frame.empty()) post->apply(std::move(frame), frame, true);
            }
            else chain.apply(std::move(frame), frame);
            if (frame.empty()) continue;
            outs.emplace_back((size_t)frame.rows * frame.cols * 3);
            frame.download(outs.back().data());
        }
    };
    for (bool overlap : {false, true})
    {
        std::vector<std::vector<uint8_t>> a, b;
        run(true, overlap, a); run(false, overlap, b);
        if (a.size() != (size_t)n - 3 || a.size() != b.size()) { std::printf("chain: %zu / %zu frames\n", a.size(), b.size()); return 1; }
        for (size_t k = 0; k < a.size(); k++) if (a[k] != b[k]) { std::printf("chain: frame %zu differs between the synchronous and the free-running chain (overlap %d)\n", k, (int)overlap); return 1; }
    }
    std::printf("chain ok: free-running == synchronous, with and without overlap\n");
    return 0;
}

// Concurrent filter instances in one process: the plugin holds one StabilizationFilter per source and runs each on whatever thread OBS
// gives it (VSFilter.hpp:54, Interop/VisionFilter.cpp:157-162; SURVEY 8b: "re-entrant per instance, no shared mutable globals").  Two
// filters on two std::threads, different clips and presets, free-running with overlap on: each stream's emitted bytes must equal its own
// single-threaded run -- with a context per filter, and with BOTH filters on ONE shared hip::Context.
int run_two_threads_check()
{
    const int rows = 360, cols = 640, n = 40;
    auto paint = [&](std::vector<uint8_t>& img, int stream, int i) {
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
            {
                uint8_t* p = &img[((size_t)y * cols + x) * 3];
                const int xs = x + (stream ? 3 : 2) * (i % 4) + stream * 17, ys = y + (i % 3) + stream * 5, cell = stream ? 18 : 13;
                p[0] = (uint8_t)((((xs / cell) + (ys / cell)) % 2) ? 200 - 20 * stream : 35 + (xs * 5 + ys * (9 + 2 * stream)) % 31);
                p[1] = (uint8_t)(100 + (x >> 3) + 9 * stream); p[2] = (uint8_t)(90 + (y >> 2));
            }
    };
    auto settings_of = [](int stream) {
        lvk::StabilizationFilterSettings st;
        st.predictive_samples = 3 + stream; st.detection_resolution = {480, 270};
        st.min_scene_quality = 0.4f; st.min_tracking_quality = 0.2f;
        if (stream == 0) { st.track_local_motions = false; st.motion_resolution = {2, 2}; st.detection_regions = {2, 1}; st.acceptance_threshold = 3.0f; }
        else { st.track_local_motions = true; st.motion_resolution = {16, 16}; st.detection_regions = {2, 2}; st.acceptance_threshold = 10.0f; }
        return st;
    };
    using Outputs = std::vector<std::vector<uint8_t>>;
    auto run_stream = [&](int stream, const std::shared_ptr<lvk::hip::Context>& ctx, Outputs& outs) {
        std::unique_ptr<lvk::StabilizationFilter> filter;
        if (ctx) filter = std::make_unique<lvk::StabilizationFilter>(lvk::StabilizationFilterSettings{}, ctx);
        else filter = std::make_unique<lvk::StabilizationFilter>();
        filter->configure(settings_of(stream));
        filter->set_overlap(true);
        std::vector<uint8_t> img((size_t)rows * cols * 3);
        std::vector<lvk::Frame> kept;
        for (int i = 0; i < n; i++)
        {
            paint(img, stream, i);
            lvk::Frame frame;
            frame.upload(img.data(), rows, cols, lvk::VideoFrame::YUV, 100 * stream + i, ctx);
            filter->apply(std::move(frame), frame);                   // free-running: nothing waits for the GPU inside the loop
            if (!frame.empty()) kept.push_back(std::move(frame));
        }
        for (auto& f : kept) { outs.emplace_back((size_t)rows * cols * 3); f.download(outs.back().data()); }
    };
    Outputs want[2];
    for (int s = 0; s < 2; s++) run_stream(s, nullptr, want[s]);
    if (want[0].size() != (size_t)n - 3 || want[1].size() != (size_t)n - 4) { std::printf("threads: reference runs emitted %zu / %zu\n", want[0].size(), want[1].size()); return 1; }
    if (want[0][10] == want[1][10]) { std::printf("threads: the two streams are not distinct\n"); return 1; }
    for (int shared = 0; shared < 2; shared++)
        for (int round = 0; round < 2; round++)
        {
            Outputs got[2];
            std::shared_ptr<lvk::hip::Context> ctx = shared ? std::make_shared<lvk::hip::Context>() : nullptr;
            std::thread t0([&] { run_stream(0, ctx, got[0]); });
            std::thread t1([&] { run_stream(1, ctx, got[1]); });
            t0.join(); t1.join();
            for (int s = 0; s < 2; s++)
            {
                if (got[s].size() != want[s].size()) { std::printf("threads: stream %d emitted %zu frames (shared %d)\n", s, got[s].size(), shared); return 1; }
                for (size_t k = 0; k < got[s].size(); k++)
                    if (got[s][k] != want[s][k]) { std::printf("threads: stream %d frame %zu differs from its single-threaded run (shared context %d, round %d)\n", s, k, shared, round); return 1; }
            }
        }
    std::printf("threads ok: 2 filters on 2 threads == their single-threaded runs, own contexts and one shared context\n");
    return 0;
}

// Two filters on DIFFERENT contexts that feed each other 4:2:0 frames from two threads, overlap off (round-4 ADVICE, medium): filter A
// (context A, thread 0) consumes planes that live on context B, filter B (context B, thread 1) planes that live on context A.  The
// non-overlap apply(const VideoFrame420&) ends with a cross-context wait that takes BOTH contexts' mutexes: held while the filter's own is
// still locked it is an ABBA deadlock between the two threads; the facade releases its own lock first.  A watchdog turns a hang into a failure;
// the outputs must equal the single-threaded runs.
int run_cross_context_420_check()
{
    // small frames, many pushes: the window in which both threads sit in their trailing waits is a microsecond wide -- 1 500 pushes per thread and
    // round make it certain that the pre-round-5 locking hangs here (checked: scripts/probes/cross_context_lock_check.sh)
    const int rows = 136, cols = 240, n = 1500;
    auto planes_of = [&](int stream, int i, std::vector<uint8_t>& buf) {
        buf.resize((size_t)rows * cols * 3 / 2);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
            {
                const int xs = x + (2 + stream) * (i % 4) + 11 * stream, ys = y + (i % 3), cell = 12 + 3 * stream;
                buf[(size_t)y * cols + x] = (uint8_t)((((xs / cell) + (ys / cell)) % 2) ? 200 : 36 + (xs * 5 + ys * 9) % 31);
            }
        std::memset(buf.data() + (size_t)rows * cols, 120 + 8 * stream, (size_t)rows * cols / 2);
    };
    lvk::StabilizationFilterSettings st; st.predictive_samples = 2; st.detection_resolution = {480, 270}; st.track_local_motions = false;
    st.min_scene_quality = 0.4f; st.min_tracking_quality = 0.2f;
    using Outputs = std::vector<std::vector<uint8_t>>;
    // stream s: frames uploaded on `frames_ctx`, filtered on `filter_ctx`
    auto run = [&](int s, const std::shared_ptr<lvk::hip::Context>& frames_ctx, const std::shared_ptr<lvk::hip::Context>& filter_ctx, Outputs& outs) {
        lvk::StabilizationFilter filter(st, filter_ctx);                       // overlap stays OFF: the path with the trailing cross-context wait
        std::vector<uint8_t> buf;
        for (int i = 0; i < n; i++)
        {
            planes_of(s, i, buf);
            lvk::VideoFrame420 in, out;
            in.upload(buf.data(), rows, cols, false, 100 * s + i, frames_ctx);
            filter.apply(in, out);
            if (out.empty() || i % 100 != 99) continue;                      // (a download synchronises: compare one frame in a hundred)
            outs.emplace_back((size_t)rows * cols * 3 / 2);
            out.download(outs.back().data());
        }
    };
    Outputs want[2];
    for (int s = 0; s < 2; s++) { auto c = std::make_shared<lvk::hip::Context>(); run(s, c, c, want[s]); }
    if (want[0].size() != (size_t)n / 100 || want[0][5] == want[1][5]) { std::printf("cross-context: reference runs implausible\n"); return 1; }
    for (int round = 0; round < 3; round++)
    {
        auto A = std::make_shared<lvk::hip::Context>(), B = std::make_shared<lvk::hip::Context>();
        Outputs got[2];
        std::atomic<int> done{0};
        std::thread t0([&] { run(0, B, A, got[0]); done++; });              // filter on A, frames on B
        std::thread t1([&] { run(1, A, B, got[1]); done++; });              // filter on B, frames on A
        for (int ms = 0; ms < 30000 && done.load() < 2; ms += 5) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        if (done.load() < 2) { std::printf("cross-context: DEADLOCK (two filters on two contexts feeding each other 4:2:0 frames, round %d)\n", round); std::fflush(stdout); std::_Exit(1); }
        t0.join(); t1.join();
        for (int s = 0; s < 2; s++)
        {
            if (got[s].size() != want[s].size()) { std::printf("cross-context: stream %d emitted %zu frames\n", s, got[s].size()); return 1; }
            for (size_t k = 0; k < got[s].size(); k++)
                if (got[s][k] != want[s][k]) { std::printf("cross-context: stream %d frame %zu differs from its single-context run (round %d)\n", s, k, round); return 1; }
        }
    }
    std::printf("cross-context ok: 2 filters on 2 contexts feeding each other 4:2:0 frames from 2 threads, no overlap: no deadlock, bytes equal\n");
    return 0;
}

// A clip FILE through VideoFilter::stream (the reference's harness: VideoProcessor.cpp:148-230 opens a cv::VideoCapture on a path and
// streams it): lvk::RawYuvCapture on a raw I420 file.  Delivered as YUV the emitted frames must equal apply() on the same frames ingested
// by lvk_hip_ingest_yuv420, stamped with the stream position; delivered as BGR (the reference's assumption) the run must complete with BGR
// frames; read(HostFrame420&) feeds the host entry point and must emit the planes of the device 4:2:0 path.
static int run_file_input_case(const char* dir, const bool nv12)
{
    const int rows = 360, cols = 640, n = 18; const double fps = 50.0;
    const std::string path = std::string(dir) + (nv12 ? "/clip_nv12.yuv" : "/clip_i420.yuv");
    std::vector<std::vector<uint8_t>> frames(n, std::vector<uint8_t>((size_t)rows * cols * 3 / 2));
    {
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) { std::printf("file input: cannot write %s\n", path.c_str()); return 1; }
        for (int i = 0; i < n; i++)
        {
            uint8_t* d = frames[i].data();
            for (int y = 0; y < rows; y++)
                for (int x = 0; x < cols; x++)
                {
                    const int xs = x + 2 * (i % 3), ys = y + (i % 2);
                    d[(size_t)y * cols + x] = (uint8_t)((((xs / 15) + (ys / 15)) % 2) ? 190 : 45 + (xs * 3 + ys * 7) % 37);
                }
            for (int k = 0; k < rows * cols / 2; k++) d[(size_t)rows * cols + k] = (uint8_t)(100 + (k * 3 + i) % 50);
            std::fwrite(d, 1, frames[i].size(), f);
        }
        std::fclose(f);
    }
    lvk::StabilizationFilterSettings st; st.predictive_samples = 4;
    // expected: apply() on the frames ingested by the library's own 4:2:0 -> 4:4:4 conversion
    std::vector<std::vector<uint8_t>> want;
    {
        lvk::StabilizationFilter filter(st);
        for (int i = 0; i < n; i++)
        {
            lvk::VideoFrame420 planes; planes.upload(frames[i].data(), rows, cols, nv12, i);
            lvk::Frame frame; frame.create({cols, rows}, CV_8UC3, planes.context());
            planes.context()->check(lvk_hip_ingest_yuv420(planes.context()->get(), planes.y(), planes.y_step(), planes.u(), planes.uv_step(), planes.v(), planes.uv_step(), nv12 ? 1 : 0,
                                                          rows, cols, frame.device_ptr(), (int)frame.step), "file input");
            frame.format = lvk::VideoFrame::YUV; frame.timestamp = i;
            filter.apply(std::move(frame), frame);
            if (frame.empty()) continue;
            want.emplace_back((size_t)rows * cols * 3); frame.download(want.back().data());
        }
    }
    {
        lvk::StabilizationFilter filter(st);
        filter.stream_keeps_frame_format(true);
        lvk::RawYuvCapture cap(path, cols, rows, fps, nv12, lvk::RawYuvCapture::Deliver::YUV);
        if (!cap.isOpened()) { std::printf("file input: capture not opened\n"); return 1; }
        size_t k = 0; bool ok = true;
        std::vector<uint8_t> host((size_t)rows * cols * 3);
        filter.stream(cap, [&](lvk::Frame& frame) {
            const uint64_t ts = (uint64_t)((double)k * 1000.0 / fps * 1.0e6);           // frame k of the file: k / fps seconds
            frame.download(host.data());
            ok = ok && k < want.size() && frame.timestamp == ts && frame.format == lvk::VideoFrame::YUV && host == want[k];
            k++;
            return false;
        });
        if (!ok || k != want.size() || cap.frames_read() != (size_t)n) { std::printf("file input: YUV stream differs (%zu of %zu frames)\n", k, want.size()); return 1; }
    }
    {
        lvk::StabilizationFilter filter(st);                            // the reference's reader: frames assumed BGR
        lvk::RawYuvCapture cap(path, cols, rows, fps, nv12);
        size_t k = 0; bool ok = true;
        filter.stream(cap, [&](lvk::Frame& frame) { ok = ok && frame.format == lvk::VideoFrame::BGR && frame.cols == cols; k++; return false; });
        if (!ok || k != (size_t)n - 4) { std::printf("file input: BGR stream emitted %zu frames\n", k); return 1; }
    }
    {
        // the raw planes into the host entry point against the device 4:2:0 overload
        std::vector<std::vector<uint8_t>> want420;
        {
            lvk::StabilizationFilter filter(st);
            for (int i = 0; i < n; i++)
            {
                lvk::VideoFrame420 in, out; in.upload(frames[i].data(), rows, cols, nv12, i);
                filter.apply(in, out);
                if (out.empty()) continue;
                want420.emplace_back(frames[i].size()); out.download(want420.back().data());
            }
        }
        {
            // device planes announced one apply() ahead (StabilizationFilter::prefetch(const VideoFrame420&)): the same bytes
            lvk::StabilizationFilter ahead(st);
            std::vector<lvk::VideoFrame420> dev((size_t)n);
            for (int i = 0; i < n; i++) dev[(size_t)i].upload(frames[i].data(), rows, cols, nv12, i);
            size_t e = 0;
            for (int i = 0; i < n; i++)
            {
                if (i + 1 < n) ahead.prefetch(dev[(size_t)i + 1]);
                lvk::VideoFrame420 out; ahead.apply(dev[(size_t)i], out);
                if (out.empty()) continue;
                std::vector<uint8_t> got(frames[i].size()); out.download(got.data());
                if (e >= want420.size() || got != want420[e]) { std::printf("file input: announced device frame %zu differs\n", e); return 1; }
                e++;
            }
            if (e != want420.size()) { std::printf("file input: the announced run emitted %zu\n", e); return 1; }
        }
        lvk::StabilizationFilter filter(st);
        lvk::RawYuvCapture cap(path, cols, rows, fps, nv12);
        lvk::HostFrame420 in; size_t k = 0;
        while (cap.read(in))
        {
            lvk::HostFrame420 out;
            filter.apply(in, out, true);
            if (out.empty()) continue;
            if (k >= want420.size() || std::memcmp(out.y(), want420[k].data(), out.bytes()) != 0) { std::printf("file input: host frame %zu differs\n", k); return 1; }
            k++;
        }
        if (k != want420.size()) { std::printf("file input: host path emitted %zu\n", k); return 1; }
    }
    std::printf("file input ok (%s): %zu frames through stream() == apply(), BGR delivery runs, host planes == device planes\n", nv12 ? "NV12" : "I420", want.size());
    return 0;
}

int run_file_input_check(const char* dir)
{
    return (run_file_input_case(dir, false) != 0 || run_file_input_case(dir, true) != 0) ? 1 : 0;
}

// --bench <rows> <cols> <frames> <clip.i420> <distinct>: steady-state frames/s through lvk::StabilizationFilter::apply with the frames resident in HBM
// (packed overload, then the 4:2:0 overload with overlap on): what bench.py measures over the C-ABI, here through the facade.
int run_bench(int rows, int cols, int count, const char* clip_path, int distinct)
{
    // clip_path: `distinct` I420 frames (Y, U, V planes back to back) written by the Python side from the bench's own clip generator, so
    // that this loop and the C-ABI loop of tests/test_facade_cpp.py see the same pictures
    std::vector<uint8_t> host((size_t)rows * cols * 3), p420((size_t)rows * cols * 3 / 2);
    std::vector<lvk::Frame> packed(distinct);
    std::vector<lvk::VideoFrame420> planar(distinct);
    FILE* cf = std::fopen(clip_path, "rb");
    if (!cf) { std::printf("bench: cannot open %s\n", clip_path); return 1; }
    for (int i = 0; i < distinct; i++)
    {
        if (std::fread(p420.data(), 1, p420.size(), cf) != p420.size()) { std::printf("bench: short clip file\n"); return 1; }
        planar[i].upload(p420.data(), rows, cols, false, i);
        // the packed overload gets the same luma with nearest-neighbour chroma (only its throughput is looked at)
        const uint8_t* up = p420.data() + (size_t)rows * cols; const uint8_t* vp = up + (size_t)(rows / 2) * (cols / 2);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
            {
                uint8_t* p = &host[((size_t)y * cols + x) * 3];
                p[0] = p420[(size_t)y * cols + x]; p[1] = up[(size_t)(y / 2) * (cols / 2) + x / 2]; p[2] = vp[(size_t)(y / 2) * (cols / 2) + x / 2];
            }
        packed[i].upload(host.data(), rows, cols, lvk::VideoFrame::YUV, i);
        lvk::hip::shared_context()->check(lvk_hip_sync(lvk::hip::shared_context()->get()), "bench");
    }
    std::fclose(cf);
    lvk::hip::shared_context()->check(lvk_hip_sync(lvk::hip::shared_context()->get()), "bench");
    lvk::StabilizationFilterSettings st;
    st.detection_resolution = {480, 270}; st.acceptance_threshold = 3.0f; st.track_local_motions = false; st.detection_regions = {2, 1};
    st.max_feature_density = 0.12f; st.min_feature_density = 0.04f; st.accumulation_rate = 3.0f; st.corrective_limits = {0.05f, 0.05f};
    st.crop_to_stable_region = true;
    for (int mode = 0; mode < 3; mode++)
    {
        lvk::StabilizationFilter filter(st);
        filter.set_overlap(mode >= 1);
        lvk::Frame out; lvk::VideoFrame420 out420;
        auto step = [&](int i) {
            const int period = 2 * distinct - 2, k = i % period, idx = k < distinct ? k : period - k;     // forward, then backward: continuous motion
            if (mode == 2) filter.apply(planar[idx], out420);
            else filter.apply(packed[idx], out);                       // const-ref overload: the resident frame is shared, not consumed
        };
        for (int i = 0; i < 40; i++) step(i);
        filter.context()->check(lvk_hip_sync(filter.context()->get()), "bench");
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < count; i++) step(40 + i);
        filter.context()->check(lvk_hip_sync(filter.context()->get()), "bench");
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("facade bench %dx%d %s: %.0f frames/s (%.3f ms/frame)\n", cols, rows,
                    mode == 0 ? "packed" : (mode == 1 ? "packed+overlap" : "i420+overlap"), count / sec, 1e3 * sec / count);
    }
    return 0;
}

} // namespace
#endif

int main(int argc, char** argv)
{
#ifdef RUN_ON_GPU
    if (argc >= 7 && std::string(argv[1]) == "--golden") return run_golden(argv[2], std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), argv[6]);
    if (argc >= 4 && std::string(argv[1]) == "--resize") return run_resize(argv[2], argv[3]);
    if (argc >= 4 && std::string(argv[1]) == "--resize420") return run_resize420(argv[2], argv[3]);
    if (argc >= 7 && std::string(argv[1]) == "--bench") return run_bench(std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), argv[5], std::atoi(argv[6]));
    if (argc >= 3 && std::string(argv[1]) == "--threads-and-files") return (run_two_threads_check() != 0 || run_cross_context_420_check() != 0 || run_file_input_check(argv[2]) != 0) ? 1 : 0;
    if (run_chain_race_check() != 0) return 1;
    {
        // VideoFilter::stream (Filters/VideoFilter.cpp:62-209; CLI use Modules/VideoEditor/VideoProcessor.cpp:148-230): reader thread ->
        // filter thread -> callback, frames the filter holds back are skipped, callback returning true stops early
        struct Synthetic : cv::VideoCapture
        {
            int n, i = 0, rows = 360, cols = 640;
            std::vector<uint8_t> host;
            explicit Synthetic(int count) : n(count), host((size_t)rows * cols * 3) {}
            bool isOpened() const override { return true; }
            double get(int) const override { return 10.0 * i; }
            bool read(lvk::VideoFrame& f) override
            {
                if (i >= n) return false;
                for (int y = 0; y < rows; y++)
                    for (int x = 0; x < cols; x++)
                    {
                        uint8_t* p = &host[((size_t)y * cols + x) * 3];
                        const int xs = x + (i % 3), ys = y + (i % 2);
                        p[0] = (uint8_t)((((xs / 12) + (ys / 12)) % 2) ? 210 : 30 + (xs * 5 + ys * 11) % 29); p[1] = 120; p[2] = 140;
                    }
                i++;
                f.upload(host.data(), rows, cols, lvk::VideoFrame::YUV, 0);
                return true;
            }
        };
        lvk::StabilizationFilterSettings st;
        st.predictive_samples = 4;
        lvk::StabilizationFilter filter(st);
        Synthetic all(20);
        std::vector<uint64_t> stamps;
        filter.stream(all, [&](lvk::Frame& out) { stamps.push_back(out.timestamp); return false; });
        if (stamps.size() != 16) { std::printf("stream: emitted %zu frames, expected 16\n", stamps.size()); return 1; }
        for (size_t k = 0; k < stamps.size(); k++)
            if (stamps[k] != (uint64_t)(10.0 * (double)(k + 1) * 1.0e6)) { std::printf("stream: bad timestamp %zu\n", k); return 1; }
        filter.restart();
        Synthetic many(200);
        int seen = 0;
        filter.stream(many, [&](lvk::Frame&) { return ++seen == 5; });
        if (seen != 5 || many.i >= 200) { std::printf("stream: early termination failed (%d, %d)\n", seen, many.i); return 1; }
        std::printf("stream ok: 16 frames in order, stopped after %d of %d read\n", seen, many.i);
    }
    {
        // ScalingFilter (Filters/ScalingFilter.cpp:27-59; CLI construction FilterParser.tpp): upscale + sharpen, output aliases the input
        const int srows = 90, scols = 160;
        std::vector<uint8_t> small((size_t)srows * scols * 3), big((size_t)180 * 320 * 3);
        for (size_t k = 0; k < small.size(); k++) small[k] = (uint8_t)((k * 37 + (k / 480) * 11) % 251);
        lvk::ScalingFilter scaler(cv::Size(320, 180), 0.8f);
        lvk::Frame frame;
        frame.upload(small.data(), srows, scols, lvk::VideoFrame::YUV, 77);
        scaler.apply(std::move(frame), frame, true);
        if (frame.empty() || frame.cols != 320 || frame.rows != 180 || frame.timestamp != 77) { std::printf("scaling: bad output frame\n"); return 1; }
        frame.download(big.data());
        // corners: upscale copies the nearest source pixel on its border band, sharpen copies its border
        if (big[0] != small[0] || big[1] != small[1] || big[2] != small[2]) { std::printf("scaling: corner pixel changed\n"); return 1; }
        scaler.reconfigure([](lvk::ScalingFilterSettings& settings) { settings.output_size = cv::Size(160, 90); settings.sharpness = 0.0f; });
        lvk::Frame same;
        same.upload(small.data(), srows, scols, lvk::VideoFrame::YUV, 78);
        scaler.apply(same, same);
        if (same.cols != 160 || same.rows != 90) { std::printf("scaling: identity size failed\n"); return 1; }
        std::printf("scaling ok: %s %.3f ms\n", scaler.alias().c_str(), scaler.timings().average().milliseconds());
    }
    {
        // CompositeFilter (Filters/CompositeFilter.cpp:28-190; the CLI builds one from its filter list, VideoProcessor.cpp): stabilizer
        // (delay 2) followed by a 2x ScalingFilter; outputs appear once the stabilizer's delay is filled, carry the delayed timestamps and
        // the scaled size; a disabled stage is skipped; save_outputs keeps the intermediate frames
        lvk::StabilizationFilterSettings st;
        st.predictive_samples = 2;
        auto stab = std::make_shared<lvk::StabilizationFilter>(st);
        auto scale = std::make_shared<lvk::ScalingFilter>(cv::Size(1280, 720), 0.5f);
        lvk::CompositeFilterSettings cs; cs.save_outputs = true;
        lvk::CompositeFilter chain({stab, scale}, cs);
        if (chain.filter_count() != 2 || chain.filters(1) != scale) { std::printf("composite: bad chain\n"); return 1; }
        std::vector<uint8_t> img((size_t)360 * 640 * 3);
        int got = 0;
        for (int i = 0; i < 8; i++)
        {
            for (int y = 0; y < 360; y++)
                for (int x = 0; x < 640; x++)
                {
                    uint8_t* p = &img[((size_t)y * 640 + x) * 3];
                    const int xs = x + (i % 3), ys = y + (i % 2);
                    p[0] = (uint8_t)((((xs / 16) + (ys / 16)) % 2) ? 200 : 40 + (xs * 7 + ys * 13) % 23); p[1] = 128; p[2] = 128;
                }
            if (i == 6) chain.disable_filter(1);
            lvk::Frame frame;
            frame.upload(img.data(), 360, 640, lvk::VideoFrame::YUV, 500 + i);
            chain.apply(std::move(frame), frame);
            if (i < 2) { if (!frame.empty()) { std::printf("composite: output before the delay is filled\n"); return 1; } continue; }
            const int want_cols = i >= 6 ? 640 : 1280;
            if (frame.empty() || frame.cols != want_cols || frame.timestamp != (uint64_t)(500 + i - 2)) { std::printf("composite: bad frame %d\n", i); return 1; }
            if (i < 6 && (chain.outputs(0).cols != 640 || chain.outputs(1).cols != 1280)) { std::printf("composite: outputs not saved\n"); return 1; }
            got++;
        }
        chain.enable_all_filters();
        if (!chain.is_filter_enabled(1)) { std::printf("composite: enable_all_filters\n"); return 1; }
        std::printf("composite ok: %d frames\n", got);
    }
    {
        // Frames in pinned host memory (the OBS asynchronous path incl. FrameIngest's upload_planes / download_planes,
        // Interop/FrameIngest.cpp:415-474): apply(HostFrame420) with and without look-ahead and overlap emits the bytes of apply(VideoFrame420)
        const int hr = 360, hc = 640, count = 14;
        auto paint = [&](uint8_t* dst, int i) {
            for (int y = 0; y < hr; y++)
                for (int x = 0; x < hc; x++)
                {
                    const int xs = x + (i % 3) * 2, ys = y + (i % 2) * 2;
                    dst[(size_t)y * hc + x] = (uint8_t)((((xs / 16) + (ys / 16)) % 2) ? 200 : 40 + (xs * 7 + ys * 13) % 23);
                }
            for (int k = 0; k < hr * hc / 2; k++) dst[(size_t)hr * hc + k] = (uint8_t)(96 + (k * 5 + i) % 64);
        };
        lvk::StabilizationFilterSettings st;
        st.predictive_samples = 3;
        std::vector<std::vector<uint8_t>> want;
        std::vector<uint64_t> want_ts;
        {
            lvk::StabilizationFilter filter(st);
            std::vector<uint8_t> img((size_t)hr * hc * 3 / 2);
            for (int i = 0; i < count; i++)
            {
                paint(img.data(), i);
                lvk::VideoFrame420 in, out;
                in.upload(img.data(), hr, hc, false, 900 + i);
                filter.apply(in, out);
                if (out.empty()) continue;
                want.emplace_back(img.size()); out.download(want.back().data()); want_ts.push_back(out.timestamp);
            }
        }
        if (want.size() != (size_t)count - 3) { std::printf("host frames: device path emitted %zu\n", want.size()); return 1; }
        for (int variant = 0; variant < 3; variant++)                 // 0: synchronous, 1: look-ahead, 2: look-ahead + overlap, outputs kept
        {
            lvk::StabilizationFilter filter(st);
            if (variant == 2) filter.set_overlap(true);
            std::vector<lvk::HostFrame420> in(count), kept;
            for (int i = 0; i < count; i++) { in[i].create({hc, hr}, false); paint(in[i].y(), i); in[i].timestamp = 900 + i; }
            size_t got = 0;
            for (int i = 0; i < count; i++)
            {
                if (variant >= 1 && i == 0) filter.prefetch(in[0]);            // announced frames are pushed in the order announced: this one first
                if (variant >= 1 && i + 1 < count) filter.prefetch(in[i + 1]);
                lvk::HostFrame420 out;
                filter.apply(in[i], out, variant == 0);
                if (out.empty()) continue;
                if (variant == 2) { kept.push_back(out); got++; continue; }
                out.wait();
                if (out.timestamp != want_ts[got] || std::memcmp(out.y(), want[got].data(), out.bytes()) != 0)
                    { std::printf("host frames: variant %d frame %zu differs\n", variant, got); return 1; }
                got++;
            }
            if (variant == 2)
            {
                if (!kept.empty()) kept.back().wait();
                for (size_t k = 0; k < kept.size(); k++)
                    if (kept[k].timestamp != want_ts[k] || std::memcmp(kept[k].y(), want[k].data(), kept[k].bytes()) != 0)
                        { std::printf("host frames: kept frame %zu differs\n", k); return 1; }
            }
            if (got != want.size()) { std::printf("host frames: variant %d emitted %zu\n", variant, got); return 1; }
        }
        std::printf("host frames ok: %zu frames x 3 variants identical to the device path\n", want.size());
    }
    lvk::VSFilterLike vs;
    vs.configure(false, true, 0.05f, 0.05f, 5, true, false, true);
    const int rows = 360, cols = 640;
    std::vector<uint8_t> host((size_t)rows * cols * 3);
    int emitted = 0;
    for (int i = 0; i < 12; i++)
    {
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
            {
                const int xs = x + (i % 3), ys = y + (i % 2);
                uint8_t* p = &host[((size_t)y * cols + x) * 3];
                p[0] = (uint8_t)((((xs / 16) + (ys / 16)) % 2) ? 200 : 40 + (xs * 7 + ys * 13) % 23);
                p[1] = 128; p[2] = 128;
            }
        lvk::Frame frame;
        frame.upload(host.data(), rows, cols, lvk::VideoFrame::YUV, 1000 + i);
        vs.filter(frame);
        if (!frame.empty())
        {
            if (frame.timestamp != (uint64_t)(1000 + i - 5)) { std::printf("bad timestamp\n"); return 1; }
            frame.download(host.data());
            emitted++;
        }
    }
    std::printf("emitted %d frames\n", emitted);
    return emitted == 7 ? 0 : 1;
#else
    (void)argc; (void)argv;
    // CLI-style construction (FilterParser.tpp:51-64)
    std::printf("conformance TU compiled; sizeof(StabilizationFilterSettings) = %zu\n", sizeof(lvk::StabilizationFilterSettings));
    if (false)
    {
        auto filter = std::make_shared<lvk::StabilizationFilter>();
        std::static_pointer_cast<lvk::Configurable<lvk::StabilizationFilterSettings>>(filter)->configure(lvk::StabilizationFilterSettings{});
        std::printf("%s %f\n", filter->alias().c_str(), filter->timings().average().milliseconds());
        auto scaler = std::make_shared<lvk::ScalingFilter>();
        std::static_pointer_cast<lvk::Configurable<lvk::ScalingFilterSettings>>(scaler)->configure(lvk::ScalingFilterSettings{});
        lvk::Frame a, b;
        lvk::upscale(a, b, cv::Size(1920, 1080));
        lvk::sharpen(b, b);
    }
    return 0;
#endif
}
