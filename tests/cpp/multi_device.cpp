// One process, several GPUs: SURVEY.md section 8e's partitioning ("stream i -> device i mod 8; one host thread + one lvk_hip_ctx per device",
// no collective) through the C++ facade, the way a host that serves several sources would use it.  The reference's contract is per-instance
// re-entrancy from whatever thread the host provides (LiveVisionKit/Functions/Image.cpp:39-45 thread_local kernels;
// Modules/OBS-Plugin/Interop/VisionFilter.cpp:157-162: the filter is driven from OBS's video thread, whichever that is).
//
//   phase A  a thread that has never called hipSetDevice drives a filter on the LAST visible device (the C-ABI makes its context's device
//            current per call and restores the caller's) -- bytes equal to the same clip on device 0 from the main thread;
//   phase B  one thread + one hip::Context + one filter per visible device, all at once, a different clip each, free running with the
//            overlap on -- every stream's bytes equal to its single-threaded run on device 0;
//   phase C  ONE thread drives filters on the first and the last device alternately, call by call (the calling thread's current device
//            must come back unchanged after every call).
// With one visible device the same phases run on device 0 and the program says so: "SKIP (1 device visible) ...".
#include <lvk/LiveVisionKit.hpp>

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

extern "C" int hipGetDevice(int* device);          // (libamdhip64, which the library links: the test only reads / sets the thread's current device)
extern "C" int hipSetDevice(int device);

namespace {

constexpr int ROWS = 360, COLS = 640, N = 36;
using Outputs = std::vector<std::vector<uint8_t>>;

void paint(std::vector<uint8_t>& img, int stream, int i)
{
    for (int y = 0; y < ROWS; y++)
        for (int x = 0; x < COLS; x++)
        {
            uint8_t* p = &img[((size_t)y * COLS + x) * 3];
            const int cell = 11 + 2 * (stream % 5), xs = x + (2 + stream % 3) * (i % 4) + 13 * stream, ys = y + (i % 3) + 7 * stream;
            p[0] = (uint8_t)((((xs / cell) + (ys / cell)) % 2) ? 205 - 9 * (stream % 8) : 30 + (xs * 5 + ys * (9 + stream)) % 37);
            p[1] = (uint8_t)(100 + (x >> 3) + 5 * stream); p[2] = (uint8_t)(90 + (y >> 2));
        }
}

lvk::StabilizationFilterSettings settings_of(int stream)
{
    lvk::StabilizationFilterSettings st;
    st.predictive_samples = 3 + (stream & 1); st.detection_resolution = {480, 270};
    st.min_scene_quality = 0.4f; st.min_tracking_quality = 0.2f;                      // the trust factor leaves zero: the warp depends on the tracker
    if ((stream & 1) == 0) { st.track_local_motions = false; st.motion_resolution = {2, 2}; st.detection_regions = {2, 1}; st.acceptance_threshold = 3.0f; }
    else { st.track_local_motions = true; st.motion_resolution = {16, 16}; st.detection_regions = {2, 2}; st.acceptance_threshold = 10.0f; }
    return st;
}

// clip `stream` through a filter on `device`, driven by the calling thread; returns false when the thread's current device changed under it
bool run_stream(int stream, int device, Outputs& outs)
{
    int before = -1; hipGetDevice(&before);
    auto ctx = std::make_shared<lvk::hip::Context>(device);
    lvk::StabilizationFilter filter(lvk::StabilizationFilterSettings{}, ctx, device);
    filter.configure(settings_of(stream));
    filter.set_overlap(true);
    std::vector<uint8_t> img((size_t)ROWS * COLS * 3);
    std::vector<lvk::Frame> kept;
    bool device_kept = true;
    for (int i = 0; i < N; i++)
    {
        paint(img, stream, i);
        lvk::Frame frame;
        frame.upload(img.data(), ROWS, COLS, lvk::VideoFrame::YUV, 1000 * stream + i, ctx);
        filter.apply(std::move(frame), frame);                     // free running: nothing waits for the GPU inside the loop
        if (!frame.empty()) kept.push_back(std::move(frame));
        int now = -2; hipGetDevice(&now);
        device_kept = device_kept && now == before;
    }
    for (auto& f : kept) { outs.emplace_back((size_t)ROWS * COLS * 3); f.download(outs.back().data()); }
    return device_kept;
}

bool same(const Outputs& a, const Outputs& b, const char* what, int stream)
{
    if (a.size() != b.size() || a.empty()) { std::printf("multi-device: %s, stream %d: %zu frames against %zu\n", what, stream, a.size(), b.size()); return false; }
    for (size_t k = 0; k < a.size(); k++)
        if (a[k] != b[k]) { std::printf("multi-device: %s, stream %d: frame %zu differs from the device-0 single-thread run\n", what, stream, k); return false; }
    return true;
}

} // namespace

int main()
{
    const int ndev = lvk_hip_device_count();
    if (ndev <= 0) { std::printf("multi-device: no gfx950 device visible (%s)\n", lvk_hip_last_error(nullptr)); return 2; }
    if (lvk_hip_abi_version() != LVK_HIP_ABI_VERSION) { std::printf("multi-device: header ABI %d, library ABI %d\n", LVK_HIP_ABI_VERSION, lvk_hip_abi_version()); return 1; }
    const int nstreams = ndev > 2 ? ndev : 2;
    const int last = ndev - 1;

    // the reference runs: every clip on device 0, from this thread
    std::vector<Outputs> want((size_t)nstreams);
    for (int s = 0; s < nstreams; s++)
        if (!run_stream(s, 0, want[(size_t)s])) { std::printf("multi-device: the calling thread's device changed (reference run %d)\n", s); return 1; }
    if (want[0].size() != (size_t)N - 3 || want[1].size() != (size_t)N - 4 || want[0][12] == want[1][12]) { std::printf("multi-device: reference runs implausible\n"); return 1; }

    // ---- phase A: a fresh thread (no hipSetDevice, current device 0 by default) drives the LAST device
    {
        Outputs got; bool kept = false;
        std::thread t([&] { kept = run_stream(1, last, got); });
        t.join();
        if (!kept) { std::printf("multi-device: phase A: the driving thread's current device was changed by a call\n"); return 1; }
        if (!same(got, want[1], "phase A (device of the context != device of the thread)", 1)) return 1;
    }
    // ---- phase B: one thread + one context + one filter per device, all at once
    {
        const int nthreads = ndev > 1 ? ndev : 2;                   // (one device: two threads on it, so that the phase still runs concurrently)
        std::vector<Outputs> got((size_t)nthreads); std::vector<char> kept((size_t)nthreads, 0);
        std::vector<std::thread> threads;
        for (int d = 0; d < nthreads; d++) threads.emplace_back([&, d] { kept[(size_t)d] = run_stream(d, d % ndev, got[(size_t)d]) ? 1 : 0; });
        for (auto& t : threads) t.join();
        for (int d = 0; d < nthreads; d++)
        {
            if (!kept[(size_t)d]) { std::printf("multi-device: phase B: thread %d's current device was changed by a call\n", d); return 1; }
            if (!same(got[(size_t)d], want[(size_t)d], "phase B (one thread per device)", d)) return 1;
        }
    }
    // ---- phase C: one thread, two filters on the first and the last device, alternating call by call; the thread sits on device 0 throughout
    {
        hipSetDevice(0);
        auto c0 = std::make_shared<lvk::hip::Context>(0); auto c1 = std::make_shared<lvk::hip::Context>(last);
        lvk::StabilizationFilter f0(lvk::StabilizationFilterSettings{}, c0, 0), f1(lvk::StabilizationFilterSettings{}, c1, last);
        f0.configure(settings_of(0)); f1.configure(settings_of(1)); f0.set_overlap(true); f1.set_overlap(true);
        std::vector<uint8_t> img((size_t)ROWS * COLS * 3);
        std::vector<lvk::Frame> k0, k1;
        for (int i = 0; i < N; i++)
        {
            for (int s = 0; s < 2; s++)
            {
                paint(img, s, i);
                lvk::Frame frame;
                frame.upload(img.data(), ROWS, COLS, lvk::VideoFrame::YUV, 1000 * s + i, s ? c1 : c0);
                (s ? f1 : f0).apply(std::move(frame), frame);
                if (!frame.empty()) (s ? k1 : k0).push_back(std::move(frame));
                int now = -1; hipGetDevice(&now);
                if (now != 0) { std::printf("multi-device: phase C: current device %d after a call on a context of device %d\n", now, s ? last : 0); return 1; }
            }
        }
        Outputs g0, g1;
        for (auto& f : k0) { g0.emplace_back((size_t)ROWS * COLS * 3); f.download(g0.back().data()); }
        for (auto& f : k1) { g1.emplace_back((size_t)ROWS * COLS * 3); f.download(g1.back().data()); }
        if (!same(g0, want[0], "phase C (one thread, alternating devices)", 0) || !same(g1, want[1], "phase C (one thread, alternating devices)", 1)) return 1;
    }
    if (ndev == 1)
        std::printf("SKIP (1 device visible): the cross-device halves of phases A-C ran on device 0 only; on an N-GPU node every phase spans devices 0 .. N-1\n");
    std::printf("multi-device ok: %d device(s), %d stream(s): foreign-thread drive, one thread per device, one thread across devices == device-0 single-thread runs\n", ndev, nstreams);
    return 0;
}
