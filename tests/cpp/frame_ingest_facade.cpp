// lvk::FrameIngest of the C++ facade (include/lvk/FrameIngest.hpp) driven the way the plugin drives its Interop/FrameIngest
// (Modules/OBS-Plugin/Interop/VisionFilter.cpp:232-253: Select by frame->format, upload_obs_frame, filter, download_ocl_frame), with a stand-in for
// obs_source_frame.  usage: frame_ingest_facade <obs format> <rows> <cols> <pad> <planes.bin> <out prefix>
//   planes.bin = the tight planes one after the other; they are laid out with `pad` extra bytes per row (linesize > width) before the upload;
//   writes <prefix>.frame (the VideoFrame's bytes) and <prefix>.planes (the planes download_ocl_frame filled, tight, from buffers that held 0x5A).
#include <lvk/FrameIngest.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct fake_obs_source_frame            // the members of libobs' obs_source_frame the plugin's FrameIngest touches
{
    uint8_t* data[8] = {};
    uint32_t linesize[8] = {};
    uint32_t width = 0, height = 0;
    uint64_t timestamp = 0;
    int format = 0;
};

static int plane_table(int fmt, int rows, int cols, int prow[3], int pbytes[3], int pwritten[3])
{
    auto set = [&](int i, int r, int w) { prow[i] = r; pbytes[i] = w; pwritten[i] = w; };
    switch (fmt)
    {
    case 1: case 13: set(0, rows, cols); set(1, rows / 2, cols / 2); set(2, rows / 2, cols / 2); return 3;
    case 2: set(0, rows, cols); set(1, rows / 2, cols); return 2;
    case 12: case 14: set(0, rows, cols); set(1, rows, cols / 2); set(2, rows, cols / 2); return 3;
    case 10: case 15: set(0, rows, cols); set(1, rows, cols); set(2, rows, cols); return 3;
    case 3: case 4: case 5: set(0, rows, 2 * cols); return 1;
    case 16: case 6: case 7: case 8: set(0, rows, 4 * cols); return 1;
    case 11: set(0, rows, 3 * cols); return 1;
    }
    return 0;
}

// The plugin's asynchronous path with the facade's classes (Modules/OBS-Plugin/Interop/VisionFilter.cpp:151-253 around VSFilter.cpp:352-364):
//   ingest->upload_obs_frame(obs frame, frame); filter.apply(std::move(frame), frame); if (!frame.empty()) ingest->download_ocl_frame(frame, obs frame)
// usage: frame_ingest_facade --stream <obs format> <rows> <cols> <n frames> <delay> <planes.bin> <out.bin>     (planes.bin: n frames' tight planes; out.bin: the emitted ones)
static int run_stream(int argc, char** argv)
{
    if (argc < 9) return 2;
    const int fmt = std::atoi(argv[2]), rows = std::atoi(argv[3]), cols = std::atoi(argv[4]), n = std::atoi(argv[5]), delay = std::atoi(argv[6]);
    int prow[3], pbytes[3], pwritten[3];
    const int np = plane_table(fmt, rows, cols, prow, pbytes, pwritten);
    size_t frame_bytes = 0;
    for (int i = 0; i < np; i++) frame_bytes += (size_t)prow[i] * pbytes[i];
    std::vector<uint8_t> clip(frame_bytes * n), back(frame_bytes);
    FILE* f = std::fopen(argv[7], "rb");
    if (!f || std::fread(clip.data(), 1, clip.size(), f) != clip.size()) return 2;
    std::fclose(f);
    FILE* out = std::fopen(argv[8], "wb");
    if (!out) return 2;
    auto ingest = lvk::FrameIngest::Select(fmt);
    if (!ingest) return 1;
    lvk::StabilizationFilter filter;                         // the plugin's order: a default-constructed filter, then the preset (VSFilter.cpp:235-294)
    filter.reconfigure([&](lvk::StabilizationFilterSettings& s) {
        s.detection_resolution = {480, 270}; s.detection_regions = {2, 1}; s.motion_resolution = {2, 2};
        s.acceptance_threshold = 3.0f; s.track_local_motions = false;
        s.max_feature_density = 0.12f; s.min_feature_density = 0.04f; s.accumulation_rate = 3.0f;
        s.corrective_limits = {0.05f, 0.05f}; s.crop_to_stable_region = true; s.background_colour = {105, 212, 235};
        s.predictive_samples = (size_t)delay; s.min_scene_quality = 0.3f; s.min_tracking_quality = 0.2f;
    });
    lvk::Frame frame;
    int emitted = 0;
    for (int k = 0; k < n; k++)
    {
        fake_obs_source_frame obs;
        obs.width = cols; obs.height = rows; obs.format = fmt; obs.timestamp = 500 + k;
        uint8_t* p = clip.data() + frame_bytes * k;
        for (int i = 0; i < np; i++) { obs.data[i] = p; obs.linesize[i] = pbytes[i]; p += (size_t)prow[i] * pbytes[i]; }
        ingest->upload_obs_frame(&obs, frame);
        filter.apply(std::move(frame), frame);
        if (frame.empty()) continue;
        if (frame.timestamp != (uint64_t)(500 + k - delay)) { std::fprintf(stderr, "timestamp\n"); return 1; }
        // the plugin downloads into the OBS frame it re-associates by timestamp (VisionFilter.cpp:232-253): here a scratch frame of the same layout
        fake_obs_source_frame dst;
        dst.width = cols; dst.height = rows; dst.format = fmt;
        std::fill(back.begin(), back.end(), 0x5A);
        p = back.data();
        for (int i = 0; i < np; i++) { dst.data[i] = p; dst.linesize[i] = pbytes[i]; p += (size_t)prow[i] * pbytes[i]; }
        ingest->download_ocl_frame(frame, &dst);
        if (dst.timestamp != frame.timestamp) return 1;
        std::fwrite(back.data(), 1, back.size(), out);
        emitted++;
    }
    std::fclose(out);
    std::printf("stream ok: %d frames\n", emitted);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::string(argv[1]) == "--stream") return run_stream(argc, argv);
    if (argc < 7) { std::fprintf(stderr, "usage\n"); return 2; }
    const int fmt = std::atoi(argv[1]), rows = std::atoi(argv[2]), cols = std::atoi(argv[3]);
    int pad = std::atoi(argv[4]);
    if (fmt == 6 || fmt == 7 || fmt == 8) pad = 0;                       // DirectIngest's 4-byte formats are a tight byte stream by definition
    if (lvk::FrameIngest::Select(9) != nullptr || lvk::FrameIngest::Select(0) != nullptr || lvk::FrameIngest::Select(17) != nullptr) { std::fprintf(stderr, "Select accepted a format it cannot take\n"); return 1; }
    auto ingest = lvk::FrameIngest::Select(fmt);
    if (!ingest) { std::fprintf(stderr, "Select(%d) = null\n", fmt); return 1; }
    if (ingest->obs_format() != fmt) return 1;
    int prow[3], pbytes[3], pwritten[3];
    const int n = plane_table(fmt, rows, cols, prow, pbytes, pwritten);
    std::vector<std::vector<uint8_t>> tight(n), padded(n);
    FILE* f = std::fopen(argv[5], "rb");
    if (!f) return 2;
    fake_obs_source_frame in;
    in.width = cols; in.height = rows; in.timestamp = 0x1234567890ull; in.format = fmt;
    for (int i = 0; i < n; i++)
    {
        tight[i].resize((size_t)prow[i] * pbytes[i]);
        if (std::fread(tight[i].data(), 1, tight[i].size(), f) != tight[i].size()) return 2;
        const int step = pbytes[i] + pad;
        padded[i].assign((size_t)prow[i] * step, 0xEE);
        for (int r = 0; r < prow[i]; r++) std::memcpy(padded[i].data() + (size_t)r * step, tight[i].data() + (size_t)r * pbytes[i], pbytes[i]);
        in.data[i] = padded[i].data(); in.linesize[i] = step;
    }
    std::fclose(f);
    if (!lvk::FrameIngest::test_obs_frame(&in)) return 1;

    lvk::VideoFrame frame;
    ingest->upload_obs_frame(&in, frame);
    for (int i = 0; i < n; i++) std::fill(padded[i].begin(), padded[i].end(), 0);          // the planes are the caller's again: wiping them must not matter
    if (frame.timestamp != in.timestamp || frame.format != ingest->ocl_format() || frame.cols != cols || frame.rows != rows) { std::fprintf(stderr, "metadata\n"); return 1; }
    std::vector<uint8_t> host((size_t)rows * cols * 3);
    frame.download(host.data());
    std::string prefix = argv[6];
    f = std::fopen((prefix + ".frame").c_str(), "wb"); std::fwrite(host.data(), 1, host.size(), f); std::fclose(f);

    // and back, into planes with the same pitch that hold 0x5A: bytes the reference does not write keep it
    fake_obs_source_frame out;
    out.width = cols; out.height = rows; out.format = fmt;
    std::vector<std::vector<uint8_t>> back(n);
    for (int i = 0; i < n; i++)
    {
        const int step = pbytes[i] + pad;
        back[i].assign((size_t)prow[i] * step, 0x5A);
        out.data[i] = back[i].data(); out.linesize[i] = step;
    }
    frame.timestamp = 77;
    ingest->download_ocl_frame(frame, &out);
    if (out.timestamp != 77) return 1;
    f = std::fopen((prefix + ".planes").c_str(), "wb");
    for (int i = 0; i < n; i++)
    {
        const int step = pbytes[i] + pad;
        for (int r = 0; r < prow[i]; r++)
        {
            std::fwrite(back[i].data() + (size_t)r * step, 1, pbytes[i], f);
            for (int k = pbytes[i]; k < step; k++) if (back[i][(size_t)r * step + k] != 0x5A) { std::fprintf(stderr, "row padding of plane %d written\n", i); return 1; }
        }
    }
    std::fclose(f);
    std::printf("ok\n");
    return 0;
}
