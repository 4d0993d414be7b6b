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

int main(int argc, char** argv)
{
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
