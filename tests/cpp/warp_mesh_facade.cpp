// lvk::Homography / lvk::WarpMesh / lvk::remap of the C++ facade (include/lvk/WarpMesh.hpp) driven the way the reference's callers drive them
// (LCFilter.cpp:133-192: set_to(map) -> crop_in -> apply; StabilizationFilter / PathSmoother: set_to(H), clamp, combine, the operators).
//   warp_mesh_facade                       host arithmetic only: prints named arrays as hexadecimal floats (the Python side recomputes them in numpy)
//   warp_mesh_facade --gpu <in> <out>      in: int32 rows, cols + packed YUV frame; out: the frames apply() / remap() produce, back to back
#include <lvk/LiveVisionKit.hpp>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

const double kH[9] = {1.01, 0.012, 3.1, -0.011, 0.995, -2.2, 2e-5, -1e-5, 1.0};

void dump(const char* name, const float* v, size_t n)
{
    std::printf("%s", name);
    for (size_t i = 0; i < n; i++) std::printf(" %a", (double)v[i]);
    std::printf("\n");
}
void dump(const char* name, const double* v, size_t n)
{
    std::printf("%s", name);
    for (size_t i = 0; i < n; i++) std::printf(" %a", v[i]);
    std::printf("\n");
}
void dump(const char* name, const lvk::WarpMesh& m) { dump(name, m.offsets(), (size_t)m.rows() * m.cols() * 2); }

// the smooth pixel map of the full-size case: identity + offsets of a few pixels (what initUndistortRectifyMap hands LCFilter, in spirit)
std::vector<float> pixel_map(int rows, int cols)
{
    std::vector<float> map((size_t)rows * cols * 2);
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++)
        {
            const float u = (float)c / (float)cols - 0.5f, v = (float)r / (float)rows - 0.5f;
            map[((size_t)r * cols + c) * 2] = (float)c + 6.0f * u * (u * u + v * v);
            map[((size_t)r * cols + c) * 2 + 1] = (float)r + 4.0f * v * (u * u + v * v);
        }
    return map;
}

int host_part()
{
    const cv::Size size(5, 4);
    const lvk::Homography H(kH);
    lvk::WarpMesh A(H, cv::Size2f(480.0f, 270.0f), size);
    dump("set_to_H", A);
    A.crop_in(cv::Rect2f(0.05f, 0.04f, 0.9f, 0.92f));
    dump("crop_in", A);
    A.clamp(cv::Size2f(0.06f, 0.05f));
    dump("clamp", A);
    lvk::WarpMesh B(size);
    B.write([](cv::Point2f& o, const cv::Point& p) { o.x = 0.001f * (float)p.x - 0.002f * (float)p.y; o.y = 0.0005f * (float)(p.x * p.y); });
    dump("write", B);
    A.combine(B, 0.35f);
    dump("combine", A);
    A += B; A -= cv::Point2f(0.01f, -0.02f); A *= 0.7f; A *= cv::Size2f(1.5f, 0.5f); A /= cv::Size2f(3.0f, 7.0f); A /= 1.3f; A -= B; A += cv::Point2f(0.25f, 0.125f);
    dump("operators", A);
    A.scale(cv::Size2f(0.9f, 0.8f));
    dump("scale", A);
    A.clamp(cv::Size2f(-0.1f, 0.0f), cv::Size2f(0.3f, 0.2f));
    dump("clamp2", A);
    lvk::WarpMesh P(size);
    P.set_to(cv::Point2f(0.03f, -0.01f));
    P *= B;
    dump("set_to_point_times", P);
    float sum[2] = {0.0f, 0.0f};
    A.read([&](const cv::Point2f& o, const cv::Point& p) { sum[0] += o.x * (float)(p.x + 1); sum[1] += o.y * (float)(p.y + 1); }, false);
    dump("read", sum, 2);
    const auto map = pixel_map(4, 5);
    lvk::WarpMesh M(map.data(), size, false, false);
    dump("set_to_map", M);
    std::vector<float> back;
    M.to_map(back);
    dump("to_map", back.data(), back.size());
    // Homography
    const lvk::Homography inv = H.invert();
    dump("H_invert", inv.data(), 9);
    lvk::Homography prod = H; prod *= inv;
    dump("H_product", prod.data(), 9);
    lvk::Homography comb = (H + inv) * 0.5 - lvk::Homography::Identity() / 4.0;
    dump("H_ops", comb.data(), 9);
    const double aff[6] = {0.99, -0.02, 4.0, 0.02, 0.99, -3.0};
    const lvk::Homography A2 = lvk::Homography::FromAffineMatrix(aff);
    std::printf("flags %d %d %d %d %d\n", (int)A2.is_affine(), (int)H.is_affine(), (int)lvk::Homography().is_identity(), (int)lvk::Homography::Zero().is_zero(), (int)inv.is_zero());
    const cv::Point2f pf = H * cv::Point2f(123.25f, 77.5f);
    const cv::Point2d pd = H * cv::Point2d(123.25, 77.5);
    const float pfv[2] = {pf.x, pf.y}; const double pdv[2] = {pd.x, pd.y};
    dump("transform_f", pfv, 2); dump("transform_d", pdv, 2);
    std::printf("host part done\n");
    return 0;
}

int gpu_part(const char* in_path, const char* out_path)
{
    FILE* f = std::fopen(in_path, "rb");
    if (!f) return 1;
    int32_t head[2];
    if (std::fread(head, sizeof(int32_t), 2, f) != 2) return 1;
    const int rows = head[0], cols = head[1];
    std::vector<uint8_t> px((size_t)rows * cols * 3), host(px.size());
    if (std::fread(px.data(), 1, px.size(), f) != px.size()) return 1;
    std::fclose(f);
    FILE* out = std::fopen(out_path, "wb");
    if (!out) return 1;
    auto emit = [&](const lvk::VideoFrame& fr, uint64_t want_ts) -> bool {
        if (fr.empty() || fr.rows != rows || fr.cols != cols || fr.timestamp != want_ts || fr.format != lvk::VideoFrame::YUV) return false;
        fr.download(host.data());
        std::fwrite(host.data(), 1, host.size(), out);
        return true;
    };
    const cv::Scalar bg(105, 212, 235);
    lvk::VideoFrame src;
    src.upload(px.data(), rows, cols, lvk::VideoFrame::YUV, 4242);
    // (a) 2 x 2 mesh: the stabilizer's homography preset
    lvk::WarpMesh m2(lvk::Homography(kH), cv::Size2f((float)cols, (float)rows));
    lvk::VideoFrame dst;
    m2.apply(src, dst, bg);
    if (!emit(dst, 4242)) { std::printf("apply 2x2: bad frame\n"); return 1; }
    // (b) 16 x 16 mesh: the vector-field preset; output aliases the input, as LCFilter / VSFilter do
    lvk::WarpMesh m16(cv::Size(16, 16));
    m16.write([](cv::Point2f& o, const cv::Point& p) {          // (integer patterns: no libm in the expected values)
        o.x = 0.008f * ((float)((p.x * 7 + p.y * 3) % 11) / 11.0f - 0.5f);
        o.y = 0.006f * ((float)((p.x * 5 + p.y * 9) % 13) / 13.0f - 0.5f);
    });
    lvk::VideoFrame alias = src.clone();
    alias.timestamp = 77;
    m16.apply(alias, alias, bg);
    if (!emit(alias, 77)) { std::printf("apply 16x16: bad frame\n"); return 1; }
    // (c) a mesh of the frame's own size: LCFilter's flow -- set_to(pixel map, not offsets, not normalised) -> crop_in(view region) -> apply
    const auto map = pixel_map(rows, cols);
    lvk::WarpMesh full(map.data(), cv::Size(cols, rows), false, false);
    full.crop_in(cv::Rect2f(0.02f, 0.03f, 0.95f, 0.94f));
    lvk::VideoFrame corrected;
    full.apply(src, corrected, bg);
    if (!emit(corrected, 4242)) { std::printf("apply full-size: bad frame\n"); return 1; }
    {
        // the device map is cached while the mesh does not change: a second apply gives the same bytes, a changed mesh different ones
        lvk::VideoFrame again, moved;
        full.apply(src, again, bg);
        std::vector<uint8_t> a(px.size()), b(px.size()), c(px.size());
        corrected.download(a.data()); again.download(b.data());
        full += cv::Point2f(0.01f, 0.0f);
        full.apply(src, moved, bg);
        moved.download(c.data());
        if (a != b || a == c) { std::printf("apply full-size: the cached map is stale or not reused\n"); return 1; }
    }
    // (d) lvk::remap(homography): dst -> src given (inverted), and src -> dst given (inverted = false: the launcher inverts)
    lvk::VideoFrame r1, r2;
    r1.timestamp = 5; r1.format = lvk::VideoFrame::YUV; r2.timestamp = 6; r2.format = lvk::VideoFrame::YUV;
    lvk::remap(src, r1, lvk::Homography(kH), bg, true);
    lvk::remap(src, r2, lvk::Homography(kH), bg, false);
    if (!emit(r1, 5) || !emit(r2, 6)) { std::printf("remap(homography): bad frame\n"); return 1; }
    // (e) lvk::remap(offset map): a device-resident pixel-offset map
    std::vector<float> offs(map.size());
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) { const size_t i = ((size_t)r * cols + c) * 2; offs[i] = map[i] - (float)c; offs[i + 1] = map[i + 1] - (float)r; }
    lvk::OffsetMap dmap;
    dmap.upload(offs.data(), cv::Size(cols, rows), src.context());
    lvk::VideoFrame r3; r3.timestamp = 9; r3.format = lvk::VideoFrame::YUV;
    lvk::remap(src, r3, dmap, bg);
    if (!emit(r3, 9)) { std::printf("remap(map): bad frame\n"); return 1; }
    // (f) the overlay launchers on a copy of the frame: LCFilter's test grid (LCFilter.cpp:177, col::MAGENTA[format]) + crosses at scaled points
    lvk::VideoFrame drawn = src.clone();
    drawn.timestamp = 11;
    lvk::draw_grid(drawn, cv::Size(8, 5), lvk::col::MAGENTA[drawn.format], 1);
    const std::vector<cv::Point2f> pts = {{10.5f, 7.25f}, {100.0f, 60.0f}, {239.6f, 134.4f}, {0.0f, 0.0f}, {130.2f, 20.9f}};
    lvk::draw_crosses(drawn, pts, lvk::yuv::GREEN, 7, 4, cv::Size2f(2.0f, 2.0f));
    if (!emit(drawn, 11)) { std::printf("draw: bad frame\n"); return 1; }
    std::fclose(out);
    std::printf("gpu part done: 7 frames\n");
    return 0;
}

} // namespace

int main(int argc, char** argv)
{
    if (argc >= 4 && std::string(argv[1]) == "--gpu") return gpu_part(argv[2], argv[3]);
    return host_part();
}
