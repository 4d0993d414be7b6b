// Sanitizer run of the CPU oracle ("-fsanitize build option"): compiled TOGETHER with oracle/*.cpp under
// -fsanitize=address,undefined (tests/test_sanitize.py; `make -C oracle SANITIZE=1` builds the shared library the same way) and driven
// through every stage on a small synthetic stream: the stateful filter with both presets (scene cut, reconfigure, restart, lens, overlays),
// the frozen robust estimator and the reference-semantics USAC leg, the mesh solver beyond the preset's size, 4:2:0 conversions on odd
// pitches, EASU upscale / RCAS, the lens map.  Prints "oracle sanitize ok"; any ASan / UBSan report aborts (-fno-sanitize-recover).
#include "lvk_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static uint32_t g_seed = 12345u;
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return (float)(g_seed >> 8) / 16777216.0f; }

static void render(std::vector<uint8_t>& f, int rows, int cols, int i, int scene)
{
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++)
        {
            const int xs = x + (i * 3) % 7 + scene * 37, ys = y + (i * 2) % 5 + scene * 11;
            uint8_t* p = &f[((size_t)y * cols + x) * 3];
            p[0] = (uint8_t)((((xs / 9) + (ys / 7)) % 2) ? 200 - (xs * 3 + ys) % 40 : 40 + (xs * 5 + ys * 11) % 31);
            p[1] = (uint8_t)(100 + (xs / 4) % 50); p[2] = (uint8_t)(150 - (ys / 3) % 60);
        }
}

int main()
{
    const int rows = 186, cols = 322;          // not multiples of the box sizes: the fractional INTER_AREA path
    std::vector<uint8_t> frame((size_t)rows * cols * 3), out(frame.size());
    for (int preset = 0; preset < 2; preset++)
    {
        lvko_stab_settings s; lvko_stab_default_settings(&s);
        lvko_stab* st = lvko_stab_create(&s);
        s.detection_width = 160; s.detection_height = 92; s.predictive_samples = 3; s.min_motion_samples = 20;
        s.detection_regions_x = 2; s.detection_regions_y = preset ? 2 : 1;
        s.track_local_motions = preset; s.motion_width = preset ? 9 : 2; s.motion_height = preset ? 7 : 2; s.acceptance_threshold = preset ? 10.0f : 3.0f;
        s.crop_to_stable_region = 1; s.corrective_limit_x = s.corrective_limit_y = 0.05f;
        lvko_stab_configure(st, &s);
        int emitted = 0;
        for (int i = 0; i < 22; i++)
        {
            render(frame, rows, cols, i, i >= 12);
            if (i == 8) { lvko_stab_settings t = s; t.predictive_samples = 2; lvko_stab_configure(st, &t); }
            if (i == 16) lvko_stab_restart(st);
            if (i == 18) { const double lens[9] = {0.8 * cols, 0.8 * cols, cols / 2.0, rows / 2.0, -0.12, 0.03, 0, 0, 0}; lvko_stab_set_lens(st, lens); }
            uint64_t ts = 0;
            emitted += lvko_stab_push(st, frame.data(), cols * 3, rows, cols, (uint64_t)i, out.data(), cols * 3, &ts, 3) == 1;
            if (i % 5 == 0) { lvko_stab_draw_trackers(st); lvko_stab_draw_motion_mesh(st); }
            lvko_stab_stats stats; lvko_stab_get_stats(st, &stats);
            std::vector<float> a(2 * 9 * 7), b(2 * 9 * 7), feat(4 * 4096), p1(2 * 4096), p2(2 * 4096); int est = 0;
            lvko_stab_get_meshes(st, a.data(), b.data(), (int)a.size());
            lvko_stab_get_features(st, feat.data(), 4096);
            lvko_stab_get_matches(st, p1.data(), p2.data(), 4096, &est);
        }
        if (emitted < 8) { std::printf("too few frames emitted (%d)\n", emitted); return 1; }
        lvko_stab_destroy(st);
    }
    // robust estimators: sizes around the partial-sum tree (256) and the minimal sample, with outliers
    for (int n : {0, 3, 4, 5, 255, 256, 257, 700})
    {
        std::vector<float> p1(2 * (size_t)n + 2), p2(2 * (size_t)n + 2); std::vector<uint8_t> mask((size_t)n + 1);
        for (int i = 0; i < n; i++)
        {
            p1[2 * i] = frand() * 480; p1[2 * i + 1] = frand() * 270;
            p2[2 * i] = 1.01f * p1[2 * i] - 0.01f * p1[2 * i + 1] + 2 + (frand() - 0.5f) * 0.2f + (i % 5 == 0 ? frand() * 60 : 0);
            p2[2 * i + 1] = 0.01f * p1[2 * i] + 1.01f * p1[2 * i + 1] - 1 + (frand() - 0.5f) * 0.2f;
        }
        double H[9]; int iters = 0;
        lvko_find_homography(p1.data(), p2.data(), n, 3.0, 480, 270, H, mask.data());
        lvko_estimate_affine_partial(p1.data(), p2.data(), n, 3.0, 480, 270, H, mask.data());
        lvko_usac_find_homography(p1.data(), p2.data(), n, 3.0, 3.0, 0, 1, H, mask.data(), &iters);
        lvko_usac_find_homography(p1.data(), p2.data(), n, 10.0, 0.0, 7, 0, H, mask.data(), &iters);
        lvko_ref_estimate_affine_partial(p1.data(), p2.data(), n, 3.0, H, mask.data());
    }
    // mesh solver beyond the preset (17 x 5), three warm-started frames, then a point in the last cell row (refused)
    {
        lvko_mesh_solver* ms = lvko_mesh_solver_create(17, 5, 480, 270, 1.0f, 20.0f);
        const int n = 300; std::vector<float> a(2 * n), b(2 * n), off(2 * 17 * 5); std::vector<uint8_t> inl(n);
        for (int fr = 0; fr < 3; fr++)
        {
            for (int i = 0; i < n; i++) { a[2 * i] = 2 + frand() * 440; a[2 * i + 1] = 2 + frand() * 190; b[2 * i] = a[2 * i] * 1.01f + 1; b[2 * i + 1] = a[2 * i + 1] * 0.99f - 1; }
            if (lvko_mesh_solver_solve(ms, a.data(), b.data(), n, 480, 270, 1.0f, 10.0f, inl.data(), off.data()) != 0) { std::printf("mesh solve failed\n"); return 1; }
        }
        a[0] = 481.0f; a[1] = 271.0f;                      // beyond the last vertex: cell (16, 4), whose far corner is not in the mesh
        if (lvko_mesh_solver_solve(ms, a.data(), b.data(), n, 480, 270, 1.0f, 10.0f, inl.data(), off.data()) == 0) { std::printf("out-of-mesh point accepted\n"); return 1; }
        lvko_mesh_solver_destroy(ms);
    }
    // 4:2:0 <-> 4:4:4 with padded pitches, I420 and NV12; upscale + sharpen; lens map + map remap
    {
        const int r = 90, c = 162, ys = c + 5, cs = c / 2 + 3;
        render(frame, rows, cols, 1, 0);
        std::vector<uint8_t> y((size_t)ys * r), u((size_t)cs * r / 2 + cs), v((size_t)cs * r / 2 + cs), uv((size_t)(c + 6) * r / 2 + c), packed((size_t)r * c * 3);
        lvko_egress_yuv420(frame.data(), cols * 3, r, c, y.data(), ys, u.data(), cs, v.data(), cs, 0);
        lvko_ingest_yuv420(y.data(), ys, u.data(), cs, v.data(), cs, 0, r, c, packed.data(), c * 3);
        lvko_egress_yuv420(frame.data(), cols * 3, r, c, y.data(), ys, uv.data(), c + 6, nullptr, 0, 1);
        lvko_ingest_yuv420(y.data(), ys, uv.data(), c + 6, nullptr, 0, 1, r, c, packed.data(), c * 3);
        std::vector<uint8_t> up((size_t)200 * 355 * 3), sh(up.size());
        lvko_upscale(packed.data(), c * 3, r, c, up.data(), 355 * 3, 200, 355, 1, 2);
        lvko_sharpen(up.data(), 355 * 3, 200, 355, sh.data(), 355 * 3, 0.7f, 2);
        const double lens[9] = {0.8 * c, 0.8 * c, c / 2.0, r / 2.0, -0.12, 0.03, 0.001, -0.001, 0.002};
        std::vector<float> map((size_t)r * c * 2); int view[4];
        if (lvko_lens_offset_map(lens, r, c, map.data(), view) == 0)
        {
            const uint8_t bg[3] = {1, 2, 3};
            std::vector<uint8_t> warped(packed.size());
            lvko_remap_map(packed.data(), c * 3, r, c, warped.data(), c * 3, map.data(), bg, 1, 2);
        }
    }
    std::printf("oracle sanitize ok\n");
    return 0;
}
