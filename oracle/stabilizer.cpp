// ORACLE (test infrastructure only -- see lvk_oracle.h).
// CPU restatement of the stateful host logic of the stabilization path, on top of the oracle's image stages:
//   StabilizationFilter::{configure,filter,restart,reset_context}   Filters/StabilizationFilter.cpp:42-159
//   FrameTracker::{configure,track,restart,estimate_global_motion}   Vision/FrameTracker.cpp:57-196,325-375
//   FeatureDetector::{configure,detect,propagate,reset}              Vision/FeatureDetector.cpp:48-214
//   PathSmoother::{configure,next,restart}                           Vision/PathSmoother.cpp:36-145
//   WarpMesh arithmetic                                              Math/WarpMesh.cpp:318-551
//   StreamBuffer / SpatialMap / VirtualGrid semantics                Data/StreamBuffer.tpp, Data/SpatialMap.tpp, Math/VirtualGrid.cpp
// Float elementwise ops follow OpenCV's scalar definitions (separately rounded, no contraction).
#include "lvk_oracle.h"
#include "parallel.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <deque>
#include <algorithm>
#include <memory>

extern "C" {
int lvko_find_homography(const float*, const float*, int, double, double, double, double*, uint8_t*);
int lvko_estimate_affine_partial(const float*, const float*, int, double, double, double, double*, uint8_t*);
}

namespace {

inline int cv_round(float v) { return (int)lrintf(v); }             // cv::saturate_cast<int>(float) == cvRound

struct KeyPoint { float x, y, response; int class_id; };

// ---------------------------------------------------------------------------------------------- WarpMesh
struct Mesh
{
    int rows = 2, cols = 2;
    std::vector<float> v;                                           // rows x cols x (x, y) normalised backward offsets
    Mesh() { v.assign(8, 0.0f); }
    Mesh(int r, int c) : rows(r), cols(c), v((size_t)r * c * 2, 0.0f) {}
    void set_identity() { std::fill(v.begin(), v.end(), 0.0f); }    // WarpMesh.cpp:318-321
    void mul(float s) { for (float& f : v) f = f * s; }             // operator*=(float)   :548-551
    void add(const Mesh& o) { for (size_t i = 0; i < v.size(); i++) v[i] = v[i] + o.v[i]; }        // :493-498
    void sub(const Mesh& o) { for (size_t i = 0; i < v.size(); i++) v[i] = v[i] - o.v[i]; }        // :502-507
    void combine(const Mesh& o, float s) { for (size_t i = 0; i < v.size(); i++) v[i] = o.v[i] * s + v[i]; }  // cv::scaleAdd :445-448
    void clamp(float mx, float my)                                  // :411-417
    {
        for (size_t i = 0; i < v.size(); i += 2)
        {
            v[i] = std::min(std::max(v[i], -mx), mx);
            v[i + 1] = std::min(std::max(v[i + 1], -my), my);
        }
    }
    void crop_in(float rx, float ry, float rw, float rh)            // :379-390
    {
        const float sx = (rw - 1.0f) / (float)(cols - 1), sy = (rh - 1.0f) / (float)(rows - 1);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++)
            {
                v[((size_t)r * cols + c) * 2] += (float)c * sx + rx;
                v[((size_t)r * cols + c) * 2 + 1] += (float)r * sy + ry;
            }
    }
    void set_to_homography(const double H[9], float scale_w, float scale_h)     // :333-342 + Homography.cpp:125-130
    {
        const float csx = scale_w / (float)(cols - 1), csy = scale_h / (float)(rows - 1);
        const float nfx = 1.0f / scale_w, nfy = 1.0f / scale_h;
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++)
            {
                const float sx = (float)c * csx, sy = (float)r * csy;
                // cv::perspectiveTransform on Point2f with a CV_64F matrix: double math, cast back to float
                double w = sx * H[6] + sy * H[7] + H[8];
                float tx = 0.0f, ty = 0.0f;
                if (std::fabs(w) > 1.1920928955078125e-07)
                {
                    w = 1. / w;
                    tx = (float)((sx * H[0] + sy * H[1] + H[2]) * w);
                    ty = (float)((sx * H[3] + sy * H[4] + H[5]) * w);
                }
                v[((size_t)r * cols + c) * 2] = (sx - tx) * nfx;
                v[((size_t)r * cols + c) * 2 + 1] = (sy - ty) * nfy;
            }
    }
};

// ---------------------------------------------------------------------------------------------- FeatureDetector
struct Region { float bx, by, bw, bh; int threshold; size_t load; };

struct Detector
{
    lvko_stab_settings s{};
    int grid_cols = 1, grid_rows = 1;
    float key_w = 1, key_h = 1;                                     // suppression grid key size
    int reg_cols = 1, reg_rows = 1;
    float reg_w = 1, reg_h = 1;
    std::vector<Region> regions;
    std::vector<long> grid;                                         // link into features, -1 = empty
    size_t grid_count = 0;
    std::vector<KeyPoint> features;                                 // m_Features (propagated between frames)
    size_t target = 0, min_load = 0;
    std::vector<int> fast_buf;

    void configure(const lvko_stab_settings& st)                    // FeatureDetector.cpp:48-83
    {
        s = st;
        const int gc = cv_round((float)s.detection_width * s.max_feature_density);
        const int gr = cv_round((float)s.detection_height * s.max_feature_density);
        if (gc != grid_cols || gr != grid_rows || grid.empty())
        {
            grid_cols = gc; grid_rows = gr;
            grid.assign((size_t)gc * gr, -1); grid_count = 0;
        }
        key_w = (float)s.detection_width / (float)grid_cols;
        key_h = (float)s.detection_height / (float)grid_rows;
        reg_cols = s.detection_regions_x; reg_rows = s.detection_regions_y;
        reg_w = (float)s.detection_width / (float)reg_cols;
        reg_h = (float)s.detection_height / (float)reg_rows;
        regions.clear();                                            // construct_detection_regions :87-110
        for (int r = 0; r < reg_rows; r++)
            for (int c = 0; c < reg_cols; c++)
                regions.push_back(Region{(float)c * reg_w, (float)r * reg_h, reg_w, reg_h, 10, 0});
        const size_t max_features = (size_t)grid_cols * grid_rows;
        const float max_regions = (float)(reg_cols * reg_rows);
        const float max_region_features = (float)max_features / max_regions;
        const float density_ratio = s.min_feature_density / s.max_feature_density;
        min_load = (size_t)(max_region_features * density_ratio);
        target = (size_t)(s.accumulation_rate * max_region_features);
    }

    size_t key_index(float x, float y) const                        // VirtualGrid::key_of + key_to_index
    {
        const size_t kx = (size_t)((x - 0.0f) / key_w), ky = (size_t)((y - 0.0f) / key_h);
        return ky * (size_t)grid_cols + kx;
    }

    float distribution_quality() const                              // SpatialMap.tpp:589-625
    {
        if (grid_count == 0) return 1.0f;
        if (grid_cols <= 4 || grid_rows <= 4) return (float)grid_count / (float)grid.size();
        const float ksw = (float)grid_cols / 4.0f, ksh = (float)grid_rows / 4.0f;
        size_t buckets[16] = {0};
        const size_t ideal = (size_t)((float)grid_count / 16.0f);
        float excess = 0.0f;
        for (int ky = 0; ky < grid_rows; ky++)
            for (int kx = 0; kx < grid_cols; kx++)
            {
                if (grid[(size_t)ky * grid_cols + kx] < 0) continue;
                const size_t sx = (size_t)((float)kx / ksw), sy = (size_t)((float)ky / ksh);
                if (++buckets[sy * 4 + sx] > ideal) excess += 1.0f;
            }
        return 1.0f - (excess / (float)(grid_count - ideal));
    }

    float detect(const uint8_t* frame, int step, std::vector<KeyPoint>& out)      // :114-178
    {
        for (Region& rg : regions)
        {
            if (s.force_detection || rg.load <= min_load)
            {
                // cv::Rect2f -> cv::Rect conversion rounds each member (FeatureDetector.cpp:132)
                const int rx = cv_round(rg.bx), ry = cv_round(rg.by), rw = cv_round(rg.bw), rh = cv_round(rg.bh);
                fast_buf.resize((size_t)std::max(1, rw * rh) * 3);
                const int n = lvko_fast9_16(frame, step, rx, ry, rw, rh, rg.threshold, fast_buf.data(), rw * rh);
                for (int i = 0; i < n; i++)
                {
                    KeyPoint f{(float)fast_buf[3 * i] + rg.bx, (float)fast_buf[3 * i + 1] + rg.by, (float)fast_buf[3 * i + 2], 0};
                    long& link = grid[key_index(f.x, f.y)];
                    if (link < 0) { link = (long)features.size(); grid_count++; features.push_back(f); }
                    else
                    {
                        KeyPoint& mx = features[(size_t)link];
                        if (f.response > mx.response && mx.class_id <= 0) mx = f;
                    }
                }
                const size_t cnt = (size_t)n;
                if (cnt > target + 150) rg.threshold = std::min(rg.threshold + 5, 250);                       // step(threshold, 250, 5)
                else if (cnt < target - 150) rg.threshold = rg.threshold > 10 ? std::max(rg.threshold - 5, 10) : std::min(rg.threshold + 5, 10);
            }
            rg.load = 0;
        }
        out.swap(features);
        features.clear();
        const float q = distribution_quality();
        std::fill(grid.begin(), grid.end(), -1); grid_count = 0;
        return q;
    }

    void propagate(const std::vector<KeyPoint>& in)                  // :182-205
    {
        for (const KeyPoint& f : in)
        {
            if (!(f.x >= 0.0f && f.x < (float)s.detection_width && f.y >= 0.0f && f.y < (float)s.detection_height)) continue;
            long& link = grid[key_index(f.x, f.y)];
            if (link < 0)
            {
                link = (long)features.size(); grid_count++;
                const size_t rx = (size_t)(f.x / reg_w), ry = (size_t)(f.y / reg_h);
                regions[ry * (size_t)reg_cols + rx].load++;
                features.push_back(f);
            }
            else
            {
                KeyPoint& mx = features[(size_t)link];
                if (f.response > mx.response && f.class_id >= mx.class_id) mx = f;
            }
        }
    }

    void reset()                                                    // :209-214 (m_Features is NOT cleared -- reference quirk)
    {
        std::fill(grid.begin(), grid.end(), -1); grid_count = 0;
        for (Region& rg : regions) rg.load = 0;
    }
};

// ---------------------------------------------------------------------------------------------- FrameTracker
struct Tracker
{
    lvko_stab_settings s{};
    Detector det;
    lvko_mesh_solver* solver = nullptr;             // m_MeshConstraints + m_OptimizedMesh
    int solver_cols = 0, solver_rows = 0;

    Tracker()
    {
        // FrameTracker(const FrameTrackerSettings& = {}) runs configure(defaults) + restart() (FrameTracker.cpp:41-53):
        // the static mesh constraints first exist for a 16x16 mesh over a 256x256 region with weights 1.0 / 20.0.
        lvko_stab_default_settings(&s);
        s.motion_width = 16; s.motion_height = 16;  // FrameTrackerSettings::motion_resolution default (FrameTracker.hpp:33)
        solver = lvko_mesh_solver_create(16, 16, 256.0f, 256.0f, s.temporal_smoothing, s.local_smoothing);
        solver_cols = solver_rows = 16;
        det.configure(s);
    }
    ~Tracker() { lvko_mesh_solver_destroy(solver); }
    Tracker(const Tracker&) = delete;
    Tracker& operator=(const Tracker&) = delete;
    bool initialized = false;
    std::vector<uint8_t> prev, cur;                                 // tracking-resolution gray frames
    int prev_w = 0, prev_h = 0, cur_w = 0, cur_h = 0;
    std::vector<KeyPoint> tracked;
    std::vector<float> tracked_pts, matched_pts;
    std::vector<uint8_t> match_status, inlier_status;
    float stability = 0.0f;
    // debug taps
    double last_H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int last_detected = 0, last_matched = 0;
    float last_distribution = 0.0f;
    std::vector<float> last_p1, last_p2; int last_estimator = 0;    // the pairs the last motion estimate saw; 0 none, 1 homography, 2 affine fallback, 3 mesh

    void configure(const lvko_stab_settings& st)                    // FrameTracker.cpp:57-93
    {
        const bool res_changed = (st.detection_width != s.detection_width || st.detection_height != s.detection_height);
        det.configure(st);
        if (st.motion_width != s.motion_width || st.motion_height != s.motion_height)
        {
            // FrameTracker.cpp:74-82: regenerated for the NEW region with the PREVIOUS settings' smoothing weights
            lvko_mesh_solver_destroy(solver);
            solver = lvko_mesh_solver_create(st.motion_width, st.motion_height, (float)st.detection_width, (float)st.detection_height,
                                             s.temporal_smoothing, s.local_smoothing);
            solver_cols = st.motion_width; solver_rows = st.motion_height;
        }
        if (res_changed && initialized)
        {
            matched_pts.clear();
            det.reset();
            // the reference rescales the cached frame to the OLD resolution (a no-op resize); the size mismatch
            // then costs exactly one nullopt frame in track() (FrameTracker.cpp:120-124)
        }
        s = st;
    }

    void restart()                                                  // :97-104
    {
        stability = 0.0f;
        tracked.clear();
        det.reset();
        initialized = false;
        lvko_mesh_solver_reset(solver);
    }

    // returns true and fills `motion` when a motion estimate exists (std::optional<WarpMesh>)
    const double* lens_model = nullptr;        // fused lens mode: estimate motion between lens-corrected point positions
    // wall time per stage, accumulated (bench.py's cpu_baseline: BASELINE.md section 3 "per-stage ms (downscale, detect, LK, estimate, smooth, remap)")
    double stage_ms[6] = {0, 0, 0, 0, 0, 0};
    struct StageClock
    {
        double& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        explicit StageClock(double& a) : acc(a) {}
        ~StageClock() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    };

    bool track(const uint8_t* frame, int step, int pix_stride, int rows, int cols, Mesh& motion, int luma_channel = 0)     // :108-196
    {
        stability = 0.0f;
        prev.swap(cur); std::swap(prev_w, cur_w); std::swap(prev_h, cur_h);
        cur_w = s.detection_width; cur_h = s.detection_height;
        cur.resize((size_t)cur_w * cur_h);
        int resize_rc;
        { StageClock clk(stage_ms[0]); resize_rc = lvko_luma_area_resize(frame, step, pix_stride, luma_channel, rows, cols, cur.data(), cur_w, cur_h, cur_w); }
        if (resize_rc != 0)
        {
            // (a refused resize used to leave the tracking frame empty and the filter silently tracking nothing: a test ran on that for two rounds)
            std::fprintf(stderr, "lvk oracle: the tracking-frame resize refused a %d x %d frame\n", cols, rows);
            std::abort();
        }
        last_detected = last_matched = 0; last_distribution = 0.0f; last_estimator = 0;
        if (!initialized || cur_w != prev_w || cur_h != prev_h) { initialized = true; return false; }

        float distribution;
        { StageClock clk(stage_ms[1]); distribution = det.detect(cur.data(), cur_w, tracked); }
        last_distribution = distribution; last_detected = (int)tracked.size();
        if (tracked.size() < (size_t)s.min_motion_samples || distribution < s.uniformity_threshold) { tracked.clear(); return false; }

        const int n = (int)tracked.size();
        tracked_pts.resize((size_t)n * 2); matched_pts.resize((size_t)n * 2); match_status.resize(n);
        for (int i = 0; i < n; i++) { tracked_pts[2 * i] = tracked[i].x; tracked_pts[2 * i + 1] = tracked[i].y; }
        {
            StageClock clk(stage_ms[2]);
            lvko_pyrlk(prev.data(), prev_w, cur.data(), cur_w, cur_h, cur_w, tracked_pts.data(), n, matched_pts.data(), match_status.data(),
                       11, 11, 3, 5, 0.01, 1e-4);                   // :33-35,42-48
        }

        // fused lens mode: the motion is estimated between lens-corrected positions; a match whose corrected positions leave the
        // tracking region is not visible in the corrected frame (the reference chain LC -> VS could not have tracked it): drop it
        std::vector<float> und_t, und_m;
        if (lens_model)
        {
            const double sx = (double)cols / (double)cur_w, sy = (double)rows / (double)cur_h;
            und_t.resize((size_t)n * 2); und_m.resize((size_t)n * 2);
            lvko_lens_undistort_points(lens_model, sx, sy, tracked_pts.data(), n, und_t.data());
            lvko_lens_undistort_points(lens_model, sx, sy, matched_pts.data(), n, und_m.data());
            const float w = (float)cur_w, h = (float)cur_h;
            auto inside = [&](const float* p) { return p[0] >= 0.0f && p[0] < w && p[1] >= 0.0f && p[1] < h; };
            for (int k = 0; k < n; k++)
                if (!(inside(&und_t[2 * k]) && inside(&und_m[2 * k]))) match_status[k] = 0;
        }

        // fast_filter(features, tracked, matched, status): back-to-front swap-erase (Container.tpp:97-121)
        {
            size_t m = (size_t)n;
            for (int k = n - 1; k >= 0; k--)
                if (!match_status[k])
                {
                    m--;
                    std::swap(tracked[k], tracked[m]);
                    std::swap(tracked_pts[2 * k], tracked_pts[2 * m]); std::swap(tracked_pts[2 * k + 1], tracked_pts[2 * m + 1]);
                    std::swap(matched_pts[2 * k], matched_pts[2 * m]); std::swap(matched_pts[2 * k + 1], matched_pts[2 * m + 1]);
                    if (lens_model)
                    {
                        std::swap(und_t[2 * k], und_t[2 * m]); std::swap(und_t[2 * k + 1], und_t[2 * m + 1]);
                        std::swap(und_m[2 * k], und_m[2 * m]); std::swap(und_m[2 * k + 1], und_m[2 * m + 1]);
                    }
                }
            tracked.resize(m); tracked_pts.resize(m * 2); matched_pts.resize(m * 2);
            if (lens_model) { und_t.resize(m * 2); und_m.resize(m * 2); }
        }
        const int m = (int)tracked.size();
        last_matched = m;
        if ((size_t)m < (size_t)s.min_motion_samples) { tracked.clear(); return false; }

        motion = Mesh(s.motion_height, s.motion_width);
        inlier_status.assign(m, 0);
        const std::vector<float> raw_matched = matched_pts;                      // propagation stays in raw coordinates
        if (lens_model) { tracked_pts = und_t; matched_pts = und_m; }
        last_p1 = tracked_pts; last_p2 = matched_pts;
        last_estimator = s.track_local_motions ? 3 : (distribution > 0.6f ? 1 : 2);
        StageClock estimate_clk(stage_ms[3]);                                    // (motion estimate + the propagation bookkeeping behind it)
        if (s.track_local_motions)
        {
            if (lvko_mesh_solver_solve(solver, tracked_pts.data(), matched_pts.data(), m, (float)cur_w, (float)cur_h,
                                       s.temporal_smoothing, s.acceptance_threshold, inlier_status.data(), motion.v.data()) != 0)
                return false;
        }
        else
        {
            const bool homography = distribution > 0.6f;            // HOMOGRAPHY_DISTRIBUTION_THRESHOLD :37
            if (homography) lvko_find_homography(tracked_pts.data(), matched_pts.data(), m, s.acceptance_threshold, cur_w, cur_h, last_H, inlier_status.data());
            else lvko_estimate_affine_partial(tracked_pts.data(), matched_pts.data(), m, s.acceptance_threshold, cur_w, cur_h, last_H, inlier_status.data());
            motion.set_to_homography(last_H, (float)cur_w, (float)cur_h);
        }

        // tracking stability = inlier ratio (Container.tpp ratio_of)
        size_t inl = 0; for (uint8_t b : inlier_status) inl += b ? 1 : 0;
        stability = (float)inl / (float)inlier_status.size();

        for (int i = m - 1; i >= 0; i--)                            // :183-192
        {
            if (inlier_status[i]) { tracked[i].class_id++; tracked[i].x = raw_matched[2 * i]; tracked[i].y = raw_matched[2 * i + 1]; }
            else { std::swap(tracked[i], tracked.back()); tracked.pop_back(); }
        }
        det.propagate(tracked);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------- PathSmoother
struct Smoother
{
    lvko_stab_settings s{};
    bool configured = false;
    double smoothing_factor = 0.0, base_factor = 0.0;
    std::deque<Mesh> trajectory;                                    // always full: 2N+1 meshes, [0] = oldest
    Mesh trace, position, scene_crop;
    float margin_x = 0, margin_y = 0, margin_w = 1, margin_h = 1;   // m_SceneMargins

    void configure(const lvko_stab_settings& st)                    // PathSmoother.cpp:36-80
    {
        const int mr = st.motion_height, mc = st.motion_width;
        if (!configured || position.rows != mr || position.cols != mc)
        {
            const size_t cap = configured ? trajectory.size() : 1;
            trajectory.assign(cap, Mesh(mr, mc));
            trace = Mesh(mr, mc); position = Mesh(mr, mc);
        }
        const size_t window = 2 * (size_t)st.predictive_samples + 1;
        if (trajectory.size() != window)
        {
            // StreamBuffer::resize keeps the newest elements; pad_front fills the missing oldest slots with identity
            while (trajectory.size() > window) trajectory.pop_front();
            while (trajectory.size() < window) trajectory.push_front(Mesh(mr, mc));
            position = trajectory.front();
            const size_t centre = (trajectory.size() - 1) / 2;
            for (size_t i = 1; i <= centre; i++) position.add(trajectory[i]);
            base_factor = (double)window / 12.0;
        }
        // crop<float>({1,1}, corrective_limits) (Functions/Math.tpp:218-233)
        margin_x = (1.0f * st.corrective_limit_x) / 2; margin_y = (1.0f * st.corrective_limit_y) / 2;
        margin_w = 1.0f - 1.0f * st.corrective_limit_x; margin_h = 1.0f - 1.0f * st.corrective_limit_y;
        scene_crop = Mesh(mr, mc);
        scene_crop.crop_in(margin_x, margin_y, margin_w, margin_h);
        s = st; configured = true;
    }

    Mesh next(const Mesh& motion)                                   // :84-135
    {
        position.sub(trajectory.front());
        trajectory.pop_front(); trajectory.push_back(motion);       // StreamBuffer::push on a full buffer
        const size_t n = trajectory.size(), centre = (n - 1) / 2;
        position.add(trajectory[centre]);

        // cv::getGaussianKernel(n, sigma, CV_32F)
        const double sigma = base_factor + smoothing_factor;
        std::vector<double> k(n); double sum = 0;
        const double scale2x = -0.5 / (sigma * sigma);
        for (size_t i = 0; i < n; i++) { const double x = (double)i - (double)(n - 1) * 0.5; k[i] = std::exp(scale2x * x * x); sum += k[i]; }
        sum = 1. / sum;
        std::vector<float> filt(n);
        for (size_t i = 0; i < n; i++) filt[i] = (float)(k[i] * sum);

        float weight = 1.0f;
        trace = trajectory.front();
        for (size_t i = 1; i < n; i++) { weight -= filt[i - 1]; trace.combine(trajectory[i], weight); }
        Mesh correction = trace; correction.sub(position);

        float max_drift = 0.0f;
        for (size_t i = 0; i < correction.v.size(); i += 2)
        {
            max_drift = std::max(max_drift, std::fabs(correction.v[i]) / margin_x);
            max_drift = std::max(max_drift, std::fabs(correction.v[i + 1]) / margin_y);
        }
        if (max_drift > 1.0f) { correction.clamp(margin_x, margin_y); max_drift = 1.0f; }

        // hysteresis<double>(drift, 0.3, smoothing_steps, 0.7, 0.0) (Functions/Logic.tpp:53-65) then EMA (Math.tpp:198-204)
        const double drift = (double)max_drift;
        const double tgt = drift >= 0.7 ? 0.0 : (drift <= 0.3 ? (double)s.smoothing_steps : drift);
        smoothing_factor = smoothing_factor + s.response_rate * (tgt - smoothing_factor);
        return correction;
    }

    void restart()                                                  // :139-145
    {
        for (Mesh& m : trajectory) m.set_identity();
        position.set_identity(); trace.set_identity();
    }
};

} // namespace

// ---------------------------------------------------------------------------------------------- StabilizationFilter
struct lvko_stab
{
    lvko_stab_settings s{};
    bool configured = false;
    Tracker tracker;
    Smoother smoother;
    struct QFrame { std::vector<uint8_t> px; int rows, cols; uint64_t ts; int format = 4; };
    std::deque<QFrame> queue;                                       // m_FrameQueue (capacity predictive_samples + 1)
    size_t queue_capacity = 1;
    float scene_quality = 0.0f, trust = 0.0f;
    Mesh last_motion, last_correction;
    bool lens = false;                                              // fused lens mode (lvko_stab_set_lens)
    double lens_params[9] = {}, lens_model[17] = {};
    int lens_rows = 0, lens_cols = 0;

    const double* model_for(int rows, int cols)
    {
        if (!lens) return nullptr;
        if (rows != lens_rows || cols != lens_cols) { lvko_lens_model(lens_params, rows, cols, lens_model); lens_rows = rows; lens_cols = cols; }
        return lens_model;
    }

    void reset_context() { tracker.restart(); smoother.restart(); }             // StabilizationFilter.cpp:155-159

    void configure(const lvko_stab_settings& st)                                // :42-65
    {
        if (configured && s.stabilize_output && !st.stabilize_output) reset_context();
        s = st;
        smoother.configure(s);
        queue_capacity = (size_t)s.predictive_samples + 1;
        while (queue.size() > queue_capacity) queue.pop_front();                // StreamBuffer::resize keeps the newest
        tracker.configure(s);
        configured = true;
    }
};

extern "C" {

void lvko_stab_default_settings(lvko_stab_settings* s)
{
    // library defaults: FeatureDetector.hpp:28-37, FrameTracker.hpp:31-44, PathSmoother.hpp:29-39, StabilizationFilter.hpp:28-39
    s->detection_width = 256; s->detection_height = 256; s->detection_regions_x = 2; s->detection_regions_y = 2; s->force_detection = 0;
    s->max_feature_density = 0.20f; s->min_feature_density = 0.05f; s->accumulation_rate = 2.0f;
    s->track_local_motions = 1; s->temporal_smoothing = 1.0f; s->local_smoothing = 20.0f;
    s->min_motion_samples = 75; s->acceptance_threshold = 8.0f; s->uniformity_threshold = 0.20f;
    s->predictive_samples = 10; s->corrective_limit_x = 0.1f; s->corrective_limit_y = 0.1f; s->smoothing_steps = 20.0f; s->response_rate = 0.04f;
    s->motion_width = 2; s->motion_height = 2;
    s->background[0] = 255; s->background[1] = 0; s->background[2] = 255;
    s->crop_to_stable_region = 0; s->stabilize_output = 1; s->min_scene_quality = 0.8f; s->min_tracking_quality = 0.3f;
}

lvko_stab* lvko_stab_create(const lvko_stab_settings* settings)
{
    auto* st = new lvko_stab();
    st->configure(*settings);
    st->reset_context();                                            // FrameTracker ctor calls restart() (FrameTracker.cpp:53)
    return st;
}

void lvko_stab_destroy(lvko_stab* st) { delete st; }

void lvko_stab_configure(lvko_stab* st, const lvko_stab_settings* settings) { st->configure(*settings); }

// StabilizationFilter::draw_trackers / draw_motion_mesh (StabilizationFilter.cpp:163-188); colours Functions/Drawing.hpp:27-71
static void overlay_colours(int format, double red[3], double green[3], double blue[3])
{
    const double R[3][3] = {{0, 0, 255}, {255, 0, 0}, {76, 84, 255}}, G[3][3] = {{0, 255, 0}, {0, 255, 0}, {149, 43, 21}},
                 B[3][3] = {{255, 0, 0}, {0, 0, 255}, {29, 255, 107}};
    const int k = format == 4 ? 2 : (format == 2 || format == 3 ? 1 : 0);
    for (int i = 0; i < 3; i++) { red[i] = R[k][i]; green[i] = G[k][i]; blue[i] = B[k][i]; }
}

void lvko_stab_draw_trackers(lvko_stab* st)
{
    if (!st || st->queue.empty()) return;
    lvko_stab::QFrame& f = st->queue.back();
    double r[3], g[3], b[3];
    overlay_colours(f.format, r, g, b);
    uint8_t col[3];
    for (int i = 0; i < 3; i++) col[i] = (uint8_t)(r[i] + (double)st->trust * (g[i] - r[i]));      // lerp (Math.tpp:124-129), Vec4b cast
    std::vector<float> pts;
    for (const KeyPoint& k : st->tracker.tracked) { pts.push_back(k.x); pts.push_back(k.y); }
    const float sx = (float)f.cols / (float)st->tracker.s.detection_width, sy = (float)f.rows / (float)st->tracker.s.detection_height;
    lvko_draw_crosses(f.px.data(), f.cols * 3, f.rows, f.cols, pts.data(), (int)(pts.size() / 2), sx, sy, col, 7, 4);    // FrameTracker.cpp:498-503
}

void lvko_stab_draw_motion_mesh(lvko_stab* st)
{
    if (!st || st->queue.empty()) return;
    lvko_stab::QFrame& f = st->queue.back();
    double r[3], g[3], b[3];
    overlay_colours(f.format, r, g, b);
    const uint8_t col[3] = {(uint8_t)b[0], (uint8_t)b[1], (uint8_t)b[2]};
    lvko_draw_grid(f.px.data(), f.cols * 3, f.rows, f.cols, st->s.motion_width - 1, st->s.motion_height - 1, col, 1);
}

// Fused lens mode (this repo's design, BASELINE config 5): params = camera profile or NULL (off).  Restarts the filter.
void lvko_stab_restart(lvko_stab* st);
void lvko_stab_set_lens(lvko_stab* st, const double* params)
{
    st->lens = params != nullptr;
    if (params) std::memcpy(st->lens_params, params, sizeof(st->lens_params));
    st->lens_rows = st->lens_cols = 0;
    lvko_stab_restart(st);
}

void lvko_stab_restart(lvko_stab* st)                               // StabilizationFilter::restart :139-144
{
    st->scene_quality = 1.0f;
    st->queue.clear();
    st->reset_context();
}

// StabilizationFilter::filter (StabilizationFilter.cpp:69-135).  frame: packed 8UC3 YUV.  Returns 1 when `out` was
// produced (it then carries the delayed frame's timestamp in *out_ts), 0 while the delay builds, < 0 on error.
int lvko_stab_push(lvko_stab* st, const uint8_t* frame, int step, int rows, int cols, uint64_t ts,
                   uint8_t* out, int out_step, uint64_t* out_ts, int nthreads)
{
    return lvko_stab_push_fmt(st, frame, step, rows, cols, ts, 4, out, out_step, out_ts, nthreads);
}

// format: VideoFrame::Format of the 3-channel frame (0 = BGR, 2 = RGB, 4 = YUV; VideoFrame.hpp) -- selects the tracking luma
// (VideoFrame.cpp:194,260) and the EASU program (Image.cpp:36-41)
int lvko_stab_push_fmt(lvko_stab* st, const uint8_t* frame, int step, int rows, int cols, uint64_t ts, int format,
                       uint8_t* out, int out_step, uint64_t* out_ts, int nthreads)
{
    if (!st || !frame || rows <= 0 || cols <= 0 || !(format == 0 || format == 2 || format == 4)) return -1;
    // the tracker's row / point-parallel stages follow the push's thread count (restored on every exit)
    struct ThreadScope { int prev; explicit ThreadScope(int n) : prev(lvko_set_num_threads(n)) {} ~ThreadScope() { lvko_set_num_threads(prev); } } thread_scope(nthreads);
    const int luma_channel = format == 4 ? 0 : (format == 0 ? -1 : -2);
    lvko_stab::QFrame qf; qf.rows = rows; qf.cols = cols; qf.ts = ts; qf.format = format; qf.px.resize((size_t)rows * cols * 3);
    for (int y = 0; y < rows; y++) std::memcpy(&qf.px[(size_t)y * cols * 3], frame + (size_t)y * step, (size_t)cols * 3);
    const uint8_t bg[3] = {(uint8_t)st->s.background[0], (uint8_t)st->s.background[1], (uint8_t)st->s.background[2]};

    if (!st->s.stabilize_output)                                    // :77-95
    {
        if (st->queue.size() == st->queue_capacity) st->queue.pop_front();
        st->queue.push_back(std::move(qf));
        if (st->queue.size() != st->queue_capacity) return 0;
        lvko_stab::QFrame f = std::move(st->queue.front()); st->queue.pop_front();
        if (st->s.crop_to_stable_region || st->lens)
        {
            const Mesh ident(2, 2);
            const Mesh& m = st->s.crop_to_stable_region ? st->smoother.scene_crop : ident;
            lvko_warpmesh_apply_lens(f.px.data(), f.cols * 3, f.rows, f.cols, out, out_step, m.v.data(), m.rows, m.cols, bg, f.format == 4 ? 1 : 0, nthreads,
                                     st->model_for(f.rows, f.cols));
        }
        else for (int y = 0; y < f.rows; y++) std::memcpy(out + (size_t)y * out_step, &f.px[(size_t)y * f.cols * 3], (size_t)f.cols * 3);
        if (out_ts) *out_ts = f.ts;
        return 1;
    }

    Mesh motion(st->s.motion_height, st->s.motion_width);           // m_NullMotion
    Mesh tracked_motion;
    st->tracker.lens_model = st->model_for(rows, cols);
    if (st->tracker.track(frame, step, 3, rows, cols, tracked_motion, luma_channel)) motion = tracked_motion;

    // quality assurance (:101-115); exp_moving_average / step from Functions/Math.tpp:133-142,198-204
    const float tq = st->tracker.stability;
    st->scene_quality = st->scene_quality + 0.1f * (tq - st->scene_quality);
    if (tq < st->s.min_tracking_quality) st->trust = 0.0f;
    else if (st->scene_quality < st->s.min_scene_quality) st->trust = st->trust > 0.0f ? std::max(st->trust - 0.05f, 0.0f) : std::min(st->trust + 0.05f, 0.0f);
    else st->trust = st->trust > 1.0f ? std::max(st->trust - 0.05f, 1.0f) : std::min(st->trust + 0.05f, 1.0f);
    motion.mul(st->trust);
    st->last_motion = motion;

    if (st->queue.size() == st->queue_capacity) st->queue.pop_front();
    st->queue.push_back(std::move(qf));

    Mesh correction;
    { Tracker::StageClock clk(st->tracker.stage_ms[4]); correction = st->smoother.next(motion); }
    if (st->queue.size() != st->queue_capacity) return 0;           // ready() == is_full()
    lvko_stab::QFrame f = std::move(st->queue.front()); st->queue.pop_front();
    if (st->s.crop_to_stable_region) correction.add(st->smoother.scene_crop);
    st->last_correction = correction;
    {
        Tracker::StageClock clk(st->tracker.stage_ms[5]);
        lvko_warpmesh_apply_lens(f.px.data(), f.cols * 3, f.rows, f.cols, out, out_step, correction.v.data(), correction.rows, correction.cols, bg, f.format == 4 ? 1 : 0, nthreads,
                                 st->model_for(f.rows, f.cols));
    }
    if (out_ts) *out_ts = f.ts;
    return 1;
}

void lvko_stab_get_stats(const lvko_stab* st, lvko_stab_stats* o)
{
    o->tracking_stability = st->tracker.stability;
    o->scene_quality = st->scene_quality;
    o->trust = st->trust;
    o->distribution = st->tracker.last_distribution;
    o->n_detected = st->tracker.last_detected;
    o->n_matched = st->tracker.last_matched;
    o->n_tracked = (int)st->tracker.tracked.size();
    o->smoothing_factor = st->smoother.smoothing_factor;
    o->frame_delay = st->s.predictive_samples;
    for (int i = 0; i < 9; i++) o->homography[i] = st->tracker.last_H[i];
}

// wall time the pushes so far spent per stage, milliseconds: downscale (luma + INTER_AREA), detect (FAST + suppression grid), LK, estimate
// (RANSAC / mesh solve + propagation), smooth (PathSmoother::next), remap (EASU warp of the delayed frame); reset != 0 zeroes the counters
void lvko_stab_get_stage_ms(lvko_stab* st, double out[6], int reset)
{
    for (int i = 0; i < 6; i++) { out[i] = st->tracker.stage_ms[i]; if (reset) st->tracker.stage_ms[i] = 0.0; }
}

// debug tap: the (tracked, matched) pairs the last motion estimate was computed from (after fast_filter, FrameTracker.cpp:149);
// *estimator: 0 no estimate ran for the last frame, 1 findHomography, 2 estimateAffinePartial2D, 3 estimate_local_motions
int lvko_stab_get_matches(const lvko_stab* st, float* p1, float* p2, int cap_pairs, int* estimator)
{
    const int n = (int)st->tracker.last_p1.size() / 2;
    if (estimator) *estimator = st->tracker.last_estimator;
    if (n > cap_pairs) return -1;
    if (n > 0)
    {
        std::memcpy(p1, st->tracker.last_p1.data(), (size_t)n * 2 * sizeof(float));
        std::memcpy(p2, st->tracker.last_p2.data(), (size_t)n * 2 * sizeof(float));
    }
    return n;
}

int lvko_stab_get_meshes(const lvko_stab* st, float* motion, float* correction, int cap_floats)
{
    const int n = (int)st->last_motion.v.size();
    if (n > cap_floats) return -1;
    std::memcpy(motion, st->last_motion.v.data(), n * sizeof(float));
    if ((int)st->last_correction.v.size() == n) std::memcpy(correction, st->last_correction.v.data(), n * sizeof(float));
    return n;
}

int lvko_stab_get_features(const lvko_stab* st, float* xy_resp_age, int cap)
{
    const int n = std::min(cap, (int)st->tracker.tracked.size());
    for (int i = 0; i < n; i++)
    {
        xy_resp_age[4 * i] = st->tracker.tracked[i].x; xy_resp_age[4 * i + 1] = st->tracker.tracked[i].y;
        xy_resp_age[4 * i + 2] = st->tracker.tracked[i].response; xy_resp_age[4 * i + 3] = (float)st->tracker.tracked[i].class_id;
    }
    return (int)st->tracker.tracked.size();
}

} // extern "C"
