// Host-side state machines of the stabilization path (no device code): warp-mesh arithmetic, the path
// smoother, and the feature detector's suppression grid / region bookkeeping.  These are tiny (<= a few
// thousand elements per frame) and inherently sequential, so they stay on the host exactly as in the reference.
//
// Reference (paths relative to LiveVisionKit/):
//   Math/WarpMesh.cpp:318-551, Math/Homography.cpp:125-130         -> WarpMeshF
//   Vision/PathSmoother.cpp:36-145, Functions/Logic.tpp:53-65       -> PathSmootherH
//   Vision/FeatureDetector.cpp:48-214, Data/SpatialMap.tpp:589-625  -> FeatureGridH
// Elementwise float ops follow OpenCV's scalar definitions: one rounding per op, no contraction.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstddef>
#include <vector>
#include <deque>
#include <algorithm>

#include "lvk_hip.h"
#include "lvk/WarpMeshCore.hpp"

namespace lvkh {

inline int cv_round(float v) { return (int)lrintf(v); }     // cv::saturate_cast<int>(float)

struct Feature { float x, y, response; int age; };          // cv::KeyPoint {pt, response, class_id}; class_id carries the age

// ------------------------------------------------------------------------------------------------ WarpMesh
// (shared with the C++ facade's lvk::WarpMesh: include/lvk/WarpMeshCore.hpp)
using WarpMeshF = lvk::detail::WarpMeshF;

// ------------------------------------------------------------------------------------------------ PathSmoother
class PathSmootherH
{
public:
    void configure(const lvk_stab_settings& s)
    {
        const int mr = s.motion_height, mc = s.motion_width;
        if (!m_configured || m_position.rows != mr || m_position.cols != mc)
        {
            const size_t keep = m_configured ? m_path.size() : 1;
            m_path.assign(keep, WarpMeshF(mr, mc));
            m_trace = WarpMeshF(mr, mc);
            m_position = WarpMeshF(mr, mc);
        }
        const size_t window = 2 * (size_t)s.predictive_samples + 1;
        if (m_path.size() != window)
        {
            while (m_path.size() > window) m_path.pop_front();                       // keep the newest samples
            while (m_path.size() < window) m_path.push_front(WarpMeshF(mr, mc));     // pad the front with identity
            m_position = m_path.front();
            for (size_t i = 1; i <= centre(); i++) m_position += m_path[i];
            m_base = (double)window / 12.0;
        }
        m_margin_x = (1.0f * s.corrective_limit_x) / 2;
        m_margin_y = (1.0f * s.corrective_limit_y) / 2;
        m_margin_w = 1.0f - 1.0f * s.corrective_limit_x;
        m_margin_h = 1.0f - 1.0f * s.corrective_limit_y;
        m_crop = WarpMeshF(mr, mc);
        m_crop.crop_in(m_margin_x, m_margin_y, m_margin_w, m_margin_h);
        m_steps = s.smoothing_steps;
        m_rate = s.response_rate;
        m_configured = true;
    }

    WarpMeshF next(const WarpMeshF& motion)
    {
        m_position -= m_path.front();
        m_path.pop_front();
        m_path.push_back(motion);
        m_position += m_path[centre()];

        // cv::getGaussianKernel(n, sigma, CV_32F): exp(-x^2 / (2 sigma^2)) in double, normalised, cast to float
        const size_t n = m_path.size();
        const double sigma = m_base + m_factor;
        const double s2 = -0.5 / (sigma * sigma);
        m_kd.resize(n);
        double total = 0;
        for (size_t i = 0; i < n; i++) { const double x = (double)i - (double)(n - 1) * 0.5; m_kd[i] = std::exp(s2 * x * x); total += m_kd[i]; }
        total = 1. / total;

        float weight = 1.0f;
        m_trace = m_path.front();
        for (size_t i = 1; i < n; i++)
        {
            weight -= (float)(m_kd[i - 1] * total);
            m_trace.scale_add(m_path[i], weight);
        }
        WarpMeshF corr = m_trace;
        corr -= m_position;

        float drift = 0.0f;
        for (size_t i = 0; i + 1 < corr.off.size(); i += 2)
        {
            drift = std::max(drift, std::fabs(corr.off[i]) / m_margin_x);
            drift = std::max(drift, std::fabs(corr.off[i + 1]) / m_margin_y);
        }
        if (drift > 1.0f) { corr.clamp(m_margin_x, m_margin_y); drift = 1.0f; }

        const double d = (double)drift;
        const double target = d >= 0.7 ? 0.0 : (d <= 0.3 ? (double)m_steps : d);      // hysteresis<double>
        m_factor = m_factor + m_rate * (target - m_factor);                            // exp_moving_average
        return corr;
    }

    void restart()
    {
        for (WarpMeshF& m : m_path) m.set_identity();
        m_position.set_identity();
        m_trace.set_identity();
    }

    const WarpMeshF& scene_crop() const { return m_crop; }
    double smoothing_factor() const { return m_factor; }
    void margins(float out[4]) const { out[0] = m_margin_x; out[1] = m_margin_y; out[2] = m_margin_w; out[3] = m_margin_h; }

private:
    size_t centre() const { return (m_path.size() - 1) / 2; }
    bool m_configured = false;
    std::deque<WarpMeshF> m_path;            // sliding window of 2N+1 frame motions, always full; front() = oldest
    WarpMeshF m_trace, m_position, m_crop;
    double m_factor = 0.0, m_base = 0.0;
    float m_margin_x = 0, m_margin_y = 0, m_margin_w = 1, m_margin_h = 1, m_steps = 20.0f, m_rate = 0.04f;
    std::vector<double> m_kd;
};

// ------------------------------------------------------------------------------------------------ FeatureDetector bookkeeping
class FeatureGridH
{
public:
    struct Zone { float x, y, w, h; int threshold; size_t load; bool ran; };

    void configure(const lvk_stab_settings& s)
    {
        m_w = s.detection_width; m_h = s.detection_height;
        const int gc = cv_round((float)m_w * s.max_feature_density), gr = cv_round((float)m_h * s.max_feature_density);
        if (gc != m_gc || gr != m_gr || m_cells.empty()) { m_gc = gc; m_gr = gr; m_cells.assign((size_t)gc * gr, -1); m_used = 0; m_used_cells.clear(); }
        m_cw = (float)m_w / (float)m_gc; m_ch = (float)m_h / (float)m_gr;
        // distribution_quality's 4 x 4 bucket of every cell, with the arithmetic of SpatialMap.tpp:589-625 (a table: the per-frame loop visits the occupied cells only)
        m_bucket.assign((size_t)m_gc * m_gr, 0);
        if (m_gc > 4 && m_gr > 4)
        {
            const float sw = (float)m_gc / 4.0f, sh = (float)m_gr / 4.0f;
            for (int y = 0; y < m_gr; y++)
                for (int x = 0; x < m_gc; x++) m_bucket[(size_t)y * m_gc + x] = (uint8_t)((size_t)((float)y / sh) * 4 + (size_t)((float)x / sw));
        }
        m_zc = s.detection_regions_x; m_zr = s.detection_regions_y;
        m_zw = (float)m_w / (float)m_zc; m_zh = (float)m_h / (float)m_zr;
        m_icw = 1.0 / (double)m_cw; m_ich = 1.0 / (double)m_ch; m_izw = 1.0 / (double)m_zw; m_izh = 1.0 / (double)m_zh;
        // FAST corners have integer coordinates: their cell column / row come from tables filled with the very expression cell_of() evaluates
        m_col_of.resize((size_t)m_w); m_row_of.resize((size_t)m_h);
        for (int x = 0; x < m_w; x++) m_col_of[(size_t)x] = (uint16_t)(size_t)((float)x / m_cw);
        for (int y = 0; y < m_h; y++) m_row_of[(size_t)y] = (uint16_t)(size_t)((float)y / m_ch);
        m_row_base.resize((size_t)m_h);
        for (int y = 0; y < m_h; y++) m_row_base[(size_t)y] = (uint32_t)m_row_of[(size_t)y] * (uint32_t)m_gc;
        held.reserve(capacity()); m_used_cells.reserve(capacity());
        zones.clear();
        for (int r = 0; r < m_zr; r++)
            for (int c = 0; c < m_zc; c++)
                zones.push_back(Zone{(float)c * m_zw, (float)r * m_zh, m_zw, m_zh, 10 /* FAST_MIN_THRESHOLD */, 0, false});
        const float per_zone = (float)((size_t)m_gc * m_gr) / (float)(m_zc * m_zr);
        m_min_load = (size_t)(per_zone * (s.min_feature_density / s.max_feature_density));
        m_target = (size_t)(s.accumulation_rate * per_zone);
        m_force = s.force_detection != 0;
    }

    size_t capacity() const { return (size_t)m_gc * m_gr; }

    // Which zones run FAST this frame (load <= minimum, or forced), with their integer ROI and threshold.
    void plan(std::vector<FastRegion>& out)
    {
        out.resize(zones.size());
        for (size_t i = 0; i < zones.size(); i++)
        {
            Zone& z = zones[i];
            z.ran = m_force || z.load <= m_min_load;
            out[i] = FastRegion{cv_round(z.x), cv_round(z.y), cv_round(z.w), cv_round(z.h), z.threshold, z.ran ? 1 : 0};
        }
    }

    // Feed one zone's raw FAST output (row-major packed x | y<<12 | score<<24, zone-local) through the suppression grid.
    void absorb(size_t zone, const uint32_t* kp, int count)
    {
        Zone& z = zones[zone];
        const bool integral = z.x == std::floor(z.x) && z.y == std::floor(z.y) && z.x >= 0.0f && z.y >= 0.0f;
        int i = 0;
        if (integral && z.x < 65536.0f && z.y < 65536.0f)
        {
            // The common case (zone origins on whole pixels, corners inside the tracking frame) without floats until a feature is stored: on
            // frames on which the detector runs this loop -- ~5 500 corners for the OBS presets -- sits between the detector's kernels and
            // the launch of the optical flow (26.6 -> 20.4 us on the build machine; same decisions in the same order).
            const unsigned zx = (unsigned)z.x, zy = (unsigned)z.y, w = (unsigned)m_col_of.size(), h = (unsigned)m_row_of.size();
            const uint16_t* col_of = m_col_of.data(); const uint32_t* row_base = m_row_base.data();
            long* cells = m_cells.data();
            for (; i < count; i++)
            {
                const uint32_t k = kp[i];
                const unsigned xi = (k & 0xFFFu) + zx, yi = ((k >> 12) & 0xFFFu) + zy;
                if (xi >= w || yi >= h) break;                                           // (never for a corner FAST reports: the general loop takes over)
                const uint32_t ci = row_base[yi] + col_of[xi];
                const float response = (float)(k >> 24);
                const long cell = cells[ci];
                if (cell < 0)
                {
                    cells[ci] = (long)held.size(); m_used++; m_used_cells.push_back(ci);
                    held.push_back(Feature{(float)xi, (float)yi, response, 0});
                }
                else
                {
                    Feature& best = held[(size_t)cell];
                    if (response > best.response && best.age <= 0) best = Feature{(float)xi, (float)yi, response, 0};
                }
            }
        }
        for (; i < count; i++)
        {
            Feature f{(float)(kp[i] & 0xFFFu) + z.x, (float)((kp[i] >> 12) & 0xFFFu) + z.y, (float)(kp[i] >> 24), 0};
            const size_t xi = (size_t)f.x, yi = (size_t)f.y;
            const size_t ci = (integral && xi < m_col_of.size() && yi < m_row_of.size()) ? (size_t)m_row_of[yi] * (size_t)m_gc + m_col_of[xi] : cell_of(f.x, f.y);
            long& cell = m_cells[ci];
            if (cell < 0) { cell = (long)held.size(); m_used++; m_used_cells.push_back((uint32_t)(&cell - m_cells.data())); held.push_back(f); }
            else if (f.response > held[(size_t)cell].response && held[(size_t)cell].age <= 0) held[(size_t)cell] = f;
        }
        feedback(z, (size_t)count);
    }

    // FeatureDetector.cpp:160-163: the region's threshold follows its raw corner count towards the target (m_target - 150 wraps for small
    // targets exactly as the reference's size_t arithmetic does)
    void feedback(Zone& z, size_t n) const
    {
        if (n > m_target + 150) z.threshold = std::min(z.threshold + 5, 250);
        else if (n < m_target - 150) z.threshold = z.threshold > 10 ? std::max(z.threshold - 5, 10) : std::min(z.threshold + 5, 10);
    }

    // ---- the suppression grid on the DEVICE (fast.hip k_fast_insert): on a frame on which the detector runs the corners go through the
    // grid inside the tracker's chain of kernels instead of on the host between two halves of it.  What the kernel needs from here: the
    // tables below, which cells hold a propagated feature (those never change: FeatureDetector.cpp:150 `class_id <= 0`), and the held
    // points; what comes back: the new features in list order.
    static constexpr int kDeviceMaxCells = 4096;
    size_t used_cells() const { return m_used; }
    int grid_cols() const { return m_gc; }
    int grid_rows() const { return m_gr; }
    const std::vector<uint16_t>& col_table() const { return m_col_of; }
    const std::vector<uint32_t>& row_base_table() const { return m_row_base; }
    const std::vector<uint8_t>& bucket_table() const { return m_bucket; }
    bool device_insert_ok() const
    {
        if (capacity() > (size_t)kDeviceMaxCells || m_w >= 4096 || m_h >= 4096 || m_mutable_marked) return false;
        for (const Zone& z : zones)          // corners at whole pixels inside the frame: the cell of a corner comes from the integer tables
            if (!(z.x == std::floor(z.x) && z.y == std::floor(z.y) && z.x >= 0.0f && z.y >= 0.0f && cv_round(z.x) + cv_round(z.w) <= m_w && cv_round(z.y) + cv_round(z.h) <= m_h)) return false;
        return true;
    }
    void occupancy(uint32_t* bits, int bucket_counts[16]) const     // one bit per cell that holds a propagated feature ((capacity + 31) / 32 words), and their number per distribution bucket
    {
        std::fill(bits, bits + (capacity() + 31) / 32, 0u);
        std::fill(bucket_counts, bucket_counts + 16, 0);
        for (uint32_t c : m_used_cells) { bits[c >> 5] |= 1u << (c & 31); bucket_counts[m_bucket[c] & 15]++; }
    }
    // End of detect() when the kernel has run the corners through the grid: kp = the new features in list order (frame coordinates,
    // x | y << 12 | score << 24), raw = every zone's raw corner count.  Same hand-over as absorb() x zones + finish().
    float finish_device(std::vector<Feature>& out, const uint32_t* kp, int n_new, const int* raw, int cap)
    {
        for (size_t i = 0; i < zones.size(); i++)
            if (zones[i].ran) feedback(zones[i], (size_t)std::min(raw[i], cap));
        for (int k = 0; k < n_new; k++)
        {
            const unsigned xi = kp[k] & 0xFFFu, yi = (kp[k] >> 12) & 0xFFFu;
            const uint32_t ci = m_row_base[yi] + m_col_of[xi];
            m_cells[ci] = (long)held.size(); m_used++; m_used_cells.push_back(ci);
            held.push_back(Feature{(float)xi, (float)yi, (float)(kp[k] >> 24), 0});
        }
        return finish(out);
    }

    // End of detect(): hand the surviving features over, compute the distribution quality, clear the grid.
    float finish(std::vector<Feature>& out)
    {
        for (Zone& z : zones) z.load = 0;
        out.swap(held);
        held.clear();
        const float q = quality();
        clear_cells();
        return q;
    }

    void propagate(const std::vector<Feature>& feats)
    {
        for (const Feature& f : feats)
        {
            if (!(f.x >= 0.0f && f.x < (float)m_w && f.y >= 0.0f && f.y < (float)m_h)) continue;
            long& cell = m_cells[cell_of(f.x, f.y)];
            if (cell < 0)
            {
                cell = (long)held.size(); m_used++; m_used_cells.push_back((uint32_t)(&cell - m_cells.data()));
                zones[quot(f.y, m_zh, m_izh) * (size_t)m_zc + quot(f.x, m_zw, m_izw)].load++;
                held.push_back(f);
                if (f.age <= 0) m_mutable_marked = true;           // (never for tracked features: their age was just raised)
            }
            else if (f.response > held[(size_t)cell].response && f.age >= held[(size_t)cell].age) { held[(size_t)cell] = f; if (f.age <= 0) m_mutable_marked = true; }
        }
    }

    void reset()       // FeatureDetector::reset: the grid and the loads are cleared, the held features are not (reference behaviour)
    {
        clear_cells();
        for (Zone& z : zones) z.load = 0;
    }

    // (size_t)(a / b) for a >= 0, b > 0 without the division: the binary64 product with 1 / b, rounded to binary32, is within one
    // binary32 ulp of the binary32 quotient, so the truncations agree unless an integer is that close -- then the division decides.
    // (Four divisions per feature were most of the 7 us per frame this bookkeeping cost on the critical path between two frames' kernels.)
    static size_t quot(float a, float b, double inv_b)
    {
        const float q = (float)((double)a * inv_b);
        const int k = (int)q;
        const float fr = q - (float)k;
        if (fr > 1e-3f && fr < 0.999f) return (size_t)k;
        return (size_t)(a / b);
    }
    std::vector<Zone> zones;
    std::vector<Feature> held;               // m_Features: propagated features waiting for the next detect()

private:
    size_t cell_of(float x, float y) const { return quot(y, m_ch, m_ich) * (size_t)m_gc + quot(x, m_cw, m_icw); }
    void clear_cells()
    {
        for (uint32_t c : m_used_cells) m_cells[c] = -1;
        m_used_cells.clear(); m_used = 0; m_mutable_marked = false;
    }

    float quality() const                    // SpatialMap::distribution_quality
    {
        if (m_used == 0) return 1.0f;
        if (m_gc <= 4 || m_gr <= 4) return (float)m_used / (float)m_cells.size();
        // every occupied cell beyond `ideal` in its bucket counts once: the sum does not depend on the order the cells are visited in
        size_t bucket[16] = {0};
        const size_t ideal = (size_t)((float)m_used / 16.0f);
        float excess = 0.0f;
        for (uint32_t c : m_used_cells)
            if (++bucket[m_bucket[c]] > ideal) excess += 1.0f;
        return 1.0f - (excess / (float)(m_used - ideal));
    }

    int m_w = 0, m_h = 0, m_gc = 0, m_gr = 0, m_zc = 1, m_zr = 1;
    float m_cw = 1, m_ch = 1, m_zw = 1, m_zh = 1;
    double m_icw = 1, m_ich = 1, m_izw = 1, m_izh = 1;
    std::vector<long> m_cells;
    std::vector<uint32_t> m_used_cells;      // indices of the occupied cells (what quality() and the clearing visit)
    std::vector<uint8_t> m_bucket;
    std::vector<uint16_t> m_col_of, m_row_of;
    std::vector<uint32_t> m_row_base;        // m_row_of[y] * m_gc
    size_t m_used = 0, m_min_load = 0, m_target = 0;
    bool m_force = false;
    bool m_mutable_marked = false;           // a marked cell holds a feature of age <= 0 (a stronger corner would replace it in place)
};

// ------------------------------------------------------------------------------------------------ local motion (vector field)
// FrameTracker::generate_mesh_constraints + estimate_local_motions (Vision/FrameTracker.cpp:200-321,380-457).
// The reference hands the sparse least-squares problem to Eigen::LeastSquaresConjugateGradient; this library solves the same
// problem exactly through its normal equations ON THE DEVICE (mesh.hip).  What stays on the host is the
// constant part: the static rows (temporal + similarity constraints) summed into a band matrix once per configuration.
class MeshSolverH
{
public:
    void generate(int cols, int rows, float gen_w, float gen_h, float temporal, float local)
    {
        m_cols = cols; m_rows = rows; m_n = 2 * cols * rows;
        m_hb = std::min(m_n - 1, 2 * (3 * cols + 3) + 1);
        m_ts = temporal;
        m_mesh.assign((size_t)m_n, 0.0f);
        m_static.assign((size_t)m_n * (m_hb + 1), 0.0);
        const float kw = (((float)cols / (float)(cols - 1)) * gen_w) / (float)cols;
        const float kh = (((float)rows / (float)(rows - 1)) * gen_h) / (float)rows;
        const double v1 = -((double)kw / (double)kh), v2 = -1.0 / v1;
        int cidx[4]; float cval[4];
        auto emit = [&](int n) {                               // one constraint row: add its outer product to the band
            for (int p = 0; p < n; p++)
                for (int q = 0; q < n; q++)
                    if (cidx[p] >= cidx[q]) at(m_static, cidx[p], cidx[q]) = at(m_static, cidx[p], cidx[q]) + (double)cval[p] * (double)cval[q];
        };
        m_static_rows = 0;
        for (int i = 0; i < m_n; i++) { cidx[0] = i; cval[0] = temporal; emit(1); m_static_rows++; }
        for (int r = 0, index = 0; r < rows; r++)
            for (int c = 0; c < cols; c++, index++)
            {
                int q = 1;
                if (c % 4 == 0 && r % 4 == 0) q = 3;
                else if ((c + r) % 2 != 1 && c != 0 && r != 0 && c != cols - 2 && r != rows - 2) continue;
                if (c >= cols - q || r >= rows - q) continue;
                const int i00 = 2 * index, i10 = i00 + 2 * q, i01 = 2 * (index + q * cols), i11 = i01 + 2 * q;
                const float w = local, w1 = (float)(v1 * w), w2 = (float)(v2 * w);
                auto row = [&](int a, float va, int b, float vb, int c2, float vc, int d, float vd) {
                    cidx[0] = a; cval[0] = va; cidx[1] = b; cval[1] = vb; cidx[2] = c2; cval[2] = vc; cidx[3] = d; cval[3] = vd;
                    emit(4); m_static_rows++;
                };
                row(i00, -w, i01, w, i01 + 1, -w2, i11 + 1, w2);
                row(i00 + 1, -w, i01, w2, i01 + 1, w, i11, -w2);
                row(i00, -w, i10, w, i10 + 1, -w1, i11 + 1, w1);
                row(i00 + 1, -w, i10, w1, i10 + 1, w, i11, -w1);
            }
    }

    bool generated() const { return m_n > 0; }
    int cols() const { return m_cols; }
    int rows() const { return m_rows; }
    void reset() { std::fill(m_mesh.begin(), m_mesh.end(), 0.0f); }

    int n() const { return m_n; }
    int hb() const { return m_hb; }
    int static_rows() const { return m_static_rows; }
    const std::vector<double>& static_band() const { return m_static; }     // lower band, column by column (see at())

private:
    // lower band, column by column: entry (i, j), j <= i <= j + hb, lives at B[j * (hb + 1) + (i - j)] -- the factorisation then walks
    // contiguous memory in its inner loop
    double& at(std::vector<double>& B, int i, int j) { return B[(size_t)j * (m_hb + 1) + (size_t)(i - j)]; }
    int m_cols = 0, m_rows = 0, m_n = 0, m_hb = 0, m_static_rows = 0;
    float m_ts = 0.0f;
    std::vector<float> m_mesh;
    std::vector<double> m_static;
};

} // namespace lvkh
