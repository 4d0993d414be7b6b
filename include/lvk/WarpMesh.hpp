// lvk::Homography, lvk::WarpMesh and the two lvk::remap launchers of the reference as part of the C++ facade (included by LiveVisionKit.hpp):
//   Math/Homography.hpp:25-110 / Homography.cpp      3 x 3 binary64 matrix, transform = cv::perspectiveTransform
//   Math/WarpMesh.hpp:31-140 / WarpMesh.cpp:34-551   mesh of NORMALISED BACKWARD offsets; the arithmetic SURVEY.md section 8 row a11 names, apply() (row a14)
//   Functions/Image.hpp:26-34 / Image.cpp:28-151      lvk::remap(src, dst, offset_map, background), lvk::remap(src, dst, homography, background, inverted) (row a15)
//   Functions/Drawing.hpp:23-71 / Drawing.tpp:53-196  colour constants, lvk::draw_grid, lvk::draw_crosses (section 8f row 4)
// These are what the plugin touches outside the stabilizer (LCFilter holds a WarpMesh: set_to(map) -> crop_in -> apply,
// Modules/OBS-Plugin/Sources/Enhancement/LCFilter.cpp:133-192).  The mesh arithmetic is lvk::detail::WarpMeshF -- the same code the library's host
// logic runs and the parity tests hold to the oracle (include/lvk/WarpMeshCore.hpp); apply() and remap() forward to the C-ABI.
//
// Differences a build without OpenCV implies: matrices are plain arrays (Homography::data() is `const double*`, row major; warp maps are
// `const float*` of rows x cols x 2), the device-resident offset map of remap() is lvk::OffsetMap (the reference: a CV_32FC2 cv::UMat).
// Third-party arithmetic restated from the published OpenCV 4.8 sources (absent from /root/reference, as for every delegated stage): cv::invert's closed
// form for 3 x 3 matrices (Homography::invert, remap(.., inverted = false)) and cv::gemm's k-ascending dot products (Homography::operator*=).
#pragma once

#include "WarpMeshCore.hpp"

namespace lvk {

// ---------------------------------------------------------------------------------------------- Math/Homography.hpp
class Homography
{
public:
    static const Homography& Zero() { static const Homography z(zero_tag{}); return z; }
    static const Homography& Identity() { static const Homography i; return i; }
    static Homography FromAffineMatrix(const double affine[6])                  // 2 x 3, row major (Homography.cpp:45-58)
    {
        Homography h;
        for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) h.m[r * 3 + c] = affine[r * 3 + c];
        return h;
    }

    Homography() { set_identity(); }                                            // "a default-initialised homography is identity"
    explicit Homography(const double matrix[9]) { for (int i = 0; i < 9; i++) m[i] = matrix[i]; }

    void set_zero() { for (double& v : m) v = 0.0; }
    void set_identity() { set_zero(); m[0] = m[4] = m[8] = 1.0; }

    // cv::perspectiveTransform (binary64 arithmetic whatever the point type; a vanishing denominator maps to the origin)
    cv::Point2d transform(const cv::Point2d& p) const
    {
        LVK_FP_CONTRACT_OFF
        double w = p.x * m[6] + p.y * m[7] + m[8];
        if (std::fabs(w) > 2.220446049250313e-16) { w = 1. / w; return {(p.x * m[0] + p.y * m[1] + m[2]) * w, (p.x * m[3] + p.y * m[4] + m[5]) * w}; }
        return {0.0, 0.0};
    }
    cv::Point2f transform(const cv::Point2f& p) const
    {
        LVK_FP_CONTRACT_OFF
        double w = p.x * m[6] + p.y * m[7] + m[8];
        if (std::fabs(w) > 1.1920928955078125e-07) { w = 1. / w; return {(float)((p.x * m[0] + p.y * m[1] + m[2]) * w), (float)((p.x * m[3] + p.y * m[4] + m[5]) * w)}; }
        return {0.0f, 0.0f};
    }
    cv::Point2d operator*(const cv::Point2d& p) const { return transform(p); }
    cv::Point2f operator*(const cv::Point2f& p) const { return transform(p); }
    void transform(const std::vector<cv::Point2f>& points, std::vector<cv::Point2f>& dst) const { dst.resize(points.size()); for (size_t i = 0; i < points.size(); i++) dst[i] = transform(points[i]); }
    void transform(const std::vector<cv::Point2d>& points, std::vector<cv::Point2d>& dst) const { dst.resize(points.size()); for (size_t i = 0; i < points.size(); i++) dst[i] = transform(points[i]); }
    std::vector<cv::Point2f> operator*(const std::vector<cv::Point2f>& points) const { std::vector<cv::Point2f> out; transform(points, out); return out; }
    std::vector<cv::Point2d> operator*(const std::vector<cv::Point2d>& points) const { std::vector<cv::Point2d> out; transform(points, out); return out; }

    const double* data() const { return m; }                                   // row major (the reference: const cv::Mat&, CV_64FC1)

    // cv::Mat::inv() -> cv::invert(DECOMP_LU), whose 3 x 3 case is the closed form d = 1 / det; t = adj * d (a singular matrix gives zero)
    Homography invert() const
    {
        LVK_FP_CONTRACT_OFF
        Homography r(zero_tag{});
        const double* s = m;
        double d = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) + s[2] * (s[3] * s[7] - s[4] * s[6]);
        if (d != 0.)
        {
            d = 1. / d;
            r.m[0] = (s[4] * s[8] - s[5] * s[7]) * d; r.m[1] = (s[2] * s[7] - s[1] * s[8]) * d; r.m[2] = (s[1] * s[5] - s[2] * s[4]) * d;
            r.m[3] = (s[5] * s[6] - s[3] * s[8]) * d; r.m[4] = (s[0] * s[8] - s[2] * s[6]) * d; r.m[5] = (s[2] * s[3] - s[0] * s[5]) * d;
            r.m[6] = (s[3] * s[7] - s[4] * s[6]) * d; r.m[7] = (s[1] * s[6] - s[0] * s[7]) * d; r.m[8] = (s[0] * s[4] - s[1] * s[3]) * d;
        }
        return r;
    }

    bool is_identity() const { return m[0] == 1.0 && m[1] == 0.0 && m[2] == 0.0 && m[3] == 0.0 && m[4] == 1.0 && m[5] == 0.0 && is_affine(); }
    bool is_affine() const { return m[6] == 0.0 && m[7] == 0.0 && m[8] == 1.0; }     // "the bottom row is unchanged from identity"
    bool is_zero() const { for (double v : m) if (v != 0.0) return false; return true; }

    void operator+=(const Homography& o) { for (int i = 0; i < 9; i++) m[i] = m[i] + o.m[i]; }
    void operator-=(const Homography& o) { for (int i = 0; i < 9; i++) m[i] = m[i] - o.m[i]; }
    void operator*=(const Homography& o)                                        // matrix product (cv::gemm: dot products with k ascending)
    {
        LVK_FP_CONTRACT_OFF
        double r[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
            {
                double acc = 0.0;
                for (int k = 0; k < 3; k++) acc = acc + m[i * 3 + k] * o.m[k * 3 + j];
                r[i * 3 + j] = acc;
            }
        for (int i = 0; i < 9; i++) m[i] = r[i];
    }
    void operator*=(const double scaling) { for (double& v : m) v = v * scaling; }
    void operator/=(const double scaling) { LVK_HIP_ASSERT(scaling != 0.0); for (double& v : m) v = v / scaling; }

private:
    struct zero_tag {};
    explicit Homography(zero_tag) { set_zero(); }
    double m[9];
};
inline Homography operator+(Homography a, const Homography& b) { a += b; return a; }
inline Homography operator-(Homography a, const Homography& b) { a -= b; return a; }
inline Homography operator*(Homography a, const Homography& b) { a *= b; return a; }
inline Homography operator*(Homography a, const double s) { a *= s; return a; }
inline Homography operator/(Homography a, const double s) { a /= s; return a; }

// ---------------------------------------------------------------------------------------------- device offset map (the CV_32FC2 cv::UMat of lvk::remap)
// rows x cols float2 offsets IN PIXELS resident in HBM: dst(x, y) samples src(x + off.x, y + off.y) (FSR.cl:362-403).
struct OffsetMap
{
    int cols = 0, rows = 0;
    size_t step = 0;
    bool empty() const { return !m_buf || cols == 0 || rows == 0; }
    cv::Size size() const { return {cols, rows}; }
    void create(const cv::Size& sz, const std::shared_ptr<hip::Context>& ctx = nullptr)
    {
        if (m_buf && m_buf.use_count() == 1 && cols == sz.width && rows == sz.height && (!ctx || ctx == m_ctx)) return;
        m_ctx = ctx ? ctx : (m_ctx ? m_ctx : hip::shared_context());
        void* p = nullptr;
        step = (size_t)sz.width * 2 * sizeof(float);
        m_ctx->check(lvk_hip_malloc(m_ctx->get(), step * (size_t)sz.height, &p), "OffsetMap::create");
        auto c = m_ctx;
        m_buf = std::shared_ptr<void>(p, [c](void* q) { lvk_hip_free(c->get(), q); });
        cols = sz.width; rows = sz.height;
    }
    void upload(const float* host_xy, const cv::Size& sz, const std::shared_ptr<hip::Context>& ctx = nullptr)      // tight rows x cols x 2 floats
    {
        create(sz, ctx);
        hip::ContextLock lock(m_ctx->mutex());
        m_ctx->check(lvk_hip_upload(m_ctx->get(), m_buf.get(), host_xy, step * (size_t)rows), "OffsetMap::upload");
        m_ctx->check(lvk_hip_sync(m_ctx->get()), "OffsetMap::upload");            // (the host array is the caller's again on return)
    }
    void* device_ptr() const { return m_buf.get(); }
    const std::shared_ptr<hip::Context>& context() const { return m_ctx; }

private:
    std::shared_ptr<void> m_buf;
    std::shared_ptr<hip::Context> m_ctx;
};

// ---------------------------------------------------------------------------------------------- Functions/Image.hpp: the two remap launchers
// lvk::remap(src, dst, offset_map, background) (Image.cpp:28-81 -> easu_remap, FSR.cl:362-403).  The reference sizes dst from the map; here the map
// has the size of src (the C-ABI's lvk_hip_remap_map; WarpMesh::apply and LCFilter never use anything else).
inline void remap(const VideoFrame& src, VideoFrame& dst, const OffsetMap& offset_map, const cv::Scalar& background = {0, 0, 0})
{
    LVK_HIP_ASSERT(src.cols > 0 && src.rows > 0 && !src.empty() && !offset_map.empty());
    LVK_HIP_ASSERT(offset_map.cols == src.cols && offset_map.rows == src.rows);
    const auto& ctx = src.context();
    if (offset_map.context() && offset_map.context() != ctx) ctx->wait_for(*offset_map.context());
    VideoFrame out;                                        // dst may be the object src refers to
    out.create(src.size(), CV_8UC3, ctx);
    const uint8_t bg[3] = {(uint8_t)background[0], (uint8_t)background[1], (uint8_t)background[2]};
    {
        hip::ContextLock lock(ctx->mutex());
        ctx->check(lvk_hip_remap_map(ctx->get(), src.device_ptr(), (int)src.step, src.rows, src.cols, out.device_ptr(), (int)out.step,
                                     offset_map.device_ptr(), (int)offset_map.step, bg, src.format == VideoFrame::YUV ? 1 : 0), "remap(offset_map)");
    }
    out.timestamp = dst.timestamp; out.format = dst.format;     // (the launcher leaves dst's metadata alone: WarpMesh::apply sets it, WarpMesh.cpp:220-222)
    dst = std::move(out);
}

// lvk::remap(src, dst, homography, background, inverted) (Image.cpp:85-151 -> easu_remap_homography, FSR.cl:407-452): `homography` maps dst -> src
// when inverted, else it is inverted first (homography.inv()); the kernel takes it cast to binary32 (Image.cpp:137-139).
inline void remap(const VideoFrame& src, VideoFrame& dst, const Homography& homography, const cv::Scalar& background = {0, 0, 0}, const bool inverted = false)
{
    LVK_HIP_ASSERT(src.cols > 0 && src.rows > 0 && !src.empty());
    const Homography t = inverted ? homography : homography.invert();
    float H[9];
    for (int i = 0; i < 9; i++) H[i] = (float)t.data()[i];
    const auto& ctx = src.context();
    VideoFrame out;
    out.create(src.size(), CV_8UC3, ctx);
    const uint8_t bg[3] = {(uint8_t)background[0], (uint8_t)background[1], (uint8_t)background[2]};
    {
        hip::ContextLock lock(ctx->mutex());
        ctx->check(lvk_hip_remap_homography(ctx->get(), src.device_ptr(), (int)src.step, src.rows, src.cols, out.device_ptr(), (int)out.step, out.rows, out.cols,
                                            0, 0, H, bg, src.format == VideoFrame::YUV ? 1 : 0), "remap(homography)");
    }
    out.timestamp = dst.timestamp; out.format = dst.format;
    dst = std::move(out);
}

// ---------------------------------------------------------------------------------------------- Math/WarpMesh.hpp
class WarpMesh
{
public:
    inline static const cv::Size MinimumSize = {2, 2};

    explicit WarpMesh(const cv::Size& size) : m(size.height, size.width) { LVK_HIP_ASSERT(size.height >= MinimumSize.height && size.width >= MinimumSize.width); }
    WarpMesh(const Homography& motion, const cv::Size2f& motion_scale, const cv::Size& size = MinimumSize) : WarpMesh(size) { set_to(motion, motion_scale); }
    // (warp_map: rows x cols x 2 floats, the reference's CV_32FC2 cv::Mat)
    WarpMesh(const float* warp_map, const cv::Size& size, const bool as_offsets, const bool normalized) : WarpMesh(size) { set_to(warp_map, size, as_offsets, normalized); }

    cv::Size size() const { return {m.cols, m.rows}; }
    int cols() const { return m.cols; }
    int rows() const { return m.rows; }
    float* offsets() { m_MapStale = true; return m.off.data(); }                // rows x cols x (dx, dy), normalised backward offsets
    const float* offsets() const { return m.off.data(); }

    // to_map (WarpMesh.cpp:159-168): offsets + identity grid of the mesh's own resolution
    void to_map(std::vector<float>& dst) const
    {
        dst.resize(m.off.size());
        for (int r = 0; r < m.rows; r++)
            for (int c = 0; c < m.cols; c++)
            {
                const size_t i = ((size_t)r * m.cols + c) * 2;
                dst[i] = m.off[i] + (float)c; dst[i + 1] = m.off[i + 1] + (float)r;
            }
    }
    void normalize(const cv::Size2f& motion_scale)                              // :172-180 (cv::multiply by the reciprocal)
    {
        m_MapStale = true;
        const float nx = 1.0f / motion_scale.width, ny = 1.0f / motion_scale.height;
        for (size_t i = 0; i + 1 < m.off.size(); i += 2) { m.off[i] = m.off[i] * nx; m.off[i + 1] = m.off[i + 1] * ny; }
    }

    void set_identity() { m_MapStale = true; m.set_identity(); }
    void set_to(const cv::Point2f& motion) { m_MapStale = true; for (size_t i = 0; i + 1 < m.off.size(); i += 2) { m.off[i] = -motion.x; m.off[i + 1] = -motion.y; } }      // "the warp is specified backwards"
    void set_to(const Homography& motion, const cv::Size2f& motion_scale) { m_MapStale = true; m.from_homography(motion.data(), motion_scale.width, motion_scale.height); }
    void set_to(const float* warp_map, const cv::Size& size, const bool as_offsets, const bool normalized)      // :345-365
    {
        m_MapStale = true;
        LVK_HIP_ASSERT(warp_map != nullptr && size.width >= 2 && size.height >= 2);
        m = detail::WarpMeshF(size.height, size.width);
        std::copy(warp_map, warp_map + m.off.size(), m.off.begin());
        if (!as_offsets)
            for (int r = 0; r < m.rows; r++)
                for (int c = 0; c < m.cols; c++) { const size_t i = ((size_t)r * m.cols + c) * 2; m.off[i] = m.off[i] - (float)c; m.off[i + 1] = m.off[i + 1] - (float)r; }
        if (!normalized) normalize(cv::Size2f((float)m.cols, (float)m.rows));
    }

    void scale(const cv::Size2f& scaling_factor)                                // :369-375
    {
        LVK_FP_CONTRACT_OFF
        m_MapStale = true;
        const float kx = ((1.0f / scaling_factor.width) - 1.0f) / (float)(m.cols - 1), ky = ((1.0f / scaling_factor.height) - 1.0f) / (float)(m.rows - 1);
        for (int r = 0; r < m.rows; r++)
            for (int c = 0; c < m.cols; c++) { const size_t i = ((size_t)r * m.cols + c) * 2; m.off[i] += (float)c * kx; m.off[i + 1] += (float)r * ky; }
    }
    void crop_in(const cv::Rect2f& region)                                      // :379-390
    {
        m_MapStale = true;
        LVK_HIP_ASSERT(region.width >= 0 && region.width <= (float)cols() && region.height >= 0 && region.height <= (float)rows() && region.x >= 0 && region.y >= 0);
        m.crop_in(region.x, region.y, region.width, region.height);
    }
    void clamp(const cv::Size2f& magnitude) { m_MapStale = true; m.clamp(magnitude.width, magnitude.height); }                     // :411-417
    void clamp(const cv::Size2f& min, const cv::Size2f& max)                    // :421-427
    {
        m_MapStale = true;
        for (size_t i = 0; i + 1 < m.off.size(); i += 2)
        {
            m.off[i] = std::min(std::max(m.off[i], min.width), max.width);
            m.off[i + 1] = std::min(std::max(m.off[i + 1], min.height), max.height);
        }
    }
    void combine(const WarpMesh& mesh, const float scaling = 1.0f) { LVK_HIP_ASSERT(size() == mesh.size()); m_MapStale = true; m.scale_add(mesh.m, scaling); }      // cv::scaleAdd, :445-448

    void read(const std::function<void(const cv::Point2f& offset, const cv::Point& coord)>& operation, const bool /*parallel*/ = true) const      // :264-287
    {
        for (int r = 0; r < m.rows; r++)
            for (int c = 0; c < m.cols; c++) { const size_t i = ((size_t)r * m.cols + c) * 2; operation(cv::Point2f(m.off[i], m.off[i + 1]), cv::Point(c, r)); }
    }
    void write(const std::function<void(cv::Point2f& offset, const cv::Point& coord)>& operation, const bool /*parallel*/ = true)                 // :291-314
    {
        m_MapStale = true;
        for (int r = 0; r < m.rows; r++)
            for (int c = 0; c < m.cols; c++)
            {
                const size_t i = ((size_t)r * m.cols + c) * 2;
                cv::Point2f v(m.off[i], m.off[i + 1]);
                operation(v, cv::Point(c, r));
                m.off[i] = v.x; m.off[i + 1] = v.y;
            }
    }

    void operator+=(const WarpMesh& other) { LVK_HIP_ASSERT(size() == other.size()); m_MapStale = true; m += other.m; }
    void operator-=(const WarpMesh& other) { LVK_HIP_ASSERT(size() == other.size()); m_MapStale = true; m -= other.m; }
    void operator*=(const WarpMesh& other) { LVK_HIP_ASSERT(size() == other.size()); m_MapStale = true; for (size_t i = 0; i < m.off.size(); i++) m.off[i] = m.off[i] * other.m.off[i]; }
    void operator+=(const cv::Point2f& offset) { m_MapStale = true; for (size_t i = 0; i + 1 < m.off.size(); i += 2) { m.off[i] = m.off[i] + offset.x; m.off[i + 1] = m.off[i + 1] + offset.y; } }
    void operator-=(const cv::Point2f& offset) { m_MapStale = true; for (size_t i = 0; i + 1 < m.off.size(); i += 2) { m.off[i] = m.off[i] - offset.x; m.off[i + 1] = m.off[i + 1] - offset.y; } }
    void operator*=(const cv::Size2f& scaling) { m_MapStale = true; for (size_t i = 0; i + 1 < m.off.size(); i += 2) { m.off[i] = m.off[i] * scaling.width; m.off[i + 1] = m.off[i + 1] * scaling.height; } }
    void operator/=(const cv::Size2f& scaling)
    {
        m_MapStale = true;
        LVK_HIP_ASSERT(scaling.width != 0.0f && scaling.height != 0.0f);
        for (size_t i = 0; i + 1 < m.off.size(); i += 2) { m.off[i] = m.off[i] / scaling.width; m.off[i + 1] = m.off[i + 1] / scaling.height; }
    }
    void operator*=(const float scaling) { m_MapStale = true; m.scale(scaling); }
    void operator/=(const float scaling) { LVK_HIP_ASSERT(scaling != 0.0f); m_MapStale = true; for (float& v : m.off) v = v / scaling; }

    // WarpMesh::apply (WarpMesh.cpp:183-223).  2 x 2: cv::getPerspectiveTransform of the displaced corners + the homography kernel; larger meshes: the
    // reference resizes the offsets to the frame (INTER_LINEAR_EXACT), multiplies by (W, H) and remaps through the map -- here the same interpolation runs
    // INSIDE the kernel from the mesh vertices (no W x H map is materialised; lvk_hip_warpmesh_apply, bit-identical, tests/test_remap_gpu.py), except for
    // a mesh that HAS the frame's size (LCFilter's correction map): there the resize is the identity and the offsets x (W, H) go up as the map.
    // Meshes between 64 KB and the frame size are not taken.  Copies timestamp and format.
    void apply(const VideoFrame& src, VideoFrame& dst, const cv::Scalar& background = {0, 0, 0}) const
    {
        LVK_HIP_ASSERT(!src.empty());
        const uint8_t bg[3] = {(uint8_t)background[0], (uint8_t)background[1], (uint8_t)background[2]};
        const auto& ctx = src.context();
        VideoFrame out;                                        // dst may be the object src refers to (LCFilter swaps, VSFilter aliases)
        if (m.cols == src.cols && m.rows == src.rows && (size_t)m.cols * m.rows * 8 > 65536)
        {
            // (the map is rebuilt and uploaded when the mesh has changed -- LCFilter applies ONE mesh to every frame: 16.6 MB at 1080p that the
            //  reference's cv::resize + cv::multiply into m_WarpMap redo per frame)
            if (m_MapStale || m_Map.empty() || m_Map.size() != src.size() || m_Map.context() != ctx)
            {
                const float w = (float)src.cols, h = (float)src.rows;
                std::vector<float> map(m.off.size());
                for (size_t i = 0; i + 1 < map.size(); i += 2) { map[i] = m.off[i] * w; map[i + 1] = m.off[i + 1] * h; }
                m_Map.upload(map.data(), src.size(), ctx);
                m_MapStale = false;
            }
            out.timestamp = src.timestamp; out.format = src.format;
            remap(src, out, m_Map, background);
        }
        else
        {
            out.create(src.size(), CV_8UC3, ctx);
            hip::ContextLock lock(ctx->mutex());
            ctx->check(lvk_hip_warpmesh_apply(ctx->get(), src.device_ptr(), (int)src.step, src.rows, src.cols, out.device_ptr(), (int)out.step,
                                              m.off.data(), m.rows, m.cols, bg, src.format == VideoFrame::YUV ? 1 : 0), "WarpMesh::apply");
        }
        out.timestamp = src.timestamp; out.format = src.format;
        dst = std::move(out);
    }

private:
    detail::WarpMeshF m;
    mutable OffsetMap m_Map;                                                   // (the reference's m_WarpMap)
    mutable bool m_MapStale = true;                                            // the mesh changed since m_Map was made
};
inline WarpMesh operator+(WarpMesh a, const WarpMesh& b) { a += b; return a; }
inline WarpMesh operator-(WarpMesh a, const WarpMesh& b) { a -= b; return a; }
inline WarpMesh operator*(WarpMesh a, const float s) { a *= s; return a; }
inline WarpMesh operator/(WarpMesh a, const float s) { a /= s; return a; }

// ---------------------------------------------------------------------------------------------- Functions/Drawing.hpp
// Colour constants (Drawing.hpp:23-71; lvk::col::X[frame.format]) and the two overlay launchers the stabilizer's test mode and LCFilter's test grid use:
// lvk::draw_grid (Drawing.tpp:53-93, kernel `grid`) and lvk::draw_crosses (Drawing.tpp:146-196, kernel `crosses`; the points are scaled by
// coord_scaling and rounded to pixels like cv::multiply(.., CV_32S)).  In place on a packed 8UC3 device frame, asynchronous on the frame's context.
namespace rgb { const cv::Scalar BLACK(0, 0, 0), WHITE(255, 255, 255), MAGENTA(255, 0, 255), GREEN(0, 255, 0), BLUE(0, 0, 255), RED(255, 0, 0); }
namespace bgr { const cv::Scalar BLACK(0, 0, 0), WHITE(255, 255, 255), MAGENTA(255, 0, 255), GREEN(0, 255, 0), BLUE(255, 0, 0), RED(0, 0, 255); }
namespace yuv { const cv::Scalar BLACK(0, 128, 128), WHITE(255, 0, 0), MAGENTA(105, 212, 234), GREEN(149, 43, 21), BLUE(29, 255, 107), RED(76, 84, 255); }
namespace gray { const cv::Scalar BLACK(0), WHITE(255), MAGENTA(105), GREEN(149), BLUE(29), RED(76); }
namespace col
{
    // Formats: BGR, BGRA, RGB, RGBA, YUV, GRAY
    const cv::Scalar BLACK[] = {bgr::BLACK, bgr::BLACK, rgb::BLACK, rgb::BLACK, yuv::BLACK, gray::BLACK};
    const cv::Scalar WHITE[] = {bgr::WHITE, bgr::WHITE, rgb::WHITE, rgb::WHITE, yuv::WHITE, gray::WHITE};
    const cv::Scalar MAGENTA[] = {bgr::MAGENTA, bgr::MAGENTA, rgb::MAGENTA, rgb::MAGENTA, yuv::MAGENTA, gray::MAGENTA};
    const cv::Scalar GREEN[] = {bgr::GREEN, bgr::GREEN, rgb::GREEN, rgb::GREEN, yuv::GREEN, gray::GREEN};
    const cv::Scalar BLUE[] = {bgr::BLUE, bgr::BLUE, rgb::BLUE, rgb::BLUE, yuv::BLUE, gray::BLUE};
    const cv::Scalar RED[] = {bgr::RED, bgr::RED, rgb::RED, rgb::RED, yuv::RED, gray::RED};
}

inline void draw_grid(VideoFrame& dst, const cv::Size& grid, const cv::Scalar& color, const int thickness)
{
    LVK_HIP_ASSERT(thickness >= 1 && !dst.empty() && grid.width >= 1 && grid.height >= 1);
    const uint8_t c[3] = {(uint8_t)color[0], (uint8_t)color[1], (uint8_t)color[2]};
    const auto& ctx = dst.context();
    hip::ContextLock lock(ctx->mutex());
    ctx->check(lvk_hip_draw_grid(ctx->get(), dst.device_ptr(), (int)dst.step, dst.rows, dst.cols, grid.width, grid.height, c, thickness), "draw_grid");
}

template <typename T>
inline void draw_crosses(VideoFrame& dst, const std::vector<cv::Point_<T>>& points, const cv::Scalar& color, const int32_t cross_size,
                         const int32_t cross_thickness, const cv::Size2f& coord_scaling = {1.0f, 1.0f})
{
    LVK_HIP_ASSERT(coord_scaling.width >= 0 && coord_scaling.height >= 0 && cross_thickness >= 1 && cross_size >= 1 && !dst.empty());
    if (points.empty()) return;
    std::vector<float> xy(points.size() * 2);
    for (size_t i = 0; i < points.size(); i++) { xy[2 * i] = (float)points[i].x; xy[2 * i + 1] = (float)points[i].y; }
    const uint8_t c[3] = {(uint8_t)color[0], (uint8_t)color[1], (uint8_t)color[2]};
    const auto& ctx = dst.context();
    hip::ContextLock lock(ctx->mutex());
    ctx->check(lvk_hip_draw_crosses(ctx->get(), dst.device_ptr(), (int)dst.step, dst.rows, dst.cols, xy.data(), (int)points.size(),
                                    coord_scaling.width, coord_scaling.height, c, cross_size, cross_thickness), "draw_crosses");
}

} // namespace lvk
