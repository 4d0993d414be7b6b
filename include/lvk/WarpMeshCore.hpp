// Warp-mesh arithmetic shared by the library's host logic (livevisionkit_amd/csrc/host_logic.hpp) and the C++ facade's lvk::WarpMesh
// (include/lvk/LiveVisionKit.hpp): ONE implementation, the one the parity tests hold to the oracle.
//
// Reference (paths relative to LiveVisionKit/): Math/WarpMesh.cpp:318-551 (set_identity, set_to, crop_in, clamp, combine, the operators),
// Math/Homography.cpp:125-130 (Homography::transform -> cv::perspectiveTransform).  Offsets are NORMALISED BACKWARD offsets, rows x cols x (dx, dy):
// offset = (identity - warped) / size ("the warp is specified backwards", WarpMesh.cpp:327).  Elementwise float ops follow OpenCV's scalar
// definitions: one rounding per op, no contraction -- the pragmas below keep a host compiler with FMA enabled (-march=native) from fusing them.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

#if defined(__clang__)
#define LVK_FP_CONTRACT_OFF _Pragma("clang fp contract(off)")
#else
#define LVK_FP_CONTRACT_OFF
#endif
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
#endif

namespace lvk { namespace detail {

class WarpMeshF
{
public:
    int rows = 2, cols = 2;
    std::vector<float> off;                                 // rows x cols x (dx, dy), normalised, "warp specified backwards"

    WarpMeshF() : off(8, 0.0f) {}
    WarpMeshF(int r, int c) : rows(r), cols(c), off((size_t)r * c * 2, 0.0f) {}

    void set_identity() { std::fill(off.begin(), off.end(), 0.0f); }
    void scale(float s) { LVK_FP_CONTRACT_OFF for (float& v : off) v = v * s; }                                                  // operator*=(float)
    void operator+=(const WarpMeshF& o) { for (size_t i = 0; i < off.size(); i++) off[i] = off[i] + o.off[i]; }
    void operator-=(const WarpMeshF& o) { for (size_t i = 0; i < off.size(); i++) off[i] = off[i] - o.off[i]; }
    void scale_add(const WarpMeshF& o, float a) { LVK_FP_CONTRACT_OFF for (size_t i = 0; i < off.size(); i++) off[i] = o.off[i] * a + off[i]; }   // combine(): cv::scaleAdd

    void clamp(float mx, float my)
    {
        for (size_t i = 0; i + 1 < off.size(); i += 2)
        {
            off[i] = std::min(std::max(off[i], -mx), mx);
            off[i + 1] = std::min(std::max(off[i + 1], -my), my);
        }
    }

    void crop_in(float x, float y, float w, float h)                                                         // WarpMesh.cpp:379-390
    {
        LVK_FP_CONTRACT_OFF
        const float kx = (w - 1.0f) / (float)(cols - 1), ky = (h - 1.0f) / (float)(rows - 1);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++)
            {
                float* p = &off[((size_t)r * cols + c) * 2];
                p[0] += (float)c * kx + x;
                p[1] += (float)r * ky + y;
            }
    }

    // WarpMesh::set_to(Homography, motion_scale) with Homography::transform -> cv::perspectiveTransform (double math)
    void from_homography(const double H[9], float sw, float sh)
    {
        LVK_FP_CONTRACT_OFF
        const float gx = sw / (float)(cols - 1), gy = sh / (float)(rows - 1);
        const float nx = 1.0f / sw, ny = 1.0f / sh;
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++)
            {
                const float px = (float)c * gx, py = (float)r * gy;
                double w = px * H[6] + py * H[7] + H[8];
                float qx = 0.0f, qy = 0.0f;
                if (std::fabs(w) > 1.1920928955078125e-07)
                {
                    w = 1. / w;
                    qx = (float)((px * H[0] + py * H[1] + H[2]) * w);
                    qy = (float)((px * H[3] + py * H[4] + H[5]) * w);
                }
                float* p = &off[((size_t)r * cols + c) * 2];
                p[0] = (px - qx) * nx;
                p[1] = (py - qy) * ny;
            }
    }
};

}} // namespace lvk::detail

#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC pop_options
#endif
