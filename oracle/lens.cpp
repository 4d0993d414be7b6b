// ORACLE (test infrastructure only -- see lvk_oracle.h).
// CPU restatement of the lens-correction warp of the OBS plugin's LCFilter (reference:
// Modules/OBS-Plugin/Sources/Enhancement/LCFilter.cpp:133-192): cv::getOptimalNewCameraMatrix(alpha = 0) +
// cv::initUndistortRectifyMap -> full-resolution WarpMesh (set_to(map, as_offsets=false, normalized=false), crop_in(view
// region)) -> WarpMesh::apply == lvk::remap(src, dst, offset_map) with easu_remap (Functions/OpenCL/Sources/FSR.cl:362-403).
// The calib3d arithmetic follows OpenCV 4.8.0 (calibration.cpp cvGetOptimalNewCameraMatrix / icvGetRectangles,
// undistort.dispatch.cpp cvUndistortPointsInternal with 5 fixed iterations, initUndistortRectifyMap with its incremental
// `_x += ir[0]` row walk); sources not in /root/reference.  Camera profile = (fx, fy, cx, cy, k1, k2, p1, p2, k3)
// (Modules/OBS-Plugin/Sources/Tools/CCTool.cpp:120-153).  All binary64 except where OpenCV stores float.
#include "lvk_oracle.h"

#include <cmath>
#include <cfloat>
#include <vector>
#include <algorithm>

namespace {

struct Cam { double fx, fy, cx, cy, k1, k2, p1, p2, k3; };

// cvUndistortPointsInternal (R = identity, optional new camera matrix P), criteria COUNT 5
void undistort_point(const Cam& c, double u, double v, const double* P /* fx', fy', cx', cy' or null */, float& ox, float& oy)
{
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    double x = (u - c.cx) * ifx, y = (v - c.cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++)
    {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
        if (icdist < 0) { x = (u - c.cx) * ifx; y = (v - c.cy) * ify; break; }
        const double deltaX = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
        const double deltaY = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y + 0 * r2 + 0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    double xx = x, yy = y;                          // R = identity: xx = x / 1, yy = y / 1
    if (P) { xx = xx * P[0] + P[2]; yy = yy * P[1] + P[3]; }
    ox = (float)xx; oy = (float)yy;
}

// icvGetRectangles: 9x9 grid of undistorted points -> inscribed / circumscribed rectangles (float)
void get_rectangles(const Cam& c, const double* P, int w, int h, float inner[4], float outer[4])
{
    const int N = 9;
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
    float oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++)
        {
            const float px = (float)x * w / (N - 1), py = (float)y * h / (N - 1);
            float ux, uy;
            undistort_point(c, px, py, P, ux, uy);
            oX0 = std::min(oX0, ux); oX1 = std::max(oX1, ux); oY0 = std::min(oY0, uy); oY1 = std::max(oY1, uy);
            if (x == 0) iX0 = std::max(iX0, ux);
            if (x == N - 1) iX1 = std::min(iX1, ux);
            if (y == 0) iY0 = std::max(iY0, uy);
            if (y == N - 1) iY1 = std::min(iY1, uy);
        }
    inner[0] = iX0; inner[1] = iY0; inner[2] = iX1 - iX0; inner[3] = iY1 - iY0;
    outer[0] = oX0; outer[1] = oY0; outer[2] = oX1 - oX0; outer[3] = oY1 - oY0;
}

inline int round_sat(double v) { return (int)lrint(v); }

} // namespace

namespace {

// cvGetOptimalNewCameraMatrix(alpha = 0, newImgSize = imgSize, centerPrincipalPoint = false) + valid-pixel ROI
void optimal_matrix(const Cam& c, int rows, int cols, double P[4], int view_xywh[4])
{
    float inner[4], outer[4];
    get_rectangles(c, nullptr, cols, rows, inner, outer);
    const double fx0 = (cols - 1) / (double)inner[2], fy0 = (rows - 1) / (double)inner[3];
    const double cx0 = -fx0 * inner[0], cy0 = -fy0 * inner[1];
    const double fx1 = (cols - 1) / (double)outer[2], fy1 = (rows - 1) / (double)outer[3];
    const double cx1 = -fx1 * outer[0], cy1 = -fy1 * outer[1];
    const double alpha = 0.0;
    P[0] = fx0 * (1 - alpha) + fx1 * alpha; P[1] = fy0 * (1 - alpha) + fy1 * alpha;
    P[2] = cx0 * (1 - alpha) + cx1 * alpha; P[3] = cy0 * (1 - alpha) + cy1 * alpha;
    float in2[4], out2[4];
    get_rectangles(c, P, cols, rows, in2, out2);
    // cv::Rect r = inner (Rect_<float> -> Rect: members rounded); r &= Rect(0, 0, w, h)
    int rx = round_sat(in2[0]), ry = round_sat(in2[1]), rw = round_sat(in2[2]), rh = round_sat(in2[3]);
    const int x1 = std::max(rx, 0), y1 = std::max(ry, 0), x2 = std::min(rx + rw, cols), y2 = std::min(ry + rh, rows);
    view_xywh[0] = x1; view_xywh[1] = y1; view_xywh[2] = std::max(x2 - x1, 0); view_xywh[3] = std::max(y2 - y1, 0);
    if (view_xywh[2] <= 0 || view_xywh[3] <= 0) { view_xywh[0] = view_xywh[1] = view_xywh[2] = view_xywh[3] = 0; }
}

} // namespace

extern "C" {

// LCFilter::prepare_undistort_maps (LCFilter.cpp:133-171).  params = fx, fy, cx, cy, k1, k2, p1, p2, k3.
// offsets: rows x cols x 2 floats = the per-pixel offset map WarpMesh::apply hands to lvk::remap (in pixels);
// view_xywh: the valid-pixel ROI of getOptimalNewCameraMatrix.
int lvko_lens_offset_map(const double params[9], int rows, int cols, float* offsets, int view_xywh[4])
{
    if (!params || !offsets || rows <= 1 || cols <= 1) return -1;
    const Cam c{params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7], params[8]};
    double P[4];
    optimal_matrix(c, rows, cols, P, view_xywh);
    // initUndistortRectifyMap(K, D, R = I, newK = P, size, CV_32FC2): ir = inverse(newK)
    const double ir[9] = {1. / P[0], 0, -P[2] / P[0], 0, 1. / P[1], -P[3] / P[1], 0, 0, 1};
    // WarpMesh::set_to(map, false, false): offsets = map - identity grid, then * (1/cols, 1/rows); crop_in(norm view region);
    // WarpMesh::apply: resize to the same size (copy), * (cols, rows)
    const float nfx = 1.0f / (float)cols, nfy = 1.0f / (float)rows;
    const float vrx = (float)view_xywh[0] / (float)cols, vry = (float)view_xywh[1] / (float)rows;
    const float vrw = (float)view_xywh[2] / (float)cols, vrh = (float)view_xywh[3] / (float)rows;
    const float csx = (vrw - 1.0f) / (float)(cols - 1), csy = (vrh - 1.0f) / (float)(rows - 1);
    for (int i = 0; i < rows; i++)
    {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < cols; j++, _x += ir[0], _y += ir[3], _w += ir[6])
        {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
            const double xd = (x * kr + c.p1 * _2xy + c.p2 * (r2 + 2 * x2) + 0 * r2 + 0 * r2 * r2);
            const double yd = (y * kr + c.p1 * (r2 + 2 * y2) + c.p2 * _2xy + 0 * r2 + 0 * r2 * r2);
            const float mu = (float)(c.fx * xd + c.cx), mv = (float)(c.fy * yd + c.cy);
            float ox = (mu - (float)j) * nfx, oy = (mv - (float)i) * nfy;                 // set_to(..., false, false)
            ox = ox + ((float)j * csx + vrx); oy = oy + ((float)i * csy + vry);             // crop_in
            offsets[((size_t)i * cols + j) * 2] = ox * (float)cols;                       // apply(): * motion_scaling
            offsets[((size_t)i * cols + j) * 2 + 1] = oy * (float)rows;
        }
    }
    return 0;
}

// ---- fused lens model (this repo's design for BASELINE config 5) ---------------------------
// The same warp as lvko_lens_offset_map, but evaluated in closed form at a fractional position of the corrected frame so
// that it can be composed with the stabilizing warp (one resampling instead of the reference chain's two):
//   F(u, v) = K * distort(P^-1 (u, v)) + crop_in term (u * kxc + vxc, v * kyc + vyc).
// model[17] = nfx, nfy, ncx, ncy (new camera matrix P), fx, fy, cx, cy, k1, k2, p1, p2, k3, kxc, vxc, kyc, vyc.
int lvko_lens_model(const double params[9], int rows, int cols, double model[17])
{
    if (!params || !model || rows <= 1 || cols <= 1) return -1;
    const Cam c{params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7], params[8]};
    double P[4]; int view[4];
    optimal_matrix(c, rows, cols, P, view);
    const float vrx = (float)view[0] / (float)cols, vry = (float)view[1] / (float)rows;
    const float vrw = (float)view[2] / (float)cols, vrh = (float)view[3] / (float)rows;
    const float csx = (vrw - 1.0f) / (float)(cols - 1), csy = (vrh - 1.0f) / (float)(rows - 1);
    model[0] = P[0]; model[1] = P[1]; model[2] = P[2]; model[3] = P[3];
    for (int i = 0; i < 9; i++) model[4 + i] = params[i];
    model[13] = (double)csx * cols; model[14] = (double)vrx * cols;
    model[15] = (double)csy * rows; model[16] = (double)vry * rows;
    return 0;
}

// F^-1 for tracked points: a point of the RAW tracking frame (scale sx, sy = frame / tracking resolution) -> the same
// point of the lens-corrected tracking frame, binary64: two passes of {remove the crop_in term, 5 undistortPoints
// iterations, apply P}.  out may alias pts.
void lvko_lens_undistort_points(const double model[17], double sx, double sy, const float* pts, int n, float* out)
{
    const double nfx = model[0], nfy = model[1], ncx = model[2], ncy = model[3];
    const double fx = model[4], fy = model[5], cx = model[6], cy = model[7];
    const double k1 = model[8], k2 = model[9], p1 = model[10], p2 = model[11], k3 = model[12];
    const double kxc = model[13], vxc = model[14], kyc = model[15], vyc = model[16];
    for (int i = 0; i < n; i++)
    {
        const double s = (double)pts[2 * i] * sx, t = (double)pts[2 * i + 1] * sy;
        double u = s, v = t;
        for (int pass = 0; pass < 2; pass++)
        {
            const double s1 = s - (u * kxc + vxc), t1 = t - (v * kyc + vyc);
            const double x0 = (s1 - cx) / fx, y0 = (t1 - cy) / fy;
            double x = x0, y = y0;
            for (int j = 0; j < 5; j++)
            {
                const double r2 = x * x + y * y;
                const double icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
                if (icdist < 0) { x = x0; y = y0; break; }
                const double dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
                const double dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
                x = (x0 - dX) * icdist;
                y = (y0 - dY) * icdist;
            }
            u = x * nfx + ncx; v = y * nfy + ncy;
        }
        out[2 * i] = (float)(u / sx); out[2 * i + 1] = (float)(v / sy);
    }
}

} // extern "C"
