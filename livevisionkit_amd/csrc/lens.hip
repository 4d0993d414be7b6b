// Lens-correction warp of the OBS plugin's LCFilter for gfx950: the undistortion map is a
// static function of the camera profile and the frame size, so it is evaluated ONCE on the host (binary64, OpenCV 4.8.0
// calib3d arithmetic) into a per-pixel float2 offset map that stays resident in HBM; the per-frame work is k_remap_map
// (remap.hip), the same EASU resampler the stabilizer uses.
//
// Replaces LCFilter::prepare_undistort_maps + LCFilter::filter (reference:
// Modules/OBS-Plugin/Sources/Enhancement/LCFilter.cpp:133-192): cv::getOptimalNewCameraMatrix(alpha 0, valid ROI) ->
// cv::initUndistortRectifyMap(CV_32FC2) -> WarpMesh::set_to(map, false, false) -> crop_in(view region) -> apply().
// Camera profile fields as the plugin stores them (Modules/OBS-Plugin/Sources/Tools/CCTool.cpp:120-153).
#include "lvk_hip_internal.hpp"

#include <cfloat>
#include <cmath>
#include <vector>

namespace {

// cvUndistortPointsInternal: 5 fixed-point iterations of the inverse Brown-Conrady model, then the new camera matrix
void undistort_one(const lvk_camera_params& c, double u, double v, const double* P, float& ox, float& oy)
{
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    double x = (u - c.cx) * ifx, y = (v - c.cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++)
    {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
        if (icdist < 0) { x = (u - c.cx) * ifx; y = (v - c.cy) * ify; break; }
        const double dX = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
        const double dY = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y + 0 * r2 + 0 * r2 * r2;
        x = (x0 - dX) * icdist;
        y = (y0 - dY) * icdist;
    }
    if (P) { x = x * P[0] + P[2]; y = y * P[1] + P[3]; }
    ox = (float)x; oy = (float)y;
}

// icvGetRectangles: inscribed rectangle of a 9 x 9 grid of undistorted border-to-border samples
void inscribed(const lvk_camera_params& c, const double* P, int w, int h, float in[4], float out[4])
{
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX, oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    for (int y = 0; y < 9; y++)
        for (int x = 0; x < 9; x++)
        {
            float px, py;
            undistort_one(c, (float)x * w / 8, (float)y * h / 8, P, px, py);
            oX0 = std::min(oX0, px); oX1 = std::max(oX1, px); oY0 = std::min(oY0, py); oY1 = std::max(oY1, py);
            if (x == 0) iX0 = std::max(iX0, px);
            if (x == 8) iX1 = std::min(iX1, px);
            if (y == 0) iY0 = std::max(iY0, py);
            if (y == 8) iY1 = std::min(iY1, py);
        }
    in[0] = iX0; in[1] = iY0; in[2] = iX1 - iX0; in[3] = iY1 - iY0;
    out[0] = oX0; out[1] = oY0; out[2] = oX1 - oX0; out[3] = oY1 - oY0;
}

// cv::getOptimalNewCameraMatrix(alpha 0, same size) -> P = (fx', fy', cx', cy') and the valid-pixel ROI
void optimal_matrix(const lvk_camera_params& c, int rows, int cols, double P[4], int view[4])
{
    float in[4], out[4];
    inscribed(c, nullptr, cols, rows, in, out);
    const double fx0 = (cols - 1) / (double)in[2], fy0 = (rows - 1) / (double)in[3], cx0 = -fx0 * in[0], cy0 = -fy0 * in[1];
    const double fx1 = (cols - 1) / (double)out[2], fy1 = (rows - 1) / (double)out[3], cx1 = -fx1 * out[0], cy1 = -fy1 * out[1];
    const double alpha = 0.0;
    P[0] = fx0 * (1 - alpha) + fx1 * alpha; P[1] = fy0 * (1 - alpha) + fy1 * alpha;
    P[2] = cx0 * (1 - alpha) + cx1 * alpha; P[3] = cy0 * (1 - alpha) + cy1 * alpha;
    inscribed(c, P, cols, rows, in, out);
    const int rx = (int)lrint(in[0]), ry = (int)lrint(in[1]), rw = (int)lrint(in[2]), rh = (int)lrint(in[3]);
    const int x1 = std::max(rx, 0), y1 = std::max(ry, 0), x2 = std::min(rx + rw, cols), y2 = std::min(ry + rh, rows);
    view[0] = x1; view[1] = y1; view[2] = std::max(x2 - x1, 0); view[3] = std::max(y2 - y1, 0);
    if (view[2] <= 0 || view[3] <= 0) view[0] = view[1] = view[2] = view[3] = 0;
}

void build_offsets(const lvk_camera_params& c, int rows, int cols, std::vector<float>& off, int view[4])
{
    double P[4];
    optimal_matrix(c, rows, cols, P, view);
    const double ir[9] = {1. / P[0], 0, -P[2] / P[0], 0, 1. / P[1], -P[3] / P[1], 0, 0, 1};
    const float nfx = 1.0f / (float)cols, nfy = 1.0f / (float)rows;
    const float vx = (float)view[0] / (float)cols, vy = (float)view[1] / (float)rows;
    const float vw = (float)view[2] / (float)cols, vh = (float)view[3] / (float)rows;
    const float kx = (vw - 1.0f) / (float)(cols - 1), ky = (vh - 1.0f) / (float)(rows - 1);
    off.resize((size_t)rows * cols * 2);
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
            float ox = (mu - (float)j) * nfx, oy = (mv - (float)i) * nfy;       // WarpMesh::set_to(map, as_offsets=false, normalized=false)
            ox = ox + ((float)j * kx + vx); oy = oy + ((float)i * ky + vy);       // WarpMesh::crop_in(norm_view_region)
            off[((size_t)i * cols + j) * 2] = ox * (float)cols;                  // WarpMesh::apply: * (cols, rows)
            off[((size_t)i * cols + j) * 2 + 1] = oy * (float)rows;
        }
    }
}

// One thread per point (lvk_lens_undistort_point, lvk_hip_internal.hpp)
__global__ void k_lens_undistort(LensModelD M, double sx, double sy, const float2* __restrict__ a, int na,
                                 const float2* __restrict__ b, int nb, float2* __restrict__ out)
{
    LVK_TRACKER_PRIORITY();                        // part of the tracker chain in the fused lens mode
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= na + nb) return;
    out[i] = lvk_lens_undistort_point(M, sx, sy, i < na ? a[i] : b[i - na]);
}

} // namespace

int lvk_lens_model_build(const lvk_camera_params& c, int rows, int cols, LensModel& out)
{
    if (rows <= 1 || cols <= 1 || c.fx == 0.0 || c.fy == 0.0) return LVK_HIP_ERR_ARG;
    double P[4];
    optimal_matrix(c, rows, cols, P, out.view);
    const float vx = (float)out.view[0] / (float)cols, vy = (float)out.view[1] / (float)rows;
    const float vw = (float)out.view[2] / (float)cols, vh = (float)out.view[3] / (float)rows;
    const float kx = (vw - 1.0f) / (float)(cols - 1), ky = (vh - 1.0f) / (float)(rows - 1);
    double* d = out.d;
    d[0] = P[0]; d[1] = P[1]; d[2] = P[2]; d[3] = P[3];
    d[4] = c.fx; d[5] = c.fy; d[6] = c.cx; d[7] = c.cy; d[8] = c.k1; d[9] = c.k2; d[10] = c.p1; d[11] = c.p2; d[12] = c.k3;
    d[13] = (double)kx * cols; d[14] = (double)vx * cols; d[15] = (double)ky * rows; d[16] = (double)vy * rows;
    out.f[0] = (float)(1.0 / d[0]); out.f[1] = (float)(1.0 / d[1]);
    for (int i = 2; i < 17; i++) out.f[i] = (float)d[i];
    return LVK_HIP_OK;
}

int lvk_launch_lens_undistort(lvk_hip_ctx* ctx, hipStream_t stream, const LensModel& model, double sx, double sy,
                              const float2* a, int na, const float2* b, int nb, float2* out)
{
    if (na + nb <= 0) return LVK_HIP_OK;
    LensModelD M;
    for (int i = 0; i < 17; i++) M.d[i] = model.d[i];
    hipLaunchKernelGGL(k_lens_undistort, dim3((unsigned)((na + nb + 255) / 256)), dim3(256), 0, stream, M, sx, sy, a, na, b, nb, out);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

extern "C" {

// test / tooling entry: corrected positions of n raw points (host arrays)
int lvk_hip_lens_undistort_points(lvk_hip_ctx* ctx, const lvk_camera_params* params, int rows, int cols, double sx, double sy,
                                  const float* pts, int n, float* out)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, params && pts && out && n >= 0);
    if (n == 0) return LVK_HIP_OK;
    LensModel m;
    if (lvk_lens_model_build(*params, rows, cols, m) != LVK_HIP_OK) return ctx->fail(LVK_HIP_ERR_ARG, "invalid camera profile");
    float2 *d_in = nullptr, *d_out = nullptr;
    LVK_HIP_CHECK(ctx, hipMalloc(&d_in, 2 * (size_t)n * sizeof(float2)));
    d_out = d_in + n;
    int rc = LVK_HIP_OK;
    hipError_t e = hipMemcpyAsync(d_in, pts, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = lvk_launch_lens_undistort(ctx, ctx->stream, m, sx, sy, d_in, n, nullptr, 0, d_out);
    if (e == hipSuccess && rc == LVK_HIP_OK) e = hipMemcpyAsync(out, d_out, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    if (e != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
    return rc;
}


int lvk_hip_lens_map_create(lvk_hip_ctx* ctx, const lvk_camera_params* params, int rows, int cols, void** d_map, int view_xywh[4])
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, params && d_map && rows > 1 && cols > 1 && params->fx != 0.0 && params->fy != 0.0);
    std::vector<float> off; int view[4];
    build_offsets(*params, rows, cols, off, view);
    void* d = nullptr;
    LVK_HIP_CHECK(ctx, hipMalloc(&d, off.size() * sizeof(float)));
    hipError_t e = hipMemcpy(d, off.data(), off.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
    *d_map = d;
    if (view_xywh) for (int i = 0; i < 4; i++) view_xywh[i] = view[i];
    return LVK_HIP_OK;
}

int lvk_hip_lens_map_destroy(lvk_hip_ctx* ctx, void* d_map)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    LVK_HIP_CHECK(ctx, hipFree(d_map));
    return LVK_HIP_OK;
}

} // extern "C"
