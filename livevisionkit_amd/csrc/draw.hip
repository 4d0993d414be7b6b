// Debug overlays of the stabilization filter for gfx950: the motion-mesh grid and the tracker
// crosses the OBS plugin's test mode draws into the newest queued frame.
//
// Replaces lvk::draw_grid / lvk::draw_crosses (reference: LiveVisionKit/Functions/Drawing.tpp:53-93,146-196) and their
// kernels `grid` / `crosses` (Functions/OpenCL/Sources/Drawing.cl:22-39,75-105), plus StabilizationFilter::draw_trackers /
// draw_motion_mesh (Filters/StabilizationFilter.cpp:163-188) and FrameTracker::draw_trackers (Vision/FrameTracker.cpp:489-505).
// Pure integer / exact-fmod work, byte stores only where a line or cross pixel lies.
#include "lvk_hip_internal.hpp"

#include <cmath>

namespace {

// Drawing.cl:22-39.  One thread per pixel of a row segment; only line pixels are written.
__global__ __launch_bounds__(256)
void k_draw_grid(uint8_t* __restrict__ dst, int dst_step, int rows, int cols, float cell_w, float cell_h, int thickness, uint32_t colour)
{
    const int x = (int)(blockIdx.x * 64 + threadIdx.x), y = (int)(blockIdx.y * 4 + threadIdx.y);
    if (x >= cols || y >= rows) return;
    const float fx = fmodf((float)x, cell_w), fy = fmodf((float)y, cell_h);        // exact by definition
    const float t = (float)thickness;
    if (fx < t || fy < t || fx > cell_w - t - 1.0f || fy > cell_h - t - 1.0f)
    {
        uint8_t* d = dst + (long)y * dst_step + 3 * x;
        d[0] = (uint8_t)colour; d[1] = (uint8_t)(colour >> 8); d[2] = (uint8_t)(colour >> 16);
    }
}

// Drawing.cl:75-105.  One thread per point; overlapping crosses store the same colour, so the store order is immaterial.
__global__ __launch_bounds__(64)
void k_draw_crosses(const int2* __restrict__ pts, int n, uint8_t* __restrict__ dst, int dst_step, int rows, int cols,
                    int cross_size, int thickness, uint32_t colour)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int2 c = pts[i];
    // the sums below cannot overflow: the host clamps coordinates to +-2^30
    int x = max(c.x - cross_size, 0), y = max(c.y - cross_size, 0);
    const int max_x = min(c.x + cross_size + 1, cols - thickness), max_y = min(c.y + cross_size + 1, rows - thickness);
    const uint8_t c0 = (uint8_t)colour, c1 = (uint8_t)(colour >> 8), c2 = (uint8_t)(colour >> 16);
    for (int k = 1; x < max_x && y < max_y; k++)
    {
        for (int dx = 0; dx < thickness; dx++)
        {
            uint8_t* f = dst + (long)y * dst_step + 3 * (x + dx);
            f[0] = c0; f[1] = c1; f[2] = c2;
            uint8_t* b = dst + (long)y * dst_step + 3 * (max_x - k + dx);
            b[0] = c0; b[1] = c1; b[2] = c2;
        }
        x++; y++;
    }
}

inline uint32_t pack3(const uint8_t c[3]) { return (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16); }

} // namespace

int lvk_launch_draw_grid(lvk_hip_ctx* ctx, hipStream_t stream, void* d_dst, int dst_step, int rows, int cols, int grid_w, int grid_h,
                         const uint8_t colour[3], int thickness)
{
    LVK_HIP_REQUIRE(ctx, d_dst && rows > 0 && cols > 0 && dst_step >= 3 * cols && colour);                  // Drawing.tpp:60-62
    LVK_HIP_REQUIRE(ctx, thickness >= 1 && grid_w >= 1 && grid_h >= 1);
    const float cw = (float)cols / (float)grid_w, ch = (float)rows / (float)grid_h;                       // Drawing.tpp:70-71
    hipLaunchKernelGGL(k_draw_grid, dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 3) / 4)), dim3(64, 4), 0, stream,
                       (uint8_t*)d_dst, dst_step, rows, cols, cw, ch, thickness, pack3(colour));
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

// pts: n (x, y) floats on the HOST; scaled like cv::multiply(points, Scalar(sx, sy), CV_32S) on 32F data: binary32 product,
// round half to even, saturate (Drawing.tpp:170-173)
int lvk_launch_draw_crosses(lvk_hip_ctx* ctx, hipStream_t stream, void* d_dst, int dst_step, int rows, int cols, const float* pts, int n,
                            float scale_x, float scale_y, const uint8_t colour[3], int cross_size, int thickness)
{
    LVK_HIP_REQUIRE(ctx, d_dst && rows > 0 && cols > 0 && dst_step >= 3 * cols && colour && (pts || n == 0) && n >= 0);
    LVK_HIP_REQUIRE(ctx, scale_x >= 0 && scale_y >= 0 && thickness >= 1 && cross_size >= 1);                // Drawing.tpp:155-159
    if (n == 0) return LVK_HIP_OK;                                                                         // Drawing.tpp:161-162
    LVK_HIP_REQUIRE(ctx, (size_t)n * sizeof(int2) <= lvk_hip_ctx::kStageBytes);
    std::vector<int2> ip((size_t)n);
    auto to_int = [](float v) -> int {
        if (!(v == v)) return 0;
        const float r = std::nearbyintf(v);                                  // FE_TONEAREST: half to even
        const float lim = 1073741824.0f;                                     // keeps the kernel's +- cross_size sums in range
        return (int)std::fmin(std::fmax(r, -lim), lim);
    };
    for (int i = 0; i < n; i++) ip[(size_t)i] = make_int2(to_int(pts[2 * i] * scale_x), to_int(pts[2 * i + 1] * scale_y));
    void* d_pts = nullptr;
    int rc = lvk_stage_params(ctx, stream, ip.data(), ip.size() * sizeof(int2), &d_pts);
    if (rc != LVK_HIP_OK) return rc;
    hipLaunchKernelGGL(k_draw_crosses, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, (const int2*)d_pts, n, (uint8_t*)d_dst, dst_step, rows, cols,
                       (cross_size + 1) / 2, thickness, pack3(colour));                                  // Drawing.tpp:183
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

extern "C" {

int lvk_hip_draw_grid(lvk_hip_ctx* ctx, void* d_dst, int dst_step, int rows, int cols, int grid_w, int grid_h, const uint8_t colour[3], int thickness)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_draw_grid(ctx, ctx->stream, d_dst, dst_step, rows, cols, grid_w, grid_h, colour, thickness);
}

int lvk_hip_draw_crosses(lvk_hip_ctx* ctx, void* d_dst, int dst_step, int rows, int cols, const float* pts_xy, int n,
                         float scale_x, float scale_y, const uint8_t colour[3], int cross_size, int thickness)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_draw_crosses(ctx, ctx->stream, d_dst, dst_step, rows, cols, pts_xy, n, scale_x, scale_y, colour, cross_size, thickness);
}

} // extern "C"
