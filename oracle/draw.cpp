// ORACLE (test infrastructure only -- see lvk_oracle.h).
// CPU restatement of the debug overlays: lvk::draw_grid / lvk::draw_crosses
// (LiveVisionKit/Functions/Drawing.tpp:53-93,146-196) with their kernels `grid` / `crosses`
// (Functions/OpenCL/Sources/Drawing.cl:22-39,75-105), each work-item run as a loop iteration.
#include "lvk_oracle.h"

#include <algorithm>
#include <cmath>
#include <vector>

extern "C" {

int lvko_draw_grid(uint8_t* dst, int dst_step, int rows, int cols, int grid_w, int grid_h, const uint8_t colour[3], int thickness)
{
    if (!dst || rows <= 0 || cols <= 0 || grid_w < 1 || grid_h < 1 || thickness < 1) return -1;
    const float cell_width = (float)cols / (float)grid_w, cell_height = (float)rows / (float)grid_h;        // Drawing.tpp:70-71
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++)
        {
            // Drawing.cl:30-35 (int arguments of fmod / the comparisons promote to float)
            const float fx = std::fmod((float)x, cell_width), fy = std::fmod((float)y, cell_height);
            if (fx < (float)thickness || fy < (float)thickness
                || fx > cell_width - (float)thickness - 1.0f || fy > cell_height - (float)thickness - 1.0f)
            {
                uint8_t* d = dst + (size_t)y * dst_step + 3 * x;
                d[0] = colour[0]; d[1] = colour[1]; d[2] = colour[2];
            }
        }
    return 0;
}

int lvko_draw_crosses(uint8_t* dst, int dst_step, int rows, int cols, const float* pts, int n, float scale_x, float scale_y,
                      const uint8_t colour[3], int cross_size, int cross_thickness)
{
    if (!dst || rows <= 0 || cols <= 0 || n < 0 || scale_x < 0 || scale_y < 0 || cross_size < 1 || cross_thickness < 1) return -1;
    if (n == 0) return 0;                                                                                  // Drawing.tpp:161-162
    // Drawing.tpp:170-173: cv::multiply(32FC2 points, Scalar(sx, sy), CV_32S): binary32 product, saturate_cast<int> (half to even)
    auto to_int = [](float v) -> int {
        if (!(v == v)) return 0;
        const float r = std::nearbyintf(v);
        const float lim = 1073741824.0f;
        return (int)std::fmin(std::fmax(r, -lim), lim);
    };
    const int size = (cross_size + 1) / 2;                                                                 // Drawing.tpp:183
    for (int i = 0; i < n; i++)
    {
        const int cx = to_int(pts[2 * i] * scale_x), cy = to_int(pts[2 * i + 1] * scale_y);
        // Drawing.cl:86-104
        int x = std::max(cx - size, 0), y = std::max(cy - size, 0);
        const int max_x = std::min(cx + size + 1, cols - cross_thickness), max_y = std::min(cy + size + 1, rows - cross_thickness);
        for (int k = 1; x < max_x && y < max_y; k++)
        {
            for (int dx = 0; dx < cross_thickness; dx++)
            {
                uint8_t* f = dst + (size_t)y * dst_step + 3 * (x + dx);
                f[0] = colour[0]; f[1] = colour[1]; f[2] = colour[2];
                uint8_t* b = dst + (size_t)y * dst_step + 3 * (max_x - k + dx);
                b[0] = colour[0]; b[1] = colour[1]; b[2] = colour[2];
            }
            x++; y++;
        }
    }
    return 0;
}

} // extern "C"
