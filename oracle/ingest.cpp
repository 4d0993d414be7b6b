// ORACLE (test infrastructure only -- see lvk_oracle.h).
// CPU restatement of the YUV420 <-> packed YUV444 conversion either side of the filter in the OBS async path
// (reference: Modules/OBS-Plugin/Interop/FrameIngest.cpp:494-557 I4XXIngest::to_ocl / to_obs and :567-602 NV12Ingest):
//   ingest: cv::resize(U, frame_size, INTER_LINEAR), same for V, cv::merge(Y, U, V) -> 8UC3
//   egress: cv::split, cv::resize(U, Size(), 0.5, 0.5, INTER_AREA), same for V
// Arithmetic = OpenCV 4.8.0 imgproc/resize.cpp CPU paths (source not in /root/reference):
//   8U INTER_LINEAR: fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), edge clamping as for the float tables;
//     coefficients (1 - fx, fx) * 2048 rounded to short; horizontal pass in int: S[sx] * a0 + S[sx + 1] * a1 (single tap
//     * 2048 beyond xmax); vertical pass: uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2), the two
//     source rows clipped individually;
//   INTER_AREA with scale exactly 2: (a + b + c + d + 2) >> 2.
#include "lvk_oracle.h"
#include "parallel.h"

#include <cmath>
#include <vector>
#include <algorithm>

namespace {

struct LinTab8 { std::vector<int> s0, s1; std::vector<short> a0, a1; };

LinTab8 make_tab(int ssize, int dsize, bool vertical)
{
    LinTab8 t; t.s0.resize(dsize); t.s1.resize(dsize); t.a0.resize(dsize); t.a1.resize(dsize);
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; d++)
    {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        if (vertical)
        {
            t.s0[d] = std::min(std::max(s, 0), ssize - 1);
            t.s1[d] = std::min(std::max(s + 1, 0), ssize - 1);
            t.a0[d] = (short)lrintf((1.f - f) * 2048.f);
            t.a1[d] = (short)lrintf(f * 2048.f);
            continue;
        }
        if (s < 0) { f = 0.f; s = 0; }
        bool single = false;
        if (s + 1 >= ssize) { single = true; if (s >= ssize - 1) { f = 0.f; s = ssize - 1; } }
        t.s0[d] = s; t.s1[d] = single ? s : s + 1;
        t.a0[d] = single ? (short)2048 : (short)lrintf((1.f - f) * 2048.f);
        t.a1[d] = single ? (short)0 : (short)lrintf(f * 2048.f);
    }
    return t;
}

// one channel of cv::resize(8UC1 or interleaved 8UCn, INTER_LINEAR) evaluated at (x, y)
inline uint8_t lin8(const uint8_t* src, int step, int pix, int ch, const LinTab8& tx, const LinTab8& ty, int x, int y)
{
    const uint8_t* r0 = src + (size_t)ty.s0[y] * step + ch;
    const uint8_t* r1 = src + (size_t)ty.s1[y] * step + ch;
    const int h0 = r0[(size_t)tx.s0[x] * pix] * tx.a0[x] + r0[(size_t)tx.s1[x] * pix] * tx.a1[x];
    const int h1 = r1[(size_t)tx.s0[x] * pix] * tx.a0[x] + r1[(size_t)tx.s1[x] * pix] * tx.a1[x];
    return (uint8_t)(((((int)ty.a0[y] * (h0 >> 4)) >> 16) + (((int)ty.a1[y] * (h1 >> 4)) >> 16) + 2) >> 2);
}

} // namespace

extern "C" {

// I420 / NV12 -> packed YUV444 8UC3.  nv12 != 0: `u` points at the interleaved UV plane (`v` ignored).  rows, cols even.
int lvko_ingest_yuv420(const uint8_t* y, int y_step, const uint8_t* u, int u_step, const uint8_t* v, int v_step, int nv12,
                       int rows, int cols, uint8_t* dst, int dst_step)
{
    if (!y || !u || (!nv12 && !v) || !dst || rows <= 0 || cols <= 0 || (rows & 1) || (cols & 1)) return -1;
    const int cr = rows / 2, cc = cols / 2;
    const LinTab8 tx = make_tab(cc, cols, false), ty = make_tab(cr, rows, true);
    lvko_parallel_for(rows, 16, [&](int r0, int r1) {
    for (int yy = r0; yy < r1; yy++)
        for (int xx = 0; xx < cols; xx++)
        {
            uint8_t* d = dst + (size_t)yy * dst_step + 3 * (size_t)xx;
            d[0] = y[(size_t)yy * y_step + xx];
            if (nv12) { d[1] = lin8(u, u_step, 2, 0, tx, ty, xx, yy); d[2] = lin8(u, u_step, 2, 1, tx, ty, xx, yy); }
            else { d[1] = lin8(u, u_step, 1, 0, tx, ty, xx, yy); d[2] = lin8(v, v_step, 1, 0, tx, ty, xx, yy); }
        }
    });
    return 0;
}

// packed YUV444 8UC3 -> I420 / NV12 (chroma by 2x2 INTER_AREA).
int lvko_egress_yuv420(const uint8_t* src, int src_step, int rows, int cols,
                       uint8_t* y, int y_step, uint8_t* u, int u_step, uint8_t* v, int v_step, int nv12)
{
    if (!src || !y || !u || (!nv12 && !v) || rows <= 0 || cols <= 0 || (rows & 1) || (cols & 1)) return -1;
    lvko_parallel_for(rows / 2, 8, [&](int c0, int c1) {
    for (int yy = 2 * c0; yy < 2 * c1; yy++)
        for (int xx = 0; xx < cols; xx++) y[(size_t)yy * y_step + xx] = src[(size_t)yy * src_step + 3 * (size_t)xx];
    for (int cy = c0; cy < c1; cy++)
        for (int cx = 0; cx < cols / 2; cx++)
            for (int ch = 1; ch <= 2; ch++)
            {
                const uint8_t* p = src + (size_t)(2 * cy) * src_step + 3 * (size_t)(2 * cx) + ch;
                const int s = p[0] + p[3] + p[src_step] + p[src_step + 3];
                const uint8_t o = (uint8_t)((s + 2) >> 2);
                if (nv12) u[(size_t)cy * u_step + 2 * cx + (ch - 1)] = o;
                else (ch == 1 ? u : v)[(size_t)cy * (ch == 1 ? u_step : v_step) + cx] = o;
            }
    });
    return 0;
}

} // extern "C"
