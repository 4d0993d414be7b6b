// ORACLE (test infrastructure only -- see lvk_oracle.h).
// CPU restatement of the YUV420 <-> packed YUV444 conversion either side of the filter in the OBS async path
// (reference: Modules/OBS-Plugin/Interop/FrameIngest.cpp:494-557 I4XXIngest::to_ocl / to_obs and :567-602 NV12Ingest; round 6: every other
// format FrameIngest::Select knows, :604-753, in lvko_ingest_obs / lvko_egress_obs below):
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

// ---- the other OBS video formats of FrameIngest::Select (FrameIngest.cpp:36-75): I4XX (I444 / I422 / I420 and their alpha twins, :476-557),
// packed 4:2:2 (YUY2 / YVYU / UYVY, P422Ingest :604-666), packed 4:4:4 (AYUV, P444Ingest :670-703), uncompressed (Y800 / RGBA / BGRX / BGRA /
// BGR3, DirectIngest :705-753).  `fmt` = libobs' enum video_format (media-io/video-io.h of libobs 27.2.4: NONE 0, I420 1, NV12 2, YVYU 3, YUY2 4,
// UYVY 5, RGBA 6, BGRA 7, BGRX 8, Y800 9, I444 10, BGR3 11, I422 12, I40A 13, I42A 14, YUVA 15, AYUV 16), the value obs_source_frame::format holds.
// Arithmetic as above: INTER_LINEAR chroma upsampling on the way in (the same two-pass fixed point also when only the width doubles: the vertical
// pass then has the coefficients (2048, 0)); on the way out cv::resize(INTER_AREA) with (0.5, 0.5) = (a + b + c + d + 2) >> 2 and with (0.5, 1.0) =
// resizeAreaFast_'s generic loop, saturate_cast<uchar>((a + b) * 0.5f): round half to EVEN.
enum { VF_I420 = 1, VF_NV12 = 2, VF_YVYU = 3, VF_YUY2 = 4, VF_UYVY = 5, VF_RGBA = 6, VF_BGRA = 7, VF_BGRX = 8, VF_Y800 = 9, VF_I444 = 10, VF_BGR3 = 11,
       VF_I422 = 12, VF_I40A = 13, VF_I42A = 14, VF_YUVA = 15, VF_AYUV = 16 };

static inline uint8_t half_even(int s) { return (uint8_t)((s + ((s >> 1) & 1)) >> 1); }     // cvRound(s * 0.5f)

int lvko_ingest_obs(int fmt, const uint8_t* const planes[3], const int steps[3], int rows, int cols, uint8_t* dst, int dst_step)
{
    if (!planes || !steps || !planes[0] || !dst || rows <= 0 || cols <= 0) return -1;
    switch (fmt)
    {
    case VF_I420: case VF_I40A: return lvko_ingest_yuv420(planes[0], steps[0], planes[1], steps[1], planes[2], steps[2], 0, rows, cols, dst, dst_step);
    case VF_NV12: return lvko_ingest_yuv420(planes[0], steps[0], planes[1], steps[1], nullptr, 0, 1, rows, cols, dst, dst_step);
    case VF_I444: case VF_YUVA:                                  // merge_planes only (:521)
        if (!planes[1] || !planes[2]) return -1;
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
            {
                uint8_t* d = dst + (size_t)y * dst_step + 3 * (size_t)x;
                d[0] = planes[0][(size_t)y * steps[0] + x]; d[1] = planes[1][(size_t)y * steps[1] + x]; d[2] = planes[2][(size_t)y * steps[2] + x];
            }
        return 0;
    case VF_I422: case VF_I42A:                                  // chroma (cols / 2) x rows -> cols x rows, INTER_LINEAR (:514-519)
    {
        if (!planes[1] || !planes[2] || (cols & 1)) return -1;
        const LinTab8 tx = make_tab(cols / 2, cols, false), ty = make_tab(rows, rows, true);
        lvko_parallel_for(rows, 16, [&](int r0, int r1) {
        for (int y = r0; y < r1; y++)
            for (int x = 0; x < cols; x++)
            {
                uint8_t* d = dst + (size_t)y * dst_step + 3 * (size_t)x;
                d[0] = planes[0][(size_t)y * steps[0] + x];
                d[1] = lin8(planes[1], steps[1], 1, 0, tx, ty, x, y); d[2] = lin8(planes[2], steps[2], 1, 0, tx, ty, x, y);
            }
        });
        return 0;
    }
    case VF_YUY2: case VF_YVYU: case VF_UYVY:                    // P422Ingest::to_ocl (:615-636)
    {
        if (cols & 1) return -1;
        const int yoff = fmt == VF_UYVY ? 1 : 0, coff = 1 - yoff;         // m_YFirst
        const bool ufirst = fmt != VF_YVYU;                                 // m_UFirst
        // extractChannel(chroma) -> reshape(2 channels): a (cols / 2) x rows image of (first, second) chroma bytes, 4 bytes apart in the source
        const LinTab8 tx = make_tab(cols / 2, cols, false), ty = make_tab(rows, rows, true);
        const uint8_t* c = planes[0] + coff;
        lvko_parallel_for(rows, 16, [&](int r0, int r1) {
        for (int y = r0; y < r1; y++)
            for (int x = 0; x < cols; x++)
            {
                uint8_t* d = dst + (size_t)y * dst_step + 3 * (size_t)x;
                d[0] = planes[0][(size_t)y * steps[0] + 2 * (size_t)x + yoff];
                const uint8_t first = lin8(c, steps[0], 4, 0, tx, ty, x, y), second = lin8(c, steps[0], 4, 2, tx, ty, x, y);
                d[1] = ufirst ? first : second; d[2] = ufirst ? second : first;
            }
        });
        return 0;
    }
    case VF_AYUV:                                                // mixChannels {1,0, 2,1, 3,2} (:686)
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
            {
                const uint8_t* s = planes[0] + (size_t)y * steps[0] + 4 * (size_t)x;
                uint8_t* d = dst + (size_t)y * dst_step + 3 * (size_t)x;
                d[0] = s[1]; d[1] = s[2]; d[2] = s[3];
            }
        return 0;
    case VF_Y800:                                                // upload_planes(src, 1).copyTo(dst): rows x cols, one channel
        for (int y = 0; y < rows; y++) std::copy(planes[0] + (size_t)y * steps[0], planes[0] + (size_t)y * steps[0] + cols, dst + (size_t)y * dst_step);
        return 0;
    case VF_BGR3:
        for (int y = 0; y < rows; y++) std::copy(planes[0] + (size_t)y * steps[0], planes[0] + (size_t)y * steps[0] + 3 * (size_t)cols, dst + (size_t)y * dst_step);
        return 0;
    case VF_RGBA: case VF_BGRA: case VF_BGRX:
        // DirectIngest::to_ocl uploads rows * cols * 3 BYTES from data[0] and views them as a rows x cols 3-channel image (:743-747, "to avoid
        // unnecessarily uploading the alpha plane"): the first three quarters of the 4-byte pixels' byte stream, re-cut into 3-byte pixels.  Restated as
        // written; the source must be tightly packed (the reference ignores linesize).
        if (steps[0] != 4 * cols) return -1;
        for (int y = 0; y < rows; y++) std::copy(planes[0] + (size_t)y * 3 * cols, planes[0] + (size_t)(y + 1) * 3 * cols, dst + (size_t)y * dst_step);
        return 0;
    }
    return -1;
}

int lvko_egress_obs(int fmt, const uint8_t* src, int src_step, int rows, int cols, uint8_t* const planes[3], const int steps[3])
{
    if (!planes || !steps || !planes[0] || !src || rows <= 0 || cols <= 0) return -1;
    switch (fmt)
    {
    case VF_I420: case VF_I40A: return lvko_egress_yuv420(src, src_step, rows, cols, planes[0], steps[0], planes[1], steps[1], planes[2], steps[2], 0);
    case VF_NV12: return lvko_egress_yuv420(src, src_step, rows, cols, planes[0], steps[0], planes[1], steps[1], nullptr, 0, 1);
    case VF_I444: case VF_YUVA:
        if (!planes[1] || !planes[2]) return -1;
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
                for (int ch = 0; ch < 3; ch++) planes[ch][(size_t)y * steps[ch] + x] = src[(size_t)y * src_step + 3 * (size_t)x + ch];
        return 0;
    case VF_I422: case VF_I42A:                                  // split + cv::resize(Size(), 0.5, 1.0, INTER_AREA) (:533-552)
        if (!planes[1] || !planes[2] || (cols & 1)) return -1;
        for (int y = 0; y < rows; y++)
        {
            const uint8_t* s = src + (size_t)y * src_step;
            for (int x = 0; x < cols; x++) planes[0][(size_t)y * steps[0] + x] = s[3 * (size_t)x];
            for (int cx = 0; cx < cols / 2; cx++)
                for (int ch = 1; ch <= 2; ch++) planes[ch][(size_t)y * steps[ch] + cx] = half_even(s[6 * (size_t)cx + ch] + s[6 * (size_t)cx + 3 + ch]);
        }
        return 0;
    case VF_YUY2: case VF_YVYU: case VF_UYVY:                    // P422Ingest::to_obs (:640-666)
    {
        if (cols & 1) return -1;
        const int yoff = fmt == VF_UYVY ? 1 : 0, coff = 1 - yoff;
        const bool ufirst = fmt != VF_YVYU;
        for (int y = 0; y < rows; y++)
        {
            const uint8_t* s = src + (size_t)y * src_step;
            uint8_t* d = planes[0] + (size_t)y * steps[0];
            for (int cx = 0; cx < cols / 2; cx++)
            {
                const uint8_t u = half_even(s[6 * (size_t)cx + 1] + s[6 * (size_t)cx + 4]), v = half_even(s[6 * (size_t)cx + 2] + s[6 * (size_t)cx + 5]);
                d[4 * (size_t)cx + yoff] = s[6 * (size_t)cx]; d[4 * (size_t)cx + 2 + yoff] = s[6 * (size_t)cx + 3];
                d[4 * (size_t)cx + coff] = ufirst ? u : v; d[4 * (size_t)cx + 2 + coff] = ufirst ? v : u;
            }
        }
        return 0;
    }
    case VF_AYUV:                                                // setTo(255, 0, 0, 0) + mixChannels {0,1, 1,2, 2,3} (:694-701)
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++)
            {
                const uint8_t* s = src + (size_t)y * src_step + 3 * (size_t)x;
                uint8_t* d = planes[0] + (size_t)y * steps[0] + 4 * (size_t)x;
                d[0] = 255; d[1] = s[0]; d[2] = s[1]; d[3] = s[2];
            }
        return 0;
    case VF_Y800:
        for (int y = 0; y < rows; y++) std::copy(src + (size_t)y * src_step, src + (size_t)y * src_step + cols, planes[0] + (size_t)y * steps[0]);
        return 0;
    case VF_BGR3:
        for (int y = 0; y < rows; y++) std::copy(src + (size_t)y * src_step, src + (size_t)y * src_step + 3 * (size_t)cols, planes[0] + (size_t)y * steps[0]);
        return 0;
    case VF_RGBA: case VF_BGRA: case VF_BGRX:                    // download_planes(src, dst): rows * cols * 3 bytes into data[0] (:751-753), the rest untouched
        if (steps[0] != 4 * cols) return -1;
        for (int y = 0; y < rows; y++) std::copy(src + (size_t)y * src_step, src + (size_t)y * src_step + 3 * (size_t)cols, planes[0] + (size_t)y * 3 * cols);
        return 0;
    }
    return -1;
}

} // extern "C"
