// ORACLE (test infrastructure only -- see lvk_oracle.h).
// CPU restatement of the OpenCV 4.8.0 image operations the tracker calls (sources are NOT in /root/reference;
// OpenCV is pinned by Scripts/setup_deb.sh:42).  Call sites:
//   cv::extractChannel(YUV, 0)                    Data/VideoFrame.cpp:260
//   cv::resize(..., INTER_AREA)                   Vision/FrameTracker.cpp:117
//   pyrDown / Scharr derivatives                  inside cv::SparsePyrLKOpticalFlow::calc, Vision/FrameTracker.cpp:140-146
//   cv::FastFeatureDetector(TYPE_9_16, nms=true)  Vision/FeatureDetector.cpp:38-41,130-134
// Everything here is integer arithmetic except the non-integer INTER_AREA path (binary32, no contraction).
// Pinning (no OpenCV binary or source in this image): independent numpy restatements of every function (tests/test_oracle_imgproc.py, tests/np_pyrlk.py),
// and third-party fixtures where the image holds an implementation -- FAST-9/16 against scikit-image's segment test (keypoints and scores equal,
// tests/test_fast_third_party.py), the integer box mean and the bilinear alignment against scikit-image (tests/test_resize_third_party.py).
#include "lvk_oracle.h"
#include "parallel.h"

#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

namespace {

inline uint8_t sat_u8_round(float v)          // cv::saturate_cast<uchar>(float): cvRound (half to even) then clamp
{
    long r = lrintf(v);
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

inline int reflect101(int p, int len)        // cv::borderInterpolate(p, len, BORDER_REFLECT_101)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len)
    {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

struct AreaTab { int di, si; float alpha; };

// imgproc/resize.cpp computeResizeAreaTab
std::vector<AreaTab> area_tab(int ssize, int dsize, double scale)
{
    std::vector<AreaTab> tab;
    for (int dx = 0; dx < dsize; dx++)
    {
        const double fsx1 = dx * scale;
        const double fsx2 = fsx1 + scale;
        const double cellWidth = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3)
            tab.push_back({dx, sx1 - 1, (float)((sx1 - fsx1) / cellWidth)});
        for (int sx = sx1; sx < sx2; sx++)
            tab.push_back({dx, sx, (float)(1.0 / cellWidth)});
        if (fsx2 - sx2 > 1e-3)
            tab.push_back({dx, sx2, (float)(std::min(std::min(fsx2 - sx2, 1.0), cellWidth) / cellWidth)});
    }
    return tab;
}

} // namespace

extern "C" {

// a3 + a4: gray = channel `channel` of a packed frame with `pix_stride` bytes per pixel (3 for 8UC3, 1 for planar), or for
// channel -1 (BGR) / -2 (RGB) cv::cvtColor(COLOR_BGR2GRAY / COLOR_RGB2GRAY) (VideoFrame.cpp:194; OpenCV 4.8 RGB2Gray<uchar>:
// (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15), then cv::resize(gray, dst, (dcols, drows), INTER_AREA).
int lvko_luma_area_resize(const uint8_t* src, int src_step, int pix_stride, int channel, int srows, int scols,
                          uint8_t* dst, int dst_step, int drows, int dcols)
{
    if (!src || !dst || srows <= 0 || scols <= 0 || drows <= 0 || dcols <= 0) return -1;
    if (channel < -2 || channel >= pix_stride || (channel < 0 && pix_stride < 3)) return -1;
    auto S = [&](int y, int x) -> int {
        const uint8_t* p = src + (size_t)y * src_step + (size_t)x * pix_stride;
        if (channel >= 0) return p[channel];
        const int b = channel == -1 ? p[0] : p[2], g = p[1], r = channel == -1 ? p[2] : p[0];
        return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15;
    };
    if (drows == srows && dcols == scols)
    {
        for (int y = 0; y < drows; y++) for (int x = 0; x < dcols; x++) dst[(size_t)y * dst_step + x] = (uint8_t)S(y, x);
        return 0;
    }
    if (dcols > scols || drows > srows)
    {
        // A frame smaller than the detection resolution on either axis.  cv::resize: "true area interpolation is only implemented for the case
        // (scale_x >= 1 && scale_y >= 1); in other cases it is emulated using some variant of bilinear interpolation" -- the INTER_LINEAR
        // machinery (2 taps per axis, fixed point for 8U: INTER_RESIZE_COEF_BITS = 11) with AREA coefficients, on both axes:
        //   sx = floor(dx * scale); fx = (dx + 1) - (sx + 1) * inv_scale; fx = fx <= 0 ? 0 : fx - floor(fx)
        //   taps (sx, sx + 1) with weights saturate_cast<short>((1 - fx) * 2048), saturate_cast<short>(fx * 2048); past the last source column
        //   (sx + 1 >= cols, and every dx after the first such one) the single tap S[min(sx, cols - 1)] * 2048; rows sy and sy + 1 clamped to rows - 1
        //   HResizeLinear: int D = S[sx] * a0 + S[sx + 1] * a1;  VResizeLinear (8U): (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
        // (imgproc/resize.cpp: cv::hal::resize, HResizeLinear, VResizeLinear<uchar, int, short, FixedPtCast<...>>; restated from the published
        // source, not part of /root/reference: parity unpinned like the other OpenCV stages.)
        const double inv_x = (double)dcols / scols, inv_y = (double)drows / srows, sc_x = 1. / inv_x, sc_y = 1. / inv_y;
        auto coef = [](int d, double scale, double inv, int& s0, float& f) {
            s0 = (int)std::floor(d * scale);
            f = (float)((d + 1) - (s0 + 1) * inv);
            f = f <= 0 ? 0.f : f - std::floor(f);
        };
        auto to_short = [](float v) -> int { long r = lrintf(v); return (int)(r < -32768 ? -32768 : r > 32767 ? 32767 : r); };
        std::vector<int> xofs(dcols), xa0(dcols), xa1(dcols);
        int xmax = dcols;
        for (int dx = 0; dx < dcols; dx++)
        {
            int sx; float fx; coef(dx, sc_x, inv_x, sx, fx);
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx + 1 >= scols) { xmax = std::min(xmax, dx); if (sx >= scols - 1) { fx = 0; sx = scols - 1; } }
            xofs[dx] = sx; xa0[dx] = to_short((1.f - fx) * 2048.f); xa1[dx] = to_short(fx * 2048.f);
        }
        for (int dy = 0; dy < drows; dy++)
        {
            int sy; float fy; coef(dy, sc_y, inv_y, sy, fy);
            const int b0 = to_short((1.f - fy) * 2048.f), b1 = to_short(fy * 2048.f);
            const int y0 = std::min(std::max(sy, 0), srows - 1), y1 = std::min(std::max(sy + 1, 0), srows - 1);
            for (int dx = 0; dx < dcols; dx++)
            {
                int r0, r1;
                if (dx < xmax) { r0 = S(y0, xofs[dx]) * xa0[dx] + S(y0, xofs[dx] + 1) * xa1[dx]; r1 = S(y1, xofs[dx]) * xa0[dx] + S(y1, xofs[dx] + 1) * xa1[dx]; }
                else { r0 = S(y0, xofs[dx]) * 2048; r1 = S(y1, xofs[dx]) * 2048; }
                dst[(size_t)dy * dst_step + dx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
            }
        }
        return 0;
    }
    const double scale_x = (double)scols / dcols, scale_y = (double)srows / drows;
    const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    const bool fast = std::fabs(scale_x - iscale_x) < 2.220446049250313e-16 && std::fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (fast)
    {
        // resizeAreaFast_: integer box sum * (1.f / area), saturate_cast (round half to even);
        // the 2x2 case uses the dedicated (a + b + c + d + 2) >> 2 vector kernel.
        const int area = iscale_x * iscale_y;
        const float scale = 1.f / (float)area;
        lvko_parallel_for(drows, 8, [&](int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < dcols; x++)
            {
                int sum = 0;
                for (int ky = 0; ky < iscale_y; ky++)
                    for (int kx = 0; kx < iscale_x; kx++)
                        sum += S(y * iscale_y + ky, x * iscale_x + kx);
                dst[(size_t)y * dst_step + x] = (iscale_x == 2 && iscale_y == 2) ? (uint8_t)((sum + 2) >> 2) : sat_u8_round((float)sum * scale);
            }
        });
        return 0;
    }
    // resizeArea_: separable "decimate alpha" tables, float accumulation in table order.
    const std::vector<AreaTab> xtab = area_tab(scols, dcols, scale_x), ytab = area_tab(srows, drows, scale_y);
    std::vector<float> buf(dcols), sum(dcols, 0.0f);
    int prev_dy = ytab.empty() ? 0 : ytab[0].di;
    for (size_t j = 0; j < ytab.size(); j++)
    {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di, sy = ytab[j].si;
        std::fill(buf.begin(), buf.end(), 0.0f);
        for (const AreaTab& t : xtab) buf[t.di] = buf[t.di] + (float)S(sy, t.si) * t.alpha;
        if (dy != prev_dy)
        {
            for (int dx = 0; dx < dcols; dx++) { dst[(size_t)prev_dy * dst_step + dx] = sat_u8_round(sum[dx]); sum[dx] = beta * buf[dx]; }
            prev_dy = dy;
        }
        else
            for (int dx = 0; dx < dcols; dx++) sum[dx] = sum[dx] + beta * buf[dx];
    }
    for (int dx = 0; dx < dcols; dx++) dst[(size_t)prev_dy * dst_step + dx] = sat_u8_round(sum[dx]);
    return 0;
}

// cv::pyrDown (8UC1, BORDER_REFLECT_101): 5x5 [1 4 6 4 1] separable, (sum + 128) >> 8, dst = ((w+1)/2, (h+1)/2).
int lvko_pyr_down(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dst, int dst_step)
{
    const int drows = (rows + 1) / 2, dcols = (cols + 1) / 2;
    for (int y = 0; y < drows; y++)
        for (int x = 0; x < dcols; x++)
        {
            static const int w[5] = {1, 4, 6, 4, 1};
            int acc = 0;
            for (int ky = 0; ky < 5; ky++)
            {
                const uint8_t* row = src + (size_t)reflect101(2 * y - 2 + ky, rows) * src_step;
                int h = 0;
                for (int kx = 0; kx < 5; kx++) h += w[kx] * row[reflect101(2 * x - 2 + kx, cols)];
                acc += w[ky] * h;
            }
            dst[(size_t)y * dst_step + x] = (uint8_t)((acc + 128) >> 8);
        }
    return 0;
}

// video/lkpyramid.cpp calcScharrDeriv: int16 (Ix, Iy) interleaved, REFLECT_101 at the image edge.
int lvko_scharr_deriv(const uint8_t* src, int src_step, int rows, int cols, int16_t* dst /* rows*cols*2 */)
{
    std::vector<int> t0(cols + 2), t1(cols + 2);
    for (int y = 0; y < rows; y++)
    {
        const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : rows > 1 ? 1 : 0) * src_step;
        const uint8_t* r1 = src + (size_t)y * src_step;
        const uint8_t* r2 = src + (size_t)(y < rows - 1 ? y + 1 : rows > 1 ? rows - 2 : 0) * src_step;
        for (int x = 0; x < cols; x++)
        {
            t0[x + 1] = (int16_t)((r0[x] + r2[x]) * 3 + r1[x] * 10);
            t1[x + 1] = (int16_t)(r2[x] - r0[x]);
        }
        const int x0 = cols > 1 ? 1 : 0, x1 = cols > 1 ? cols - 2 : 0;
        t0[0] = t0[x0 + 1]; t0[cols + 1] = t0[x1 + 1];
        t1[0] = t1[x0 + 1]; t1[cols + 1] = t1[x1 + 1];
        for (int x = 0; x < cols; x++)
        {
            dst[((size_t)y * cols + x) * 2 + 0] = (int16_t)(t0[x + 2] - t0[x]);
            dst[((size_t)y * cols + x) * 2 + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    return 0;
}

// features2d/fast.cpp FAST_t<16> + cornerScore<16> on an ROI of `img` (the ROI edge is the image edge).
// Emits (x, y, score) in ROI-local coordinates, row-major (the CPU path's order).  Returns the count
// (all of them are counted even when `cap` is smaller).
int lvko_fast9_16(const uint8_t* img, int step, int roi_x, int roi_y, int roi_w, int roi_h, int threshold,
                  int* out_xys /* 3 ints each */, int cap)
{
    static const int off[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
    threshold = std::min(std::max(threshold, 0), 255);
    const uint8_t* base = img + (size_t)roi_y * step + roi_x;
    std::vector<uint8_t> score((size_t)roi_w * roi_h, 0);
    for (int i = 3; i < roi_h - 3; i++)
        for (int j = 3; j < roi_w - 3; j++)
        {
            const uint8_t* p = base + (size_t)i * step + j;
            const int v = p[0];
            int d[25];
            for (int k = 0; k < 25; k++) d[k] = v - p[off[k % 16][0] + off[k % 16][1] * step];
            // corner test: > 8 contiguous ring pixels darker than v - t or brighter than v + t (strict)
            bool corner = false;
            int count = 0;
            for (int k = 0; k < 25 && !corner; k++) { if (d[k] > threshold) { if (++count > 8) corner = true; } else count = 0; }
            count = 0;
            for (int k = 0; k < 25 && !corner; k++) { if (d[k] < -threshold) { if (++count > 8) corner = true; } else count = 0; }
            if (!corner) continue;
            // cornerScore<16>
            int a0 = threshold;
            for (int k = 0; k < 16; k += 2)
            {
                int a = std::min(d[k + 1], d[k + 2]);
                a = std::min(a, d[k + 3]);
                if (a <= a0) continue;
                a = std::min(a, d[k + 4]); a = std::min(a, d[k + 5]); a = std::min(a, d[k + 6]); a = std::min(a, d[k + 7]); a = std::min(a, d[k + 8]);
                a0 = std::max(a0, std::min(a, d[k]));
                a0 = std::max(a0, std::min(a, d[k + 9]));
            }
            int b0 = -a0;
            for (int k = 0; k < 16; k += 2)
            {
                int b = std::max(d[k + 1], d[k + 2]);
                b = std::max(b, d[k + 3]); b = std::max(b, d[k + 4]); b = std::max(b, d[k + 5]);
                if (b >= b0) continue;
                b = std::max(b, d[k + 6]); b = std::max(b, d[k + 7]); b = std::max(b, d[k + 8]);
                b0 = std::min(b0, std::max(b, d[k]));
                b0 = std::min(b0, std::max(b, d[k + 9]));
            }
            score[(size_t)i * roi_w + j] = (uint8_t)(-b0 - 1);
        }
    // 3x3 non-max suppression: strictly greater than all 8 neighbours (non-corners score 0)
    int n = 0;
    for (int i = 3; i < roi_h - 3; i++)
        for (int j = 3; j < roi_w - 3; j++)
        {
            const int s = score[(size_t)i * roi_w + j];
            if (s == 0) continue;
            const uint8_t* pp = &score[(size_t)(i - 1) * roi_w + j];
            const uint8_t* pc = &score[(size_t)i * roi_w + j];
            const uint8_t* pn = &score[(size_t)(i + 1) * roi_w + j];
            if (s > pc[-1] && s > pc[1] && s > pp[-1] && s > pp[0] && s > pp[1] && s > pn[-1] && s > pn[0] && s > pn[1])
            {
                if (n < cap) { out_xys[3 * n] = j; out_xys[3 * n + 1] = i; out_xys[3 * n + 2] = s; }
                n++;
            }
        }
    return n;
}

static int g_lvko_threads = 1;
int lvko_set_num_threads(int n) { const int prev = g_lvko_threads; g_lvko_threads = n < 1 ? 1 : n; return prev; }

} // extern "C"

int lvko_num_threads() { return g_lvko_threads; }
