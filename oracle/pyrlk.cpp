// ORACLE (test infrastructure only -- see lvk_oracle.h).
// CPU restatement of cv::SparsePyrLKOpticalFlow::calc as the reference configures it
// (Vision/FrameTracker.cpp:33-35,42-48,140-146: win 11x11, maxLevel 3, criteria COUNT+EPS (5, 0.01), flags 0,
//  minEigThreshold 1e-4).  Follows OpenCV 4.8.0 video/lkpyramid.cpp (buildOpticalFlowPyramid, calcScharrDeriv,
//  LKTrackerInvoker, CPU/fixed-point path); the source is not in /root/reference.
//
// Arithmetic definition where OpenCV is not bit-defined: the covariance sums (A11, A12, A22) and the
// mismatch sums (b1, b2) are accumulated EXACTLY in integers and converted to binary32 once (OpenCV
// accumulates in float in a SIMD-width dependent order); all other float ops are the ones written in
// lkpyramid.cpp, separately rounded (no contraction).
#include "lvk_oracle.h"
#include "parallel.h"

#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

extern "C" int lvko_pyr_down(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dst, int dst_step);
extern "C" int lvko_scharr_deriv(const uint8_t* src, int src_step, int rows, int cols, int16_t* dst);

namespace {

inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * (len - 1) - p; }
    return p;
}

struct Level
{
    int rows = 0, cols = 0;
    std::vector<uint8_t> img;       // rows x cols
    std::vector<int16_t> deriv;     // rows x cols x 2 (only needed for the previous frame)
    // buildOpticalFlowPyramid pads each level by winSize with BORDER_REFLECT_101; derivatives are padded with 0.
    int at(int y, int x) const { return img[(size_t)reflect101(y, rows) * cols + reflect101(x, cols)]; }
    int dat(int y, int x, int c) const
    {
        if (x < 0 || y < 0 || x >= cols || y >= rows) return 0;
        return deriv[((size_t)y * cols + x) * 2 + c];
    }
};

std::vector<Level> build_pyramid(const uint8_t* img, int step, int rows, int cols, int max_level, int win_w, int win_h, bool derivs)
{
    std::vector<Level> pyr;
    Level l0; l0.rows = rows; l0.cols = cols; l0.img.resize((size_t)rows * cols);
    for (int y = 0; y < rows; y++) std::memcpy(&l0.img[(size_t)y * cols], img + (size_t)y * step, cols);
    pyr.push_back(std::move(l0));
    for (int level = 1; level <= max_level; level++)
    {
        const Level& p = pyr.back();
        const int r = (p.rows + 1) / 2, c = (p.cols + 1) / 2;
        if (c <= win_w || r <= win_h) break;                     // lkpyramid.cpp: level too small -> stop
        Level l; l.rows = r; l.cols = c; l.img.resize((size_t)r * c);
        lvko_pyr_down(p.img.data(), p.cols, p.rows, p.cols, l.img.data(), c);
        pyr.push_back(std::move(l));
    }
    if (derivs)
        for (Level& l : pyr)
        {
            l.deriv.resize((size_t)l.rows * l.cols * 2);
            lvko_scharr_deriv(l.img.data(), l.cols, l.rows, l.cols, l.deriv.data());
        }
    return pyr;
}

inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// How a window sum of int products is formed.
//   lanes == 0: the SPECIFICATION -- exact integer sum, converted to binary32 once (what the HIP kernel computes).
//   lanes >= 1: the way OpenCV 4.8's LKTrackerInvoker forms it (video/src/lkpyramid.cpp), kept to MEASURE how far the specification is from
//               it: binary32 accumulation in window row-major order.  lanes == 1 is the scalar loop (`iA11 += (float)(ixval * ixval)`).
//               lanes = 4 / 8 / 16 model the SIMD loops: per window row, full groups of (pairs ? 2 : 1) * lanes elements are added lane-wise
//               into `lanes` binary32 accumulators -- with pairs, adjacent products are first summed exactly in int32, as v_dotprod does for
//               the int16 operands (the universal-intrinsic loop: v_int16x8, 8 elements per step, 4 float lanes) --, the rest of the row goes
//               to a scalar binary32 accumulator in order; the result is scalar + (lane 0 + lane 1 + ...), as v_reduce_sum adds them.
struct WinSum
{
    int lanes, pairs;
    long long exact = 0;
    float scalar = 0.f, lane[16] = {0.f};
    int row_fill = 0; long long pending = 0; int pending_n = 0;
    WinSum(int lanes_, int pairs_) : lanes(lanes_), pairs(pairs_) {}
    // one window row of products, in order
    void add_row(const long long* prod, int n)
    {
        if (lanes == 0) { for (int i = 0; i < n; i++) exact += prod[i]; return; }
        if (lanes == 1) { for (int i = 0; i < n; i++) scalar += (float)prod[i]; return; }
        const int per = pairs ? 2 : 1, group = per * lanes;
        int i = 0;
        for (; i + group <= n; i += group)
            for (int l = 0; l < lanes; l++)
            {
                long long v = prod[i + per * l];
                if (pairs) v += prod[i + per * l + 1];
                lane[l] += (float)v;
            }
        for (; i < n; i++) scalar += (float)prod[i];
    }
    float total(float scale) const
    {
        if (lanes == 0) return (float)exact * scale;
        float r = 0.f;
        if (lanes > 1) { for (int l = 0; l < lanes; l++) r += lane[l]; }
        return (scalar + r) * scale;
    }
};

} // namespace

namespace {

// Returns the number of pyramid levels used minus one (the effective maxLevel), or < 0 on error.
int pyrlk_impl(const uint8_t* prev, int prev_step, const uint8_t* next, int next_step, int rows, int cols,
               const float* prev_pts, int n, float* next_pts, uint8_t* status,
               int win_w, int win_h, int max_level, int max_count, double epsilon, double min_eig_threshold, int acc_lanes, int acc_pairs)
{
    if (!prev || !next || n < 0) return -1;
    // SparsePyrLKOpticalFlowImpl: criteria clamp, epsilon squared
    max_count = std::min(std::max(max_count, 0), 100);
    epsilon = std::min(std::max(epsilon, 0.), 10.);
    epsilon *= epsilon;

    const std::vector<Level> P = build_pyramid(prev, prev_step, rows, cols, max_level, win_w, win_h, true);
    const std::vector<Level> N = build_pyramid(next, next_step, rows, cols, max_level, win_w, win_h, false);
    const int top = (int)P.size() - 1;

    for (int i = 0; i < n; i++) status[i] = 1;
    const float halfx = (win_w - 1) * 0.5f, halfy = (win_h - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);

    for (int level = top; level >= 0; level--)
    {
        const Level& I = P[level];
        const Level& J = N[level];
        // the points are independent of each other (OpenCV runs this loop under parallel_for_ as well)
        lvko_parallel_for(n, 16, [&](int pt0, int pt1) {
        std::vector<int> Iw((size_t)win_w * win_h), Ixw((size_t)win_w * win_h), Iyw((size_t)win_w * win_h);
        std::vector<long long> r0((size_t)win_w), r1((size_t)win_w), r2((size_t)win_w);
        for (int pt = pt0; pt < pt1; pt++)
        {
            float px = prev_pts[2 * pt] * (float)(1. / (1 << level));
            float py = prev_pts[2 * pt + 1] * (float)(1. / (1 << level));
            float nx, ny;
            if (level == top) { nx = px; ny = py; }
            else { nx = next_pts[2 * pt] * 2.f; ny = next_pts[2 * pt + 1] * 2.f; }
            next_pts[2 * pt] = nx; next_pts[2 * pt + 1] = ny;

            px -= halfx; py -= halfy;
            const int ipx = (int)std::floor(px), ipy = (int)std::floor(py);
            if (ipx < -win_w || ipx >= I.cols || ipy < -win_h || ipy >= I.rows)
            {
                if (level == 0) status[pt] = 0;
                continue;
            }
            float a = px - ipx, b = py - ipy;
            int iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
            int iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

            WinSum sA11(acc_lanes, acc_pairs), sA12(acc_lanes, acc_pairs), sA22(acc_lanes, acc_pairs);
            for (int y = 0; y < win_h; y++)
            {
                for (int x = 0; x < win_w; x++)
                {
                    const int yy = ipy + y, xx = ipx + x;
                    const int ival = descale(I.at(yy, xx) * iw00 + I.at(yy, xx + 1) * iw01 + I.at(yy + 1, xx) * iw10 + I.at(yy + 1, xx + 1) * iw11, W_BITS - 5);
                    const int ixval = descale(I.dat(yy, xx, 0) * iw00 + I.dat(yy, xx + 1, 0) * iw01 + I.dat(yy + 1, xx, 0) * iw10 + I.dat(yy + 1, xx + 1, 0) * iw11, W_BITS);
                    const int iyval = descale(I.dat(yy, xx, 1) * iw00 + I.dat(yy, xx + 1, 1) * iw01 + I.dat(yy + 1, xx, 1) * iw10 + I.dat(yy + 1, xx + 1, 1) * iw11, W_BITS);
                    Iw[(size_t)y * win_w + x] = (int16_t)ival;
                    Ixw[(size_t)y * win_w + x] = (int16_t)ixval;
                    Iyw[(size_t)y * win_w + x] = (int16_t)iyval;
                    r0[(size_t)x] = (long long)ixval * ixval;
                    r1[(size_t)x] = (long long)ixval * iyval;
                    r2[(size_t)x] = (long long)iyval * iyval;
                }
                sA11.add_row(r0.data(), win_w); sA12.add_row(r1.data(), win_w); sA22.add_row(r2.data(), win_w);
            }
            const float A11 = sA11.total(FLT_SCALE), A12 = sA12.total(FLT_SCALE), A22 = sA22.total(FLT_SCALE);
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win_w * win_h);
            if (minEig < (float)min_eig_threshold || D < 1.1920928955078125e-07f /* FLT_EPSILON */)
            {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= halfx; ny -= halfy;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_count; j++)
            {
                const int inx = (int)std::floor(nx), iny = (int)std::floor(ny);
                if (inx < -win_w || inx >= J.cols || iny < -win_h || iny >= J.rows)
                {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                a = nx - inx; b = ny - iny;
                iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
                iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                WinSum sb1(acc_lanes, acc_pairs), sb2(acc_lanes, acc_pairs);
                for (int y = 0; y < win_h; y++)
                {
                    for (int x = 0; x < win_w; x++)
                    {
                        const int yy = iny + y, xx = inx + x;
                        const int diff = descale(J.at(yy, xx) * iw00 + J.at(yy, xx + 1) * iw01 + J.at(yy + 1, xx) * iw10 + J.at(yy + 1, xx + 1) * iw11, W_BITS - 5)
                                         - Iw[(size_t)y * win_w + x];
                        r0[(size_t)x] = (long long)diff * Ixw[(size_t)y * win_w + x];
                        r1[(size_t)x] = (long long)diff * Iyw[(size_t)y * win_w + x];
                    }
                    sb1.add_row(r0.data(), win_w); sb2.add_row(r1.data(), win_w);
                }
                const float b1 = sb1.total(FLT_SCALE), b2 = sb2.total(FLT_SCALE);
                const float dx = (A12 * b2 - A22 * b1) * D;
                const float dy = (A12 * b1 - A11 * b2) * D;
                nx += dx; ny += dy;
                next_pts[2 * pt] = nx + halfx; next_pts[2 * pt + 1] = ny + halfy;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01)
                {
                    next_pts[2 * pt] -= dx * 0.5f; next_pts[2 * pt + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
        }
        });
    }
    return top;
}

} // namespace

extern "C" {

// The specification (exact integer window sums).
int lvko_pyrlk(const uint8_t* prev, int prev_step, const uint8_t* next, int next_step, int rows, int cols,
               const float* prev_pts, int n, float* next_pts, uint8_t* status,
               int win_w, int win_h, int max_level, int max_count, double epsilon, double min_eig_threshold)
{
    return pyrlk_impl(prev, prev_step, next, next_step, rows, cols, prev_pts, n, next_pts, status, win_w, win_h, max_level, max_count, epsilon,
                      min_eig_threshold, 0, 0);
}

// The same tracker with OpenCV's binary32 window sums (see WinSum): lanes = 1 scalar loop, 4 / 8 / 16 SIMD lane counts; pairs != 0: adjacent
// products pre-added exactly (v_dotprod).  NOT the specification -- tests/test_pyrlk_float_order.py measures the distance between the two.
int lvko_pyrlk_float(const uint8_t* prev, int prev_step, const uint8_t* next, int next_step, int rows, int cols,
                     const float* prev_pts, int n, float* next_pts, uint8_t* status,
                     int win_w, int win_h, int max_level, int max_count, double epsilon, double min_eig_threshold, int lanes, int pairs)
{
    if (!(lanes == 1 || lanes == 4 || lanes == 8 || lanes == 16)) return -1;
    return pyrlk_impl(prev, prev_step, next, next_step, rows, cols, prev_pts, n, next_pts, status, win_w, win_h, max_level, max_count, epsilon,
                      min_eig_threshold, lanes, pairs ? 1 : 0);
}

// Level geometry helper for tests: writes rows/cols of each level, returns the level count.
int lvko_pyramid_levels(int rows, int cols, int max_level, int win_w, int win_h, int* out_rows, int* out_cols)
{
    int n = 0;
    out_rows[n] = rows; out_cols[n] = cols; n++;
    for (int level = 1; level <= max_level; level++)
    {
        const int r = (out_rows[n - 1] + 1) / 2, c = (out_cols[n - 1] + 1) / 2;
        if (c <= win_w || r <= win_h) break;
        out_rows[n] = r; out_cols[n] = c; n++;
    }
    return n;
}

} // extern "C"
