// Tracker image operations for gfx950: luma extraction fused with the INTER_AREA downscale to the tracking
// resolution, the pyrDown pyramid and the Scharr derivative images used by the sparse optical flow.
//
// Replaces (reference call sites; the arithmetic is OpenCV 4.8.0's):
//   VideoFrame::viewAsFormat(GRAY) for YUV frames            LiveVisionKit/Data/VideoFrame.cpp:260
//   cv::resize(gray, detection_resolution, INTER_AREA)        LiveVisionKit/Vision/FrameTracker.cpp:117
//   buildOpticalFlowPyramid / calcScharrDeriv inside calc()   LiveVisionKit/Vision/FrameTracker.cpp:140-146
// All integer except the non-integer-scale INTER_AREA path (binary32, no contraction, table order).
#include "lvk_hip_internal.hpp"

#include <cmath>
#include <algorithm>

namespace {

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { p = (p < 0) ? -p : 2 * (len - 1) - p; }
    return p;
}

__device__ __forceinline__ uint8_t sat_u8_rint(float v)
{
    const float r = __builtin_rintf(v);                      // v_rndne_f32: half to even, like cvRound
    return (uint8_t)(int)__builtin_fminf(__builtin_fmaxf(r, 0.0f), 255.0f);
}

// gray value of one source pixel: channel >= 0 extracts it (YUV: channel 0, VideoFrame.cpp:260); -1 / -2 = cv::cvtColor
// BGR2GRAY / RGB2GRAY (VideoFrame.cpp:194; OpenCV 4.8 RGB2Gray<uchar>, 15-bit fixed point, round to nearest)
__device__ __forceinline__ int gray_of(const uint8_t* __restrict__ p, int channel)
{
    if (channel >= 0) return p[channel];
    const int b = channel == -1 ? p[0] : p[2], g = p[1], r = channel == -1 ? p[2] : p[0];
    return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15;
}

// ---- INTER_AREA, integer scale (resizeAreaFast_): box sum * (1.f/area), round half to even; 2x2 -> (s+2)>>2 ----
__global__ __launch_bounds__(256)
void k_area_fast(const uint8_t* __restrict__ src, int src_step, int pix_stride, int channel,
                 uint8_t* __restrict__ dst, int dst_step, int drows, int dcols, int sx, int sy)
{
    LVK_TRACKER_PRIORITY();
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dcols || y >= drows) return;
    const uint8_t* p = src + (long)(y * sy) * src_step + (long)(x * sx) * pix_stride;
    int sum = 0;
    for (int ky = 0; ky < sy; ky++, p += src_step)
        for (int kx = 0; kx < sx; kx++)
            sum += gray_of(p + kx * pix_stride, channel);
    uint8_t out;
    if (sx == 2 && sy == 2) out = (uint8_t)((sum + 2) >> 2);
    else out = sat_u8_rint((float)sum * (1.f / (float)(sx * sy)));
    dst[(long)y * dst_step + x] = out;
}

// ---- INTER_AREA towards a LARGER image (a frame smaller than the detection resolution on either axis) -----------------------------------
// cv::resize emulates it "using some variant of bilinear interpolation": two taps per axis with AREA coefficients in 11-bit fixed point
// (imgproc/resize.cpp: cv::hal::resize with area_mode, HResizeLinear, VResizeLinear<uchar>); the tables hold, per destination index, the
// two source indices (the second one clamped) and the two weights -- see build_enlarge_tab.
__global__ __launch_bounds__(256)
void k_area_enlarge(const uint8_t* __restrict__ src, int src_step, int pix_stride, int channel,
                    uint8_t* __restrict__ dst, int dst_step, int drows, int dcols, const int4* __restrict__ xtab, const int4* __restrict__ ytab)
{
    LVK_TRACKER_PRIORITY();
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dcols || y >= drows) return;
    const int4 tx = xtab[x], ty = ytab[y];
    const uint8_t* r0 = src + (long)ty.x * src_step; const uint8_t* r1 = src + (long)ty.y * src_step;
    const int h0 = gray_of(r0 + (long)tx.x * pix_stride, channel) * tx.z + gray_of(r0 + (long)tx.y * pix_stride, channel) * tx.w;
    const int h1 = gray_of(r1 + (long)tx.x * pix_stride, channel) * tx.z + gray_of(r1 + (long)tx.y * pix_stride, channel) * tx.w;
    dst[(long)y * dst_step + x] = (uint8_t)((((ty.z * (h0 >> 4)) >> 16) + ((ty.w * (h1 >> 4)) >> 16) + 2) >> 2);
}

// Same arithmetic, specialised for the stabilizer's cases (4K: 8x8, 1080p: 4x4; packed 8UC3 channel 0 or planar): every
// source row segment of a destination pixel is SX * PIX contiguous bytes = whole aligned dwords, so the SY * SX * PIX / 4
// loads of a thread are independent and issued back to back (the generic kernel's byte loads in a runtime-bound loop
// serialise on memory latency: rocprofv3 showed 88 % of its wave time in s_waitcnt).
template <int SX, int SY, int PIX, int MODE = 0>          // MODE 0: channel 0; 1: BGR -> gray; 2: RGB -> gray (PIX == 3)
__global__ __launch_bounds__(256)
void k_area_fast_dw(const uint8_t* __restrict__ src, int src_step, uint8_t* __restrict__ dst, int dst_step, int drows, int dcols)
{
    LVK_TL(0);
    LVK_TRACKER_PRIORITY();
    constexpr int NW = SX * PIX / 4;
    static_assert((SX * PIX) % 4 == 0, "row segment must be whole dwords");
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dcols || y >= drows) return;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (long)(y * SY) * src_step) + (long)x * NW;
    uint32_t w[SY][NW];
#pragma unroll
    for (int ky = 0; ky < SY; ky++)
#pragma unroll
        for (int k = 0; k < NW; k++)
            w[ky][k] = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(p) + (long)ky * src_step)[k];
    int sum = 0;
#pragma unroll
    for (int ky = 0; ky < SY; ky++)
#pragma unroll
        for (int kx = 0; kx < SX; kx++)
        {
            const int b = kx * PIX;                       // byte of channel 0 of source pixel kx inside the segment
            const int c0 = (int)((w[ky][b >> 2] >> ((b & 3) * 8)) & 0xffu);
            if (MODE == 0) sum += c0;
            else
            {
                const int c1 = (int)((w[ky][(b + 1) >> 2] >> (((b + 1) & 3) * 8)) & 0xffu);
                const int c2 = (int)((w[ky][(b + 2) >> 2] >> (((b + 2) & 3) * 8)) & 0xffu);
                const int bl = MODE == 1 ? c0 : c2, rd = MODE == 1 ? c2 : c0;
                sum += (bl * 3735 + c1 * 19235 + rd * 9798 + (1 << 14)) >> 15;
            }
        }
    uint8_t out;
    if (SX == 2 && SY == 2) out = (uint8_t)((sum + 2) >> 2);
    else out = sat_u8_rint((float)sum * (1.f / (float)(SX * SY)));
    dst[(long)y * dst_step + x] = out;
}

// ---- INTER_AREA, general scale (resizeArea_ + computeResizeAreaTab) ----
__global__ __launch_bounds__(256)
void k_area_general(const uint8_t* __restrict__ src, int src_step, int pix_stride, int channel,
                    uint8_t* __restrict__ dst, int dst_step, int drows, int dcols,
                    const int2* __restrict__ xrange, const AreaTabEntry* __restrict__ xtab,
                    const int2* __restrict__ yrange, const AreaTabEntry* __restrict__ ytab)
{
    LVK_TRACKER_PRIORITY();
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dcols || y >= drows) return;
    const int2 xr = xrange[x], yr = yrange[y];
    float sum = 0.0f;
    for (int j = yr.x; j < yr.x + yr.y; j++)
    {
        const uint8_t* row = src + (long)ytab[j].si * src_step;
        float buf = 0.0f;
        for (int k = xr.x; k < xr.x + xr.y; k++)
            buf = buf + (float)gray_of(row + (long)xtab[k].si * pix_stride, channel) * xtab[k].alpha;
        sum = sum + ytab[j].alpha * buf;
    }
    dst[(long)y * dst_step + x] = sat_u8_rint(sum);
}

// The same sums in the same order with every load in flight at once: up to MAXT taps per axis, the tap lists and the MAXT x MAXT source
// bytes are fetched first (clamped indices, no data-dependent trip counts), then the float chain runs over registers.  A tap beyond a
// list's end contributes v * 0.0f, and x + 0.0f == x for the non-negative sums here, so the result is bit-identical to k_area_general --
// whose byte loads inside two runtime-bound loops serialise on memory latency: 34 us for 2560x1440 -> 480x270, 27 us for 1920x1200,
// against 5-8 us for the integer scales.
template <int MAXT>
__global__ __launch_bounds__(256)
void k_area_general_taps(const uint8_t* __restrict__ src, int src_step, int pix_stride, int channel,
                         uint8_t* __restrict__ dst, int dst_step, int drows, int dcols,
                         const int2* __restrict__ xrange, const AreaTabEntry* __restrict__ xtab,
                         const int2* __restrict__ yrange, const AreaTabEntry* __restrict__ ytab)
{
    LVK_TRACKER_PRIORITY();
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dcols || y >= drows) return;
    const int2 xr = xrange[x], yr = yrange[y];
    int xs[MAXT], ys[MAXT]; float xa[MAXT], ya[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++)
    {
        const AreaTabEntry ex = xtab[xr.x + min(k, xr.y - 1)], ey = ytab[yr.x + min(k, yr.y - 1)];
        xs[k] = ex.si * pix_stride + channel; xa[k] = k < xr.y ? ex.alpha : 0.0f;
        ys[k] = ey.si; ya[k] = k < yr.y ? ey.alpha : 0.0f;
    }
    uint8_t v[MAXT][MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; j++)
    {
        const uint8_t* row = src + (long)ys[j] * src_step;
#pragma unroll
        for (int k = 0; k < MAXT; k++) v[j][k] = row[xs[k]];
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXT; j++)
    {
        float buf = 0.0f;
#pragma unroll
        for (int k = 0; k < MAXT; k++) buf = buf + (float)v[j][k] * xa[k];
        sum = sum + ya[j] * buf;
    }
    dst[(long)y * dst_step + x] = sat_u8_rint(sum);
}

// The same again for a PLANAR source with the block's source window staged in LDS: a 64 x 4 tile of the destination reads a window of at most
// AT_W x AT_H source bytes, fetched once with coalesced dword loads (the per-thread byte loads of the kernel above touch every cache line
// ~5 times); the taps then come out of LDS.  Same sums in the same order.  2560x1440 -> 480x270 next to the remap: 17.7 -> 12.7 us (8.0 best): three
// dependent round trips (tap ranges, tap lists + window corners, window) are what is left.
constexpr int AT_W = 576, AT_H = 40;                     // window bound: 64 * 8 + slack bytes wide (incl. the dword alignment), 4 * 8 + slack rows
template <int MAXT>
__global__ __launch_bounds__(256)
void k_area_general_tile(const uint8_t* __restrict__ src, int src_step, int scols,
                         uint8_t* __restrict__ dst, int dst_step, int drows, int dcols,
                         const int2* __restrict__ xrange, const AreaTabEntry* __restrict__ xtab,
                         const int2* __restrict__ yrange, const AreaTabEntry* __restrict__ ytab)
{
    LVK_TRACKER_PRIORITY();
    __shared__ __attribute__((aligned(16))) uint8_t tile[AT_H * AT_W];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int bx0 = blockIdx.x * 64, by0 = blockIdx.y * 4;
    const int bx1 = min(bx0 + 63, dcols - 1), by1 = min(by0 + 3, drows - 1);
    const int x = bx0 + threadIdx.x, y = by0 + threadIdx.y;
    // round trip 1: the tap ranges of the block's corners (its source window) and of this thread's pixel, together
    const int2 xa = xrange[bx0], xb = xrange[bx1], ya = yrange[by0], yb = yrange[by1];
    const int2 xr = xrange[min(x, dcols - 1)], yr = yrange[min(y, drows - 1)];
    // round trip 2: the window's corner taps and this thread's tap lists (tap lists are ascending in si)
    const int sx0 = xtab[xa.x].si & ~3, sx1 = xtab[xb.x + xb.y - 1].si;
    const int sy0 = ytab[ya.x].si, sy1 = ytab[yb.x + yb.y - 1].si;
    AreaTabEntry ex[MAXT], ey[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++) { ex[k] = xtab[xr.x + min(k, xr.y - 1)]; ey[k] = ytab[yr.x + min(k, yr.y - 1)]; }
    // round trip 3: the window, coalesced dwords
    const int wd = (sx1 - sx0 + 4) >> 2, ht = sy1 - sy0 + 1;                     // dwords per row, rows
    for (int i = tid; i < wd * ht; i += 256)
    {
        const int r = i / wd, c = i - r * wd;
        const uint8_t* p = src + (long)(sy0 + r) * src_step + sx0 + 4 * c;
        uint32_t v;
        if (sx0 + 4 * c + 3 < scols && ((reinterpret_cast<uintptr_t>(p) & 3u) == 0)) v = *reinterpret_cast<const uint32_t*>(p);
        else
        {
            v = 0;
            for (int b = 0; b < 4; b++) if (sx0 + 4 * c + b < scols) v |= (uint32_t)p[b] << (8 * b);
        }
        *reinterpret_cast<uint32_t*>(&tile[r * AT_W + 4 * c]) = v;
    }
    __syncthreads();
    if (x >= dcols || y >= drows) return;
    int xs[MAXT], ys[MAXT]; float xw[MAXT], yw[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++)
    {
        xs[k] = ex[k].si - sx0; xw[k] = k < xr.y ? ex[k].alpha : 0.0f;
        ys[k] = (ey[k].si - sy0) * AT_W; yw[k] = k < yr.y ? ey[k].alpha : 0.0f;
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXT; j++)
    {
        float buf = 0.0f;
#pragma unroll
        for (int k = 0; k < MAXT; k++) buf = buf + (float)tile[ys[j] + xs[k]] * xw[k];
        sum = sum + yw[j] * buf;
    }
    dst[(long)y * dst_step + x] = sat_u8_rint(sum);
}

// ---- cv::pyrDown 8UC1 BORDER_REFLECT_101 ----
__global__ __launch_bounds__(256)
void k_pyr_down(const uint8_t* __restrict__ src, int src_step, int rows, int cols,
                uint8_t* __restrict__ dst, int dst_step, int drows, int dcols)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dcols || y >= drows) return;
    int xi[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xi[k] = reflect101(2 * x - 2 + k, cols);
    int acc = 0;
#pragma unroll
    for (int ky = 0; ky < 5; ky++)
    {
        const uint8_t* row = src + (long)reflect101(2 * y - 2 + ky, rows) * src_step;
        const int h = row[xi[0]] + row[xi[4]] + 4 * (row[xi[1]] + row[xi[3]]) + 6 * row[xi[2]];
        const int w = (ky == 0 || ky == 4) ? 1 : (ky == 2 ? 6 : 4);
        acc += w * h;
    }
    dst[(long)y * dst_step + x] = (uint8_t)((acc + 128) >> 8);
}

// ---- calcScharrDeriv: (Ix, Iy) int16 interleaved, reflect-101 at the image edge ----
__global__ __launch_bounds__(256)
void k_scharr(const uint8_t* __restrict__ src, int src_step, int rows, int cols, short2* __restrict__ dst)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const uint8_t* r0 = src + (long)(y > 0 ? y - 1 : (rows > 1 ? 1 : 0)) * src_step;
    const uint8_t* r1 = src + (long)y * src_step;
    const uint8_t* r2 = src + (long)(y < rows - 1 ? y + 1 : (rows > 1 ? rows - 2 : 0)) * src_step;
    const int xm = x > 0 ? x - 1 : (cols > 1 ? 1 : 0);
    const int xp = x < cols - 1 ? x + 1 : (cols > 1 ? cols - 2 : 0);
    // vertical pass at columns xm, x, xp
    const int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10, t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
    const int t1m = r2[xm] - r0[xm], t1c = r2[x] - r0[x], t1p = r2[xp] - r0[xp];
    short2 o;
    o.x = (short)(t0p - t0m);
    o.y = (short)((t1p + t1m) * 3 + t1c * 10);
    dst[(long)y * cols + x] = o;
}

// ---- pyramid levels 1..3 in ONE launch -----------------------------------------------------------------------------
// The pyramid images are tiny (240x135, 120x68, 60x34 for the 480x270 tracking frame), so seven dependent launches
// (3 x pyrDown + 4 x Scharr) cost far more in launch gaps than in work.  Here a block owns an 8x8 tile of level 3 and
// the matching 16x16 / 32x32 tiles of levels 2 / 1: it stages the 85x85 window of level 0 those need in LDS, computes
// the 41x41 level-1 window (halo recomputed redundantly by neighbouring blocks), from it the 19x19 level-2 window and
// from that its level-3 tile, writing only the pixels it owns.  Same integer arithmetic as k_pyr_down
// ((sum + 128) >> 8, reflect-101 applied on each level's own index range).
constexpr int T3 = 4;                                   // level-3 tile edge owned by a block (level 2: 2*T3, level 1: 4*T3)
constexpr int F2 = 2 * (T3 - 1) + 5, F1 = 2 * (F2 - 1) + 5, F0 = 2 * (F1 - 1) + 5;     // 11, 25, 53: windows incl. filter halo
constexpr int F0P = F0 + 3, F1P = F1 + 3, F2P = F2 + 1;

__device__ __forceinline__ int pyr_tap(const uint8_t* t, int pitch, int ox, int oy, int x, int y, int cols, int rows)
{
    // 5x5 [1 4 6 4 1] around (2x, 2y) of the source level (size cols x rows), samples fetched from tile t whose (0,0) is (ox, oy)
    int xi[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xi[k] = reflect101(2 * x - 2 + k, cols) - ox;
    int acc = 0;
#pragma unroll
    for (int ky = 0; ky < 5; ky++)
    {
        const uint8_t* row = t + (reflect101(2 * y - 2 + ky, rows) - oy) * pitch;
        const int h = row[xi[0]] + row[xi[4]] + 4 * (row[xi[1]] + row[xi[3]]) + 6 * row[xi[2]];
        acc += ((ky == 0 || ky == 4) ? 1 : (ky == 2 ? 6 : 4)) * h;
    }
    return (acc + 128) >> 8;
}

__global__ __launch_bounds__(256)
void k_pyr_fused3(PyrArgs a)
{
    LVK_TL(1);
    LVK_TRACKER_PRIORITY();
    __shared__ uint8_t t0[F0 * F0P], t1[F1 * F1P], t2[F2 * F2P];
    const int tid = threadIdx.x;
    const int x3 = blockIdx.x * T3, y3 = blockIdx.y * T3;
    const int o2x = 2 * x3 - 2, o2y = 2 * y3 - 2, o1x = 2 * o2x - 2, o1y = 2 * o2y - 2, o0x = 2 * o1x - 2, o0y = 2 * o1y - 2;
    const PyrLevel L0 = a.lv[0], L1 = a.lv[1], L2 = a.lv[2], L3 = a.lv[3];

    // All of a thread's level-0 loads are issued before the first is consumed (clamped addresses, no branches): one global round
    // trip for the window instead of one per 256 bytes -- the loop form spent 11 dependent round trips here, 7 of the kernel's 10 us.
    {
        constexpr int N0 = (F0 * F0 + 255) / 256;
        uint8_t v[N0];
#pragma unroll
        for (int k = 0; k < N0; k++)
        {
            const int i = min(tid + 256 * k, F0 * F0 - 1);
            const int ty = i / F0, tx = i - ty * F0;
            const int gx = min(max(o0x + tx, 0), L0.cols - 1), gy = min(max(o0y + ty, 0), L0.rows - 1);
            v[k] = L0.img[(long)gy * L0.step + gx];
        }
#pragma unroll
        for (int k = 0; k < N0; k++)
        {
            const int i = tid + 256 * k;
            const int ty = i / F0, tx = i - ty * F0;
            const int gx = o0x + tx, gy = o0y + ty;
            if (i < F0 * F0 && gx >= 0 && gy >= 0 && gx < L0.cols && gy < L0.rows) t0[ty * F0P + tx] = v[k];
        }
    }
    __syncthreads();
    for (int i = tid; i < F1 * F1; i += 256)
    {
        const int ty = i / F1, tx = i - ty * F1;
        const int gx = o1x + tx, gy = o1y + ty;
        if (gx >= 0 && gy >= 0 && gx < L1.cols && gy < L1.rows)
        {
            const int v = pyr_tap(t0, F0P, o0x, o0y, gx, gy, L0.cols, L0.rows);
            t1[ty * F1P + tx] = (uint8_t)v;
            if (gx >= 4 * x3 && gx < 4 * x3 + 4 * T3 && gy >= 4 * y3 && gy < 4 * y3 + 4 * T3)
                const_cast<uint8_t*>(L1.img)[(long)gy * L1.step + gx] = (uint8_t)v;
        }
    }
    __syncthreads();
    for (int i = tid; i < F2 * F2; i += 256)
    {
        const int ty = i / F2, tx = i - ty * F2;
        const int gx = o2x + tx, gy = o2y + ty;
        if (gx >= 0 && gy >= 0 && gx < L2.cols && gy < L2.rows)
        {
            const int v = pyr_tap(t1, F1P, o1x, o1y, gx, gy, L1.cols, L1.rows);
            t2[ty * F2P + tx] = (uint8_t)v;
            if (gx >= 2 * x3 && gx < 2 * x3 + 2 * T3 && gy >= 2 * y3 && gy < 2 * y3 + 2 * T3)
                const_cast<uint8_t*>(L2.img)[(long)gy * L2.step + gx] = (uint8_t)v;
        }
    }
    __syncthreads();
    if (tid < T3 * T3)
    {
        const int gx = x3 + (tid % T3), gy = y3 + (tid / T3);
        if (gx < L3.cols && gy < L3.rows)
            const_cast<uint8_t*>(L3.img)[(long)gy * L3.step + gx] = (uint8_t)pyr_tap(t2, F2P, o2x, o2y, gx, gy, L2.cols, L2.rows);
    }
}

// Scharr derivative images of all pyramid levels in one launch (blockIdx.z = level).
__global__ __launch_bounds__(256)
void k_scharr_all(PyrArgs a)
{
    const PyrLevel L = a.lv[blockIdx.z];
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= L.cols || y >= L.rows) return;
    const int rows = L.rows, cols = L.cols;
    const uint8_t* r0 = L.img + (long)(y > 0 ? y - 1 : (rows > 1 ? 1 : 0)) * L.step;
    const uint8_t* r1 = L.img + (long)y * L.step;
    const uint8_t* r2 = L.img + (long)(y < rows - 1 ? y + 1 : (rows > 1 ? rows - 2 : 0)) * L.step;
    const int xm = x > 0 ? x - 1 : (cols > 1 ? 1 : 0);
    const int xp = x < cols - 1 ? x + 1 : (cols > 1 ? cols - 2 : 0);
    const int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10, t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
    const int t1m = r2[xm] - r0[xm], t1c = r2[x] - r0[x], t1p = r2[xp] - r0[xp];
    short2 o;
    o.x = (short)(t0p - t0m);
    o.y = (short)((t1p + t1m) * 3 + t1c * 10);
    const_cast<short2*>(L.deriv)[(long)y * cols + x] = o;
}

// imgproc/resize.cpp computeResizeAreaTab, grouped per destination index.
void build_area_tab(int ssize, int dsize, std::vector<int2>& range, std::vector<AreaTabEntry>& tab)
{
    const double scale = (double)ssize / dsize;
    range.resize(dsize);
    tab.clear();
    for (int dx = 0; dx < dsize; dx++)
    {
        const int start = (int)tab.size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; sx++) tab.push_back({sx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3) tab.push_back({sx2, (float)(std::min(std::min(fsx2 - sx2, 1.0), cell) / cell)});
        range[dx] = make_int2(start, (int)tab.size() - start);
    }
}

// cv::hal::resize, area_mode with ksize = 2: s = floor(d * scale), f = (d + 1) - (s + 1) * inv_scale, f = f <= 0 ? 0 : f - floor(f);
// weights saturate_cast<short>((1 - f) * 2048), saturate_cast<short>(f * 2048); from the first index whose second tap would leave the
// source on, the single tap S[ssize - 1] * 2048 (HResizeLinear's tail; the rows are clamped the same way, with their weights kept).
void build_enlarge_tab(int ssize, int dsize, bool horizontal, std::vector<int4>& tab)
{
    const double inv = (double)dsize / ssize, scale = 1. / inv;
    auto to_short = [](float v) -> int { const long r = lrintf(v); return (int)(r < -32768 ? -32768 : r > 32767 ? 32767 : r); };
    tab.resize((size_t)dsize);
    bool tail = false;
    for (int d = 0; d < dsize; d++)
    {
        int s0 = (int)std::floor(d * scale);
        float f = (float)((d + 1) - (s0 + 1) * inv);
        f = f <= 0 ? 0.f : f - std::floor(f);
        if (horizontal)
        {
            if (s0 < 0) { f = 0; s0 = 0; }
            if (s0 + 1 >= ssize) { tail = true; if (s0 >= ssize - 1) { f = 0; s0 = ssize - 1; } }
            const int w0 = tail ? 2048 : to_short((1.f - f) * 2048.f), w1 = tail ? 0 : to_short(f * 2048.f);
            tab[(size_t)d] = make_int4(s0, std::min(s0 + 1, ssize - 1), w0, w1);
        }
        else
            tab[(size_t)d] = make_int4(std::min(std::max(s0, 0), ssize - 1), std::min(std::max(s0 + 1, 0), ssize - 1), to_short((1.f - f) * 2048.f), to_short(f * 2048.f));
    }
}

} // namespace

static int lvk_get_enlargetab(lvk_hip_ctx* ctx, int ssize, int dsize, bool horizontal, const int4** d_tab)
{
    const auto key = std::make_pair(horizontal ? ssize : -ssize, dsize);
    auto it = ctx->enlargetabs.find(key);
    if (it == ctx->enlargetabs.end())
    {
        std::vector<int4> tab;
        build_enlarge_tab(ssize, dsize, horizontal, tab);
        int4* dev = nullptr;
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&dev, tab.size() * sizeof(int4)));
        LVK_HIP_CHECK(ctx, hipMemcpy(dev, tab.data(), tab.size() * sizeof(int4), hipMemcpyHostToDevice));
        it = ctx->enlargetabs.emplace(key, dev).first;
    }
    *d_tab = it->second;
    return LVK_HIP_OK;
}

int lvk_get_areatab(lvk_hip_ctx* ctx, int ssize, int dsize, const int2** d_range, const AreaTabEntry** d_tab, int* max_taps, int* span64, int* span4)
{
    const auto key = std::make_pair(ssize, dsize);
    auto it = ctx->areatabs.find(key);
    if (it == ctx->areatabs.end())
    {
        std::vector<int2> range; std::vector<AreaTabEntry> tab;
        build_area_tab(ssize, dsize, range, tab);
        lvk_hip_ctx::AreaTabDev dev;
        for (const int2& r : range) dev.max_taps = std::max(dev.max_taps, r.y);
        for (int group : {64, 4})
            for (int d0 = 0; d0 < dsize; d0 += group)
            {
                const int d1 = std::min(d0 + group - 1, dsize - 1);
                const int span = tab[(size_t)range[d1].x + range[d1].y - 1].si - tab[(size_t)range[d0].x].si + 1;
                (group == 64 ? dev.span64 : dev.span4) = std::max(group == 64 ? dev.span64 : dev.span4, span);
            }
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&dev.range, range.size() * sizeof(int2)));
        LVK_HIP_CHECK(ctx, hipMalloc((void**)&dev.tab, tab.size() * sizeof(AreaTabEntry)));
        LVK_HIP_CHECK(ctx, hipMemcpy(dev.range, range.data(), range.size() * sizeof(int2), hipMemcpyHostToDevice));
        LVK_HIP_CHECK(ctx, hipMemcpy(dev.tab, tab.data(), tab.size() * sizeof(AreaTabEntry), hipMemcpyHostToDevice));
        it = ctx->areatabs.emplace(key, dev).first;
    }
    *d_range = it->second.range;
    *d_tab = it->second.tab;
    if (max_taps) *max_taps = it->second.max_taps;
    if (span64) *span64 = it->second.span64;
    if (span4) *span4 = it->second.span4;
    return LVK_HIP_OK;
}

int lvk_launch_luma_area_resize(lvk_hip_ctx* ctx, const void* d_src, int src_step, int pix_stride, int channel,
                                int srows, int scols, void* d_dst, int dst_step, int drows, int dcols)
{
    LVK_HIP_REQUIRE(ctx, d_src && d_dst && srows > 0 && scols > 0 && drows > 0 && dcols > 0);
    LVK_HIP_REQUIRE(ctx, pix_stride >= 1 && channel >= -2 && channel < pix_stride && (channel >= 0 || pix_stride >= 3));
    const dim3 block(64, 4), grid((dcols + 63) / 64, (drows + 3) / 4);
    if (drows > srows || dcols > scols)
    {
        // a frame smaller than the detection resolution on either axis (FrameTracker.cpp:117 resizes whatever it is given)
        const int4 *xt, *yt;
        int rc;
        if ((rc = lvk_get_enlargetab(ctx, scols, dcols, true, &xt)) != LVK_HIP_OK) return rc;
        if ((rc = lvk_get_enlargetab(ctx, srows, drows, false, &yt)) != LVK_HIP_OK) return rc;
        hipLaunchKernelGGL(k_area_enlarge, grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, pix_stride, channel, (uint8_t*)d_dst, dst_step, drows, dcols, xt, yt);
        LVK_HIP_CHECK(ctx, hipGetLastError());
        return LVK_HIP_OK;
    }
    const int isx = scols / dcols, isy = srows / drows;
    const bool exact = scols % dcols == 0 && srows % drows == 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(d_src) & 3u) == 0 && (src_step & 3) == 0;
    const bool dw_ok = exact && channel == 0 && aligned;
    const bool dw_rgb = exact && channel < 0 && aligned && pix_stride == 3;
    if (dw_rgb && isx == 8 && isy == 8 && channel == -1)
        hipLaunchKernelGGL((k_area_fast_dw<8, 8, 3, 1>), grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, (uint8_t*)d_dst, dst_step, drows, dcols);
    else if (dw_rgb && isx == 8 && isy == 8)
        hipLaunchKernelGGL((k_area_fast_dw<8, 8, 3, 2>), grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, (uint8_t*)d_dst, dst_step, drows, dcols);
    else if (dw_rgb && isx == 4 && isy == 4 && channel == -1)
        hipLaunchKernelGGL((k_area_fast_dw<4, 4, 3, 1>), grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, (uint8_t*)d_dst, dst_step, drows, dcols);
    else if (dw_rgb && isx == 4 && isy == 4)
        hipLaunchKernelGGL((k_area_fast_dw<4, 4, 3, 2>), grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, (uint8_t*)d_dst, dst_step, drows, dcols);
    else if (dw_ok && isx == 8 && isy == 8 && pix_stride == 3)
        hipLaunchKernelGGL((k_area_fast_dw<8, 8, 3>), grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, (uint8_t*)d_dst, dst_step, drows, dcols);
    else if (dw_ok && isx == 4 && isy == 4 && pix_stride == 3)
        hipLaunchKernelGGL((k_area_fast_dw<4, 4, 3>), grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, (uint8_t*)d_dst, dst_step, drows, dcols);
    else if (dw_ok && isx == 8 && isy == 8 && pix_stride == 1)
        hipLaunchKernelGGL((k_area_fast_dw<8, 8, 1>), grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, (uint8_t*)d_dst, dst_step, drows, dcols);
    else if (dw_ok && isx == 4 && isy == 4 && pix_stride == 1)
        hipLaunchKernelGGL((k_area_fast_dw<4, 4, 1>), grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, (uint8_t*)d_dst, dst_step, drows, dcols);
    else if (exact)
    {
        hipLaunchKernelGGL(k_area_fast, grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, pix_stride, channel,
                           (uint8_t*)d_dst, dst_step, drows, dcols, scols / dcols, srows / drows);
    }
    else
    {
        const int2 *xr, *yr; const AreaTabEntry *xt, *yt;
        int rc;
        int xtaps = 0, ytaps = 0, xspan = 0, yspan = 0, unused = 0;
        if ((rc = lvk_get_areatab(ctx, scols, dcols, &xr, &xt, &xtaps, &xspan, &unused)) != LVK_HIP_OK) return rc;
        if ((rc = lvk_get_areatab(ctx, srows, drows, &yr, &yt, &ytaps, &unused, &yspan)) != LVK_HIP_OK) return rc;
        const int taps = std::max(xtaps, ytaps);
        const bool tile_ok = pix_stride == 1 && channel == 0 && taps >= 1 && taps <= 8 && xspan + 6 <= AT_W && yspan <= AT_H;
        if (tile_ok && taps <= 4)
            hipLaunchKernelGGL(k_area_general_tile<4>, grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, scols, (uint8_t*)d_dst, dst_step, drows, dcols, xr, xt, yr, yt);
        else if (tile_ok)
            hipLaunchKernelGGL(k_area_general_tile<8>, grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, scols, (uint8_t*)d_dst, dst_step, drows, dcols, xr, xt, yr, yt);
        else if (channel >= 0 && taps >= 1 && taps <= 4)
            hipLaunchKernelGGL(k_area_general_taps<4>, grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, pix_stride, channel,
                               (uint8_t*)d_dst, dst_step, drows, dcols, xr, xt, yr, yt);
        else if (channel >= 0 && taps >= 1 && taps <= 8)
            hipLaunchKernelGGL(k_area_general_taps<8>, grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, pix_stride, channel,
                               (uint8_t*)d_dst, dst_step, drows, dcols, xr, xt, yr, yt);
        else
        hipLaunchKernelGGL(k_area_general, grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, pix_stride, channel,
                           (uint8_t*)d_dst, dst_step, drows, dcols, xr, xt, yr, yt);
    }
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

int lvk_launch_pyr_down(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step)
{
    LVK_HIP_REQUIRE(ctx, d_src && d_dst && rows > 0 && cols > 0);
    const int drows = (rows + 1) / 2, dcols = (cols + 1) / 2;
    const dim3 block(64, 4), grid((dcols + 63) / 64, (drows + 3) / 4);
    hipLaunchKernelGGL(k_pyr_down, grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, rows, cols, (uint8_t*)d_dst, dst_step, drows, dcols);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

int lvk_launch_scharr(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst)
{
    LVK_HIP_REQUIRE(ctx, d_src && d_dst && rows > 0 && cols > 0);
    const dim3 block(64, 4), grid((cols + 63) / 64, (rows + 3) / 4);
    hipLaunchKernelGGL(k_scharr, grid, block, 0, ctx->stream, (const uint8_t*)d_src, src_step, rows, cols, (short2*)d_dst);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

// Levels 1.. and all derivative images of a pyramid whose level 0 is filled: two launches for the usual 4-level pyramid.
int lvk_launch_pyramid(lvk_hip_ctx* ctx, const PyrArgs& args, bool derivs)
{
    int rc;
    if (args.nlevels == 4)
    {
        const dim3 grid((args.lv[3].cols + T3 - 1) / T3, (args.lv[3].rows + T3 - 1) / T3);
        hipLaunchKernelGGL(k_pyr_fused3, grid, dim3(256), 0, ctx->stream, args);
    }
    else
        for (int i = 1; i < args.nlevels; i++)
            if ((rc = lvk_launch_pyr_down(ctx, args.lv[i - 1].img, args.lv[i - 1].step, args.lv[i - 1].rows, args.lv[i - 1].cols,
                                          const_cast<uint8_t*>(args.lv[i].img), args.lv[i].step)) != LVK_HIP_OK) return rc;
    if (derivs)                               // (the flow kernel derives them from its staged windows; only the test entry asks for the images)
    {
        const dim3 sgrid((args.lv[0].cols + 63) / 64, (args.lv[0].rows + 3) / 4, args.nlevels);
        hipLaunchKernelGGL(k_scharr_all, sgrid, dim3(64, 4), 0, ctx->stream, args);
    }
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

extern "C" {

int lvk_hip_luma_area_resize(lvk_hip_ctx* ctx, const void* d_src, int src_step, int pix_stride, int channel,
                             int srows, int scols, void* d_dst, int dst_step, int drows, int dcols)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_luma_area_resize(ctx, d_src, src_step, pix_stride, channel, srows, scols, d_dst, dst_step, drows, dcols);
}

int lvk_hip_pyr_down(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_pyr_down(ctx, d_src, src_step, rows, cols, d_dst, dst_step);
}

int lvk_hip_scharr(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_scharr(ctx, d_src, src_step, rows, cols, d_dst);
}

} // extern "C"

LVK_TL_EXPORT(imgproc)
