// Pyramidal Lucas-Kanade sparse optical flow for gfx950: one block per feature and one 64-lane wavefront per pyramid
// level inside one launch (the levels' windows, derivatives and 2x2 systems are prepared concurrently, the iterations
// then run coarse to fine out of LDS), integer fixed-point bilinear sampling and exact integer wave reductions.
//
// Replaces cv::SparsePyrLKOpticalFlow::calc as the reference configures it (reference call:
// LiveVisionKit/Vision/FrameTracker.cpp:33-35,42-48,140-146; arithmetic: OpenCV 4.8.0 video/lkpyramid.cpp
// LKTrackerInvoker, CPU fixed-point path).  The covariance / mismatch sums are exact
// integers converted to binary32 once, every other float op is the one written in lkpyramid.cpp (no contraction),
// so results are bit-identical to the specification regardless of lane order.
#include "lvk_hip_internal.hpp"

#include <climits>

namespace {

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { p = (p < 0) ? -p : 2 * (len - 1) - p; }
    return p;
}

constexpr int LK_MARGIN = 8;     // search margin (pixels) of the staged next-frame window around the zero-flow prediction

// bytes of one pyramid level's windows and patches in the dynamic LDS block, rounded to 16
__host__ __device__ inline size_t lvk_pyrlk_part_offset(int win_w, int win_h)
{
    const size_t tarea = (size_t)(win_w + 1) * (win_h + 1), area = (size_t)win_w * win_h;
    const size_t earea = (size_t)(win_w + 3) * (win_h + 3);                  // image window + 1-px ring for the in-kernel Scharr
    const size_t jarea = (size_t)(win_w + 1 + 2 * LK_MARGIN) * (win_h + 1 + 2 * LK_MARGIN);
    return ((tarea * 4 + area * 6 + earea + jarea) + 15) & ~(size_t)15;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ __forceinline__ long long wave_sum(long long v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum of one int32 per lane over the wavefront with DPP moves (VALU only, no LDS, no barrier): row_shr 1/2/4/8 leave every
// 16-lane row's total in its last lane, row_bcast15 / row_bcast31 carry the totals across the rows into lane 63.
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Exact integer sums of K per-lane partials over the wavefront.  Integer addition is associative, so the order is free; a
// partial is split into a signed high part and an unsigned 16-bit low part whose 64-lane sums both fit in 32 bits (per-lane
// partials are < 2^46 here: <= 16 window pixels x 2^25), and each part is one DPP reduction -- about 5x cheaper than the
// LDS column sums with their three barriers that this replaces, and 10x cheaper than 64-bit ds_bpermute butterflies.
// Every lane returns all K totals in v[].
template <int K>
__device__ __forceinline__ void wave_sums(long long (&v)[K])
{
#pragma unroll
    for (int k = 0; k < K; k++)
    {
        const int lo = (int)(v[k] & 0xffff), hi = (int)(v[k] >> 16);
        v[k] = (long long)wave_sum_i32(hi) * 65536 + (long long)wave_sum_i32(lo);
    }
}

__device__ __forceinline__ void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11)
{
    w00 = (int)__builtin_rintf((1.f - a) * (1.f - b) * 16384.0f);
    w01 = (int)__builtin_rintf(a * (1.f - b) * 16384.0f);
    w10 = (int)__builtin_rintf((1.f - a) * b * 16384.0f);
    w11 = 16384 - w00 - w01 - w10;
}

// What phase A leaves behind for one pyramid level (in LDS, next to the level's windows and patches).
struct LevelState
{
    float A11, A12, A22, Dinv;     // covariance matrix of the patch gradients (scaled) and 1 / det
    int jx0, jy0;                  // origin of the staged next-frame window (INT_MIN / 2: nothing staged)
    int skip;                      // the level contributes nothing: window outside the image, or a singular system
    int pad;
};

// LDS writes of a wave followed by LDS reads of other lanes of the SAME wave: the LDS unit serves one wave's instructions in order, so
// only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The loop form is the specification's; the first two reflections are peeled so that the common cases need no loop.
__device__ __forceinline__ int reflect101_fast(int p, int len)
{
    if (len == 1) return 0;
    p = p < 0 ? -p : p;
    p = p >= len ? 2 * (len - 1) - p : p;
    while (p < 0 || p >= len) { p = (p < 0) ? -p : 2 * (len - 1) - p; }
    return p;
}

// Copies the w x h window of an image whose top-left corner is (x0, y0) (reflect-101 outside the image) into LDS, row-major with
// pitch w.  A lane's loads are issued CHUNK at a time before any of them is consumed: a window costs ceil(w h / (64 CHUNK)) global
// round trips instead of one per row group.
struct WindowSrc { const uint8_t* img; int rows, cols, step, x0, y0, w, h; };

template <int CHUNK>
__device__ __forceinline__ void window_loads(uint8_t (&v)[CHUNK], const WindowSrc& s, int base, int lane)
{
    const int total = s.w * s.h;
    const float rw = 1.0f / (float)s.w;
#pragma unroll
    for (int k = 0; k < CHUNK; k++)
    {
        const int idx = min(base + k * 64 + lane, total - 1);            // lanes past the end repeat the last pixel (not stored)
        int ty = (int)(((float)idx + 0.5f) * rw), tx = idx - ty * s.w;     // idx / w, idx % w (fixed up below: the quotient may be off by one)
        if (tx < 0) { ty--; tx += s.w; }
        if (tx >= s.w) { ty++; tx -= s.w; }
        v[k] = s.img[(long)reflect101_fast(s.y0 + ty, s.rows) * s.step + reflect101_fast(s.x0 + tx, s.cols)];
    }
}

template <int CHUNK>
__device__ __forceinline__ void window_stores(uint8_t* __restrict__ tile, const uint8_t (&v)[CHUNK], int total, int base, int lane)
{
#pragma unroll
    for (int k = 0; k < CHUNK; k++)
    {
        const int idx = base + k * 64 + lane;
        if (idx < total) tile[idx] = v[k];
    }
}

template <int CHUNK>
__device__ __forceinline__ void stage_window(uint8_t* __restrict__ tile, const WindowSrc& s, int lane)
{
    const int total = s.w * s.h;
    for (int base = 0; base < total; base += 64 * CHUNK)
    {
        uint8_t v[CHUNK];
        window_loads<CHUNK>(v, s, base, lane);
        window_stores<CHUNK>(tile, v, total, base, lane);
    }
}

// Interior windows (no reflection anywhere, and the last dword of a row stays inside the image row): rows are fetched as unaligned
// dwords with plain address arithmetic -- 4 instead of 17 loads per lane for the default windows and none of the ~90 instructions of
// reflect / divide bookkeeping per byte that made the byte path instruction-bound (5.4 us of a level's 7.4 us preparation).
struct __attribute__((packed, aligned(1))) LkU4B { uint32_t w; };

__device__ __forceinline__ bool window_is_interior(const WindowSrc& s)
{
    const int dpr = (s.w + 3) >> 2;
    return s.x0 >= 0 && s.y0 >= 0 && s.x0 + 4 * dpr <= s.cols && s.y0 + s.h <= s.rows;
}

template <int CHUNK>
__device__ __forceinline__ void interior_loads(uint32_t (&v)[CHUNK], const WindowSrc& s, int lane)
{
    const int dpr = (s.w + 3) >> 2, total = s.h * dpr;
    const float rd = 1.0f / (float)dpr;
    const uint8_t* origin = s.img + (long)s.y0 * s.step + s.x0;
#pragma unroll
    for (int k = 0; k < CHUNK; k++)
    {
        const int q = min(k * 64 + lane, total - 1);
        int ty = (int)(((float)q + 0.5f) * rd), tq = q - ty * dpr;
        if (tq < 0) { ty--; tq += dpr; }
        if (tq >= dpr) { ty++; tq -= dpr; }
        v[k] = reinterpret_cast<const LkU4B*>(origin + ty * s.step + 4 * tq)->w;
    }
}

template <int CHUNK>
__device__ __forceinline__ void interior_stores(uint8_t* __restrict__ tile, const uint32_t (&v)[CHUNK], const WindowSrc& s, int lane)
{
    const int dpr = (s.w + 3) >> 2, total = s.h * dpr;
    const float rd = 1.0f / (float)dpr;
#pragma unroll
    for (int k = 0; k < CHUNK; k++)
    {
        const int q = k * 64 + lane;
        int ty = (int)(((float)q + 0.5f) * rd), tq = q - ty * dpr;
        if (tq < 0) { ty--; tq += dpr; }
        if (tq >= dpr) { ty++; tq -= dpr; }
        if (q < total)
        {
            uint8_t* d = tile + ty * s.w + 4 * tq;
            const int nb = min(4, s.w - 4 * tq);
            d[0] = (uint8_t)v[k];
            if (nb > 1) d[1] = (uint8_t)(v[k] >> 8);
            if (nb > 2) d[2] = (uint8_t)(v[k] >> 16);
            if (nb > 3) d[3] = (uint8_t)(v[k] >> 24);
        }
    }
}

// Two windows in the same round trip(s): the loads of both are in flight before either is stored.  CA / CB = loads per lane that
// cover the default windows in one chunk; larger windows take the generic chunk loop.
template <int CA, int CB, int FA, int FB>
__device__ __forceinline__ void stage_two_windows(uint8_t* __restrict__ ta, const WindowSrc& a, uint8_t* __restrict__ tb, const WindowSrc& b, bool b_valid, int lane)
{
    const bool fa = window_is_interior(a) && a.h * ((a.w + 3) >> 2) <= 64 * FA;
    const bool fb = b_valid && window_is_interior(b) && b.h * ((b.w + 3) >> 2) <= 64 * FB;
    if (fa && fb)                                                           // wave-uniform: the common case
    {
        uint32_t va[FA], vb[FB];
        interior_loads<FA>(va, a, lane);
        interior_loads<FB>(vb, b, lane);
        interior_stores<FA>(ta, va, a, lane);
        interior_stores<FB>(tb, vb, b, lane);
        return;
    }
    const int na = a.w * a.h, nb = b_valid ? b.w * b.h : 0;
    for (int t = 0; t * 64 * CA < na || t * 64 * CB < nb; t++)
    {
        const bool do_a = t * 64 * CA < na, do_b = t * 64 * CB < nb;       // wave-uniform
        uint8_t va[CA], vb[CB];
        if (do_a) window_loads<CA>(va, a, t * 64 * CA, lane);
        if (do_b) window_loads<CB>(vb, b, t * 64 * CB, lane);
        if (do_a) window_stores<CA>(ta, va, na, t * 64 * CA, lane);
        if (do_b) window_stores<CB>(tb, vb, nb, t * 64 * CB, lane);
    }
}

template <int CB, int FB>
__device__ __forceinline__ void stage_one_window(uint8_t* __restrict__ tb, const WindowSrc& b, int lane)
{
    if (window_is_interior(b) && b.h * ((b.w + 3) >> 2) <= 64 * FB)
    {
        uint32_t vb[FB];
        interior_loads<FB>(vb, b, lane);
        interior_stores<FB>(tb, vb, b, lane);
        return;
    }
    stage_window<CB>(tb, b, lane);
}

// One block per feature, one wave per pyramid level.
//
// Phase A (all levels at once, wave w = level w): everything that depends only on the feature's previous position -- the (win + 3)^2
// window of the previous image, its Scharr derivatives, the bilinearly sampled patches I / Ix / Iy, the 2x2 covariance matrix and its
// eigenvalue test -- plus the next-frame window around the ZERO-FLOW prediction of where the search will start, with a LK_MARGIN
// search margin.  The four global round trips of the levels overlap instead of following each other.
// Phase B (wave 0 alone, coarse to fine): the Newton iterations, out of LDS; a level whose start (twice the coarser level's result)
// or whose track leaves the staged window re-centres it with one more round trip.  Same integers and float operations in the same
// order as the one-wave-per-feature kernel this replaces (38 us -> see DESIGN.md), which did the levels' staging one after the other.
//
// LDS per level: dtile[(win_h+1) * tw] short2 derivative samples, the cached patches I, Ix, Iy (int16 each, win_w * win_h), etile =
// image window with a 1-px ring, jtile = next-frame window; then one LevelState per level.
// WIN: the (square) window as a compile-time constant -- the walks over the window, its LDS pitches and the staging loops then fold
// (the tracker's window is always 11 x 11: FrameTracker.cpp:33); 0: the window of the arguments.
// LENS (fused lens mode): the feature's block also writes the lens-corrected positions of its point pair -- (previous | matched), what
// the motion is estimated from -- instead of a kernel of its own between the flow and the motion estimate.
// n_dev: the number of points when it is decided on the device (the detector's suppression grid runs inside the chain, fast.hip): the grid is
// launched for the largest possible count and workgroups beyond *n_dev leave at once
struct LensPointArgs { LensModelD model; double sx, sy; float2* und; const int* n_dev; };
template <int WIN, bool LENS>
__global__ __launch_bounds__(64 * LVK_MAX_PYR_LEVELS)
void k_pyrlk(PyrArgs prev, PyrArgs next, const float2* __restrict__ prev_pts, float2* __restrict__ prev_copy, int n,
             float2* __restrict__ next_pts, uint8_t* __restrict__ status,
             int win_w_arg, int win_h_arg, int max_count, double epsilon_sq, float min_eig_threshold, int level_bytes_arg, LensPointArgs la)
{
    LVK_TL(0);
    LVK_TRACKER_PRIORITY();
    const int win_w = WIN ? WIN : win_w_arg, win_h = WIN ? WIN : win_h_arg;
    const int level_bytes = WIN ? (int)lvk_pyrlk_part_offset(WIN, WIN) : level_bytes_arg;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int pt = blockIdx.x;                                                // grid = n
    if (la.n_dev) { n = *la.n_dev; if (pt >= n) return; }                      // (workgroup-uniform: before any barrier)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tw = win_w + 1, th = win_h + 1, area = win_w * win_h, tarea = tw * th;
    const int jw = tw + 2 * LK_MARGIN, jh = th + 2 * LK_MARGIN;               // next-frame window incl. search margin
    const int ew = tw + 2, eh = th + 2;                                       // image window with a 1-px ring (Scharr support)
    LevelState* states = reinterpret_cast<LevelState*>(smem + (size_t)prev.nlevels * level_bytes);
    // division-free walk over the window: pixel p = lane + 64 r  ->  (y, x) advances by (64 / win_w, 64 % win_w)
    const int py0 = lane / win_w, px0 = lane - py0 * win_w, pdy_ = 64 / win_w, pdx_ = 64 - pdy_ * win_w;
    const int lx = lane & 15, ly = lane >> 4;                                 // 16 x 4 lane grid for the staging loops

    const float2 p0 = prev_pts[pt];                        // may live in pinned host memory: one 8-byte read per wave
    if (threadIdx.x == 0 && prev_copy) prev_copy[pt] = p0; // device-resident copy for the kernels that follow
    const float halfx = (win_w - 1) * 0.5f, halfy = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int top = prev.nlevels - 1;

#ifdef LVK_LK_TIMING
    long long lt[8]; int ln = 0; int iters_total = 0;
    lt[ln++] = wall_clock64();
#endif
    // ---- phase A: this wave's level
    {
        const int level = wave;
        uint8_t* base = smem + (size_t)level * level_bytes;
        short2* dtile = reinterpret_cast<short2*>(base);                          // tarea * 4 B
        short* Iw = reinterpret_cast<short*>(base + (size_t)tarea * 4);           // area * 2 B
        short* Ixw = Iw + area;
        short* Iyw = Ixw + area;
        uint8_t* etile = reinterpret_cast<uint8_t*>(Iyw + area);                  // ew * eh B
        uint8_t* jtile = etile + ew * eh;                                         // jw * jh B
        const PyrLevel I = prev.lv[level];
        const PyrLevel J = next.lv[level];
        float px = p0.x * (float)(1. / (1 << level));
        float py = p0.y * (float)(1. / (1 << level));
        const float nx = px, ny = py;                                             // zero-flow prediction of the level's starting point
        LevelState st;
        st.A11 = st.A12 = st.A22 = st.Dinv = 0.f; st.jx0 = st.jy0 = INT_MIN / 2; st.skip = 0; st.pad = 0;
        px -= halfx; py -= halfy;
        const int ipx = (int)__builtin_floorf(px), ipy = (int)__builtin_floorf(py);
        if (ipx < -win_w || ipx >= I.cols || ipy < -win_h || ipy >= I.rows) st.skip = 1;
        else
        {
            const float a = px - ipx, b = py - ipy;
            int w00, w01, w10, w11;
            bilinear_weights(a, b, w00, w01, w10, w11);
            // stage the window of the previous image (reflect-101 border) and -- in the same round trip -- the next-frame window
            int jx0 = (int)__builtin_floorf(nx - halfx), jy0 = (int)__builtin_floorf(ny - halfy);
            // only positions that pass the tracker's own bounds test are staged (anything else never samples the image)
            const bool j_valid = !(jx0 < -win_w || jx0 >= J.cols || jy0 < -win_h || jy0 >= J.rows);
            jx0 = j_valid ? jx0 - LK_MARGIN : INT_MIN / 2; jy0 = j_valid ? jy0 - LK_MARGIN : INT_MIN / 2;
            // (defaults: 14 x 14 and 28 x 28 bytes = 1 + 4 dword loads per lane for interior windows, 4 + 13 byte loads at the border; one round trip)
            stage_two_windows<4, 13, 1, 4>(etile, WindowSrc{I.img, I.rows, I.cols, I.step, ipx - 1, ipy - 1, ew, eh},
                                     jtile, WindowSrc{J.img, J.rows, J.cols, J.step, jx0, jy0, jw, jh}, j_valid, lane);
            wave_lds_sync();
            // calcScharrDeriv on the staged window (reflect-101 ring, same integers as k_scharr_all); positions outside the image get
            // zero derivatives like the zero border the derivative images used to be read with
            for (int ty = ly; ty < th; ty += 4)
                for (int tx = lx; tx < tw; tx += 16)
                {
                    const int xx = ipx + tx, yy = ipy + ty;
                    short2 d = make_short2(0, 0);
                    if (xx >= 0 && yy >= 0 && xx < I.cols && yy < I.rows)
                    {
                        const uint8_t* r0 = etile + ty * ew + tx; const uint8_t* r1 = r0 + ew; const uint8_t* r2 = r1 + ew;
                        const int t0m = (r0[0] + r2[0]) * 3 + r1[0] * 10, t0p = (r0[2] + r2[2]) * 3 + r1[2] * 10;
                        const int t1m = r2[0] - r0[0], t1c = r2[1] - r0[1], t1p = r2[2] - r0[2];
                        d = make_short2((short)(t0p - t0m), (short)((t1p + t1m) * 3 + t1c * 10));
                    }
                    dtile[ty * tw + tx] = d;
                }
            wave_lds_sync();
            const uint8_t* tile = etile + ew + 1;                                     // the (win+1)^2 image window, pitch ew
            long long sA[3] = {0, 0, 0};
            for (int p = lane, y = py0, x = px0; p < area; p += 64)
            {
                const int i00 = y * tw + x, i01 = i00 + 1, i10 = i00 + tw, i11 = i10 + 1;
                const int e00 = y * ew + x, e01 = e00 + 1, e10 = e00 + ew, e11 = e10 + 1;
                // every factor fits 24 signed bits (samples <= 255, |derivatives| <= 4080, weights <= 16384): v_mul_i32_i24 / v_mad_i32_i24
                // instead of the multi-pass 32-bit v_mul_lo_u32; the products are exact either way
                const int ival = descale(__mul24(tile[e00], w00) + __mul24(tile[e01], w01) + __mul24(tile[e10], w10) + __mul24(tile[e11], w11), 14 - 5);
                const int ixval = descale(__mul24(dtile[i00].x, w00) + __mul24(dtile[i01].x, w01) + __mul24(dtile[i10].x, w10) + __mul24(dtile[i11].x, w11), 14);
                const int iyval = descale(__mul24(dtile[i00].y, w00) + __mul24(dtile[i01].y, w01) + __mul24(dtile[i10].y, w10) + __mul24(dtile[i11].y, w11), 14);
                Iw[p] = (short)ival; Ixw[p] = (short)ixval; Iyw[p] = (short)iyval;
                sA[0] += (long long)__mul24(ixval, ixval);           // |ixval|, |iyval| <= 4080: the squares fit 32 bits
                sA[1] += (long long)__mul24(ixval, iyval);
                sA[2] += (long long)__mul24(iyval, iyval);
                y += pdy_; x += pdx_; if (x >= win_w) { x -= win_w; y++; }
            }
            wave_sums<3>(sA);
            const float A11 = (float)(double)sA[0] * FLT_SCALE, A12 = (float)(double)sA[1] * FLT_SCALE, A22 = (float)(double)sA[2] * FLT_SCALE;
            const float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - __builtin_sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win_w * win_h);
            if (minEig < min_eig_threshold || D < 1.1920928955078125e-07f) st.skip = 1;
            else { st.A11 = A11; st.A12 = A12; st.A22 = A22; st.Dinv = 1.f / D; st.jx0 = jx0; st.jy0 = jy0; }
        }
        if (lane == 0) states[level] = st;
    }
#ifdef LVK_LK_TIMING
    lt[ln++] = wall_clock64();
#endif
    __syncthreads();
    if (wave != 0) return;
#ifdef LVK_LK_TIMING
    lt[ln++] = wall_clock64();
#endif

    // ---- phase B: coarse to fine
    float outx = 0.f, outy = 0.f;
    bool ok = true;                                                           // status (initialised to 1 by calc())
    for (int level = top; level >= 0; level--)
    {
        const PyrLevel J = next.lv[level];
        const LevelState st = states[level];
        uint8_t* base = smem + (size_t)level * level_bytes;
        const short* Iw = reinterpret_cast<const short*>(base + (size_t)tarea * 4);
        const short* Ixw = Iw + area;
        const short* Iyw = Ixw + area;
        uint8_t* jtile = base + (size_t)tarea * 4 + (size_t)area * 6 + ew * eh;
        float nx, ny;
        if (level == top) { nx = p0.x * (float)(1. / (1 << level)); ny = p0.y * (float)(1. / (1 << level)); }
        else { nx = outx * 2.f; ny = outy * 2.f; }
        outx = nx; outy = ny;
        if (st.skip)
        {
            if (level == 0) ok = false;
            continue;
        }
        const float A11 = st.A11, A12 = st.A12, A22 = st.A22, D = st.Dinv;
        int jx0 = st.jx0, jy0 = st.jy0;
        nx -= halfx; ny -= halfy;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < max_count; j++)
        {
            const int inx = (int)__builtin_floorf(nx), iny = (int)__builtin_floorf(ny);
            if (inx < -win_w || inx >= J.cols || iny < -win_h || iny >= J.rows)
            {
                if (level == 0) ok = false;
                break;
            }
            const float a = nx - inx, b = ny - iny;
            int w00, w01, w10, w11;
            bilinear_weights(a, b, w00, w01, w10, w11);
            if (inx < jx0 || iny < jy0 || inx + tw > jx0 + jw || iny + th > jy0 + jh)
            {
                // the start or the track is outside the staged window: re-centre it on the current position
                jx0 = inx - LK_MARGIN; jy0 = iny - LK_MARGIN;
                wave_lds_sync();
                stage_one_window<13, 4>(jtile, WindowSrc{J.img, J.rows, J.cols, J.step, jx0, jy0, jw, jh}, lane);
                wave_lds_sync();
            }
            const uint8_t* jt = jtile + (iny - jy0) * jw + (inx - jx0);
            long long sb[2] = {0, 0};
            for (int p = lane, y = py0, x = px0; p < area; p += 64)
            {
                const int i00 = y * jw + x, i01 = i00 + 1, i10 = i00 + jw, i11 = i10 + 1;
                const int diff = descale(__mul24(jt[i00], w00) + __mul24(jt[i01], w01) + __mul24(jt[i10], w10) + __mul24(jt[i11], w11), 14 - 5) - Iw[p];
                sb[0] += (long long)__mul24(diff, Ixw[p]);              // |diff| <= 8160, |Ix| <= 4080
                sb[1] += (long long)__mul24(diff, Iyw[p]);
                y += pdy_; x += pdx_; if (x >= win_w) { x -= win_w; y++; }
            }
            wave_sums<2>(sb);
            const float b1 = (float)(double)sb[0] * FLT_SCALE, b2 = (float)(double)sb[1] * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            outx = nx + halfx; outy = ny + halfy;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= epsilon_sq) break;
            if (j > 0 && (double)__builtin_fabsf(dx + pdx) < 0.01 && (double)__builtin_fabsf(dy + pdy) < 0.01)
            {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
#ifdef LVK_LK_TIMING
            iters_total++;
#endif
        }
#ifdef LVK_LK_TIMING
        if (ln < 8) lt[ln++] = wall_clock64();
#endif
    }
#ifdef LVK_LK_TIMING
    if (lane == 0 && (pt == 0 || pt == 300) && n > 400)
    {
        printf("pyrlk pt %d (100 MHz ticks): phase A %lld, barrier %lld, levels", pt, lt[1] - lt[0], lt[2] - lt[1]);
        for (int k = 3; k < ln; k++) printf(" %lld", lt[k] - lt[k - 1]);
        printf(" | full iterations %d\n", iters_total);
    }
#endif
    if (lane == 0)
    {
        next_pts[pt] = make_float2(outx, outy);
        status[pt] = ok ? 1 : 0;
    }
    if (LENS && lane < 2)
    {
        // lane 0: the previous position, lane 1: the matched one (the same function k_lens_undistort applies, point by point)
        const float2 q = lvk_lens_undistort_point(la.model, la.sx, la.sy, lane == 0 ? p0 : make_float2(outx, outy));
        la.und[lane == 0 ? pt : n + pt] = q;
    }
}

} // namespace

// dynamic LDS of k_pyrlk: the windows and patches of every level, then the per-level state
size_t lvk_pyrlk_lds_bytes(int win_w, int win_h, int nlevels)
{
    return (size_t)nlevels * lvk_pyrlk_part_offset(win_w, win_h) + (size_t)nlevels * sizeof(LevelState);
}

int lvk_launch_pyrlk(lvk_hip_ctx* ctx, const PyrArgs& prev, const PyrArgs& next, const float2* d_prev_pts, int n,
                     float2* d_next_pts, uint8_t* d_status, int win_w, int win_h, int max_count, double epsilon, double min_eig, float2* d_prev_copy,
                     const LensModel* lens, double lens_sx, double lens_sy, float2* d_und, const int* d_n)
{
    LVK_HIP_REQUIRE(ctx, prev.nlevels >= 1 && prev.nlevels == next.nlevels && prev.nlevels <= LVK_MAX_PYR_LEVELS);
    LVK_HIP_REQUIRE(ctx, win_w >= 3 && win_h >= 3 && win_w <= 31 && win_h <= 31);
    if (n <= 0) return LVK_HIP_OK;
    // SparsePyrLKOpticalFlowImpl: criteria clamp, epsilon squared
    max_count = std::min(std::max(max_count, 0), 100);
    epsilon = std::min(std::max(epsilon, 0.), 10.);
    epsilon *= epsilon;
    const size_t lds = lvk_pyrlk_lds_bytes(win_w, win_h, prev.nlevels);
    LensPointArgs la{};
    const bool with_lens = lens != nullptr && d_und != nullptr;
    if (with_lens) { for (int i = 0; i < 17; i++) la.model.d[i] = lens->d[i]; la.sx = lens_sx; la.sy = lens_sy; la.und = d_und; }
    la.n_dev = d_n;
    auto launch = [&](auto kernel) -> int {
        if (lds > 48 * 1024)
            LVK_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3(n), dim3(64 * prev.nlevels), lds, ctx->stream, prev, next, d_prev_pts, d_prev_copy, n,
                           d_next_pts, d_status, win_w, win_h, max_count, epsilon, (float)min_eig, (int)lvk_pyrlk_part_offset(win_w, win_h), la);
        return LVK_HIP_OK;
    };
    int lrc;
    if (win_w == 11 && win_h == 11) lrc = with_lens ? launch(k_pyrlk<11, true>) : launch(k_pyrlk<11, false>);
    else lrc = with_lens ? launch(k_pyrlk<0, true>) : launch(k_pyrlk<0, false>);
    if (lrc != LVK_HIP_OK) return lrc;
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

// Level geometry of buildOpticalFlowPyramid: halve until a level would be no larger than the window.
int lvk_pyramid_geometry(int rows, int cols, int max_level, int win_w, int win_h, int* lrows, int* lcols)
{
    int n = 0;
    lrows[n] = rows; lcols[n] = cols; n++;
    for (int level = 1; level <= max_level && n < LVK_MAX_PYR_LEVELS; level++)
    {
        const int r = (lrows[n - 1] + 1) / 2, c = (lcols[n - 1] + 1) / 2;
        if (c <= win_w || r <= win_h) break;
        lrows[n] = r; lcols[n] = c; n++;
    }
    return n;
}

int DevicePyramid::allocate(lvk_hip_ctx* ctx, int rows, int cols, int max_level, int win_w, int win_h)
{
    release();
    int lr[LVK_MAX_PYR_LEVELS], lc[LVK_MAX_PYR_LEVELS];
    const int n = lvk_pyramid_geometry(rows, cols, max_level, win_w, win_h, lr, lc);
    size_t img_bytes = 0, der_bytes = 0;
    for (int i = 0; i < n; i++) { img_bytes += (size_t)lr[i] * ((lc[i] + 63) & ~63); der_bytes += (size_t)lr[i] * lc[i] * 4; }
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&img_base, img_bytes));
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&deriv_base, der_bytes));
    uint8_t* ip = img_base; uint8_t* dp = deriv_base;
    args.nlevels = n;
    for (int i = 0; i < n; i++)
    {
        const int step = (lc[i] + 63) & ~63;
        args.lv[i].img = ip; args.lv[i].deriv = reinterpret_cast<const short2*>(dp);
        args.lv[i].rows = lr[i]; args.lv[i].cols = lc[i]; args.lv[i].step = step;
        ip += (size_t)lr[i] * step; dp += (size_t)lr[i] * lc[i] * 4;
    }
    return LVK_HIP_OK;
}

void DevicePyramid::release()
{
    if (img_base) (void)hipFree(img_base);
    if (deriv_base) (void)hipFree(deriv_base);
    img_base = deriv_base = nullptr;
    args.nlevels = 0;
}

// Level 0 must already hold the tracking-resolution image; builds levels 1.. and all derivative images.
int DevicePyramid::build(lvk_hip_ctx* ctx, bool derivs) { return lvk_launch_pyramid(ctx, args, derivs); }

extern "C" {

// Synchronous test entry point mirroring calc(prevImg, nextImg, prevPts, nextPts, status): device images of the
// tracking resolution, host point arrays.
int lvk_hip_pyrlk(lvk_hip_ctx* ctx, const void* d_prev, int prev_step, const void* d_next, int next_step, int rows, int cols,
                  const float* prev_pts, int n, float* next_pts, uint8_t* status,
                  int win_w, int win_h, int max_level, int max_count, double epsilon, double min_eig_threshold)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, d_prev && d_next && rows > 0 && cols > 0 && n >= 0 && (n == 0 || (prev_pts && next_pts && status)));
    DevicePyramid P, N;
    int rc;
    if ((rc = P.allocate(ctx, rows, cols, max_level, win_w, win_h)) != LVK_HIP_OK) return rc;
    if ((rc = N.allocate(ctx, rows, cols, max_level, win_w, win_h)) != LVK_HIP_OK) { P.release(); return rc; }
    float2 *d_p = nullptr, *d_n = nullptr; uint8_t* d_s = nullptr;
    auto cleanup = [&]() { P.release(); N.release(); if (d_p) (void)hipFree(d_p); if (d_n) (void)hipFree(d_n); if (d_s) (void)hipFree(d_s); };
    hipError_t e;
    if ((e = hipMemcpy2DAsync(const_cast<uint8_t*>(P.args.lv[0].img), P.args.lv[0].step, d_prev, prev_step, cols, rows, hipMemcpyDeviceToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpy2DAsync(const_cast<uint8_t*>(N.args.lv[0].img), N.args.lv[0].step, d_next, next_step, cols, rows, hipMemcpyDeviceToDevice, ctx->stream)) != hipSuccess)
    { cleanup(); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
    if ((rc = P.build(ctx)) != LVK_HIP_OK || (rc = N.build(ctx)) != LVK_HIP_OK) { cleanup(); return rc; }
    if (n > 0)
    {
        if ((e = hipMalloc((void**)&d_p, n * sizeof(float2))) != hipSuccess || (e = hipMalloc((void**)&d_n, n * sizeof(float2))) != hipSuccess ||
            (e = hipMalloc((void**)&d_s, n)) != hipSuccess ||
            (e = hipMemcpyAsync(d_p, prev_pts, n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        { cleanup(); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
        if ((rc = lvk_launch_pyrlk(ctx, P.args, N.args, d_p, n, d_n, d_s, win_w, win_h, max_count, epsilon, min_eig_threshold)) != LVK_HIP_OK) { cleanup(); return rc; }
        if ((e = hipMemcpyAsync(next_pts, d_n, n * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
            (e = hipMemcpyAsync(status, d_s, n, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        { cleanup(); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
    }
    e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
    return P.args.nlevels >= 0 ? LVK_HIP_OK : LVK_HIP_ERR_RUNTIME;
}


// Synchronous test entry point: builds the optical-flow pyramid (levels + Scharr derivative images) of a device image and
// returns it to the host, tightly packed level after level.  Returns the level count (or a negative status).
int lvk_hip_build_pyramid(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols, int max_level, int win_w, int win_h,
                          uint8_t* levels, int16_t* derivs, int* level_rows, int* level_cols)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, d_img && levels && derivs && level_rows && level_cols && rows > 0 && cols > 0);
    DevicePyramid P;
    int rc;
    if ((rc = P.allocate(ctx, rows, cols, max_level, win_w, win_h)) != LVK_HIP_OK) return rc;
    hipError_t e = hipMemcpy2DAsync(const_cast<uint8_t*>(P.args.lv[0].img), P.args.lv[0].step, d_img, step, cols, rows, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess && (rc = P.build(ctx, true)) == LVK_HIP_OK)
    {
        size_t lo = 0, dofs = 0;
        for (int i = 0; i < P.args.nlevels && e == hipSuccess; i++)
        {
            const PyrLevel& L = P.args.lv[i];
            level_rows[i] = L.rows; level_cols[i] = L.cols;
            e = hipMemcpy2DAsync(levels + lo, L.cols, L.img, L.step, L.cols, L.rows, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(derivs + dofs, L.deriv, (size_t)L.rows * L.cols * 4, hipMemcpyDeviceToHost, ctx->stream);
            lo += (size_t)L.rows * L.cols; dofs += (size_t)L.rows * L.cols * 2;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    const int n = P.args.nlevels;
    P.release();
    if (e != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
    return rc == LVK_HIP_OK ? n : rc;
}

} // extern "C"

LVK_TL_EXPORT(pyrlk)
