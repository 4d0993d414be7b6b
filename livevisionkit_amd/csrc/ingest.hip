// YUV420 (I420 / NV12) <-> packed YUV444 8UC3 conversion for gfx950: the step either side of the stabilization
// filter in the OBS asynchronous path.
//
// Replaces I4XXIngest::to_ocl / to_obs and NV12Ingest::to_ocl / to_obs (reference:
// Modules/OBS-Plugin/Interop/FrameIngest.cpp:494-557,567-602): chroma upsampling with cv::resize(INTER_LINEAR) + cv::merge
// on the way in, cv::split + cv::resize(0.5, 0.5, INTER_AREA) on the way out.  Arithmetic = OpenCV 4.8.0 8-bit CPU paths
// (fixed-point bilinear with 11-bit coefficients and the ((b * (S >> 4)) >> 16) vertical pass; 2x2 area = (s + 2) >> 2).
// Integer only -> bit-exact.  Both kernels are pure streaming kernels: every byte is read and written once.
#include "lvk_hip_internal.hpp"

#include <cmath>
#include <algorithm>
#include <string>

// streaming stores: the converted frame is read N pushes later, see remap.hip
#define LVK_STREAM_STORE(ptr, v) __builtin_nontemporal_store((uint32_t)(v), (ptr))

namespace {

__device__ __forceinline__ uint32_t lin8(const uint8_t* __restrict__ r0, const uint8_t* __restrict__ r1, int pix,
                                         const Lin8Entry tx, int b0, int b1)
{
    const int h0 = r0[tx.s0 * pix] * tx.a0 + r0[tx.s1 * pix] * tx.a1;
    const int h1 = r1[tx.s0 * pix] * tx.a0 + r1[tx.s1 * pix] * tx.a1;
    return (uint32_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xffu;
}

// thread = 4 horizontally adjacent output pixels (12 packed bytes = 3 dwords)
template <bool NV12>
__global__ __launch_bounds__(256)
void k_ingest_yuv420(const uint8_t* __restrict__ yp, int y_step, const uint8_t* __restrict__ up, int u_step,
                     const uint8_t* __restrict__ vp, int v_step, int rows, int cols,
                     uint8_t* __restrict__ dst, int dst_step, const Lin8Entry* __restrict__ xtab, const Lin8Entry* __restrict__ ytab, int fast)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= cols || y >= rows) return;
    const Lin8Entry ty = ytab[y];
    const uint8_t* u0 = up + (long)ty.s0 * u_step; const uint8_t* u1 = up + (long)ty.s1 * u_step;
    const uint8_t* v0 = NV12 ? u0 + 1 : vp + (long)ty.s0 * v_step;
    const uint8_t* v1 = NV12 ? u1 + 1 : vp + (long)ty.s1 * v_step;
    constexpr int PIX = NV12 ? 2 : 1;
    const int npx = min(4, cols - x0);
    uint32_t px[4];
#pragma unroll
    for (int p = 0; p < 4; p++)
    {
        px[p] = 0;
        if (p < npx)
        {
            const Lin8Entry tx = xtab[x0 + p];
            px[p] = (uint32_t)yp[(long)y * y_step + x0 + p] | (lin8(u0, u1, PIX, tx, ty.a0, ty.a1) << 8) | (lin8(v0, v1, PIX, tx, ty.a0, ty.a1) << 16);
        }
    }
    uint8_t* drow = dst + (long)y * dst_step;
    if (fast && npx == 4)
    {
        uint32_t* d = reinterpret_cast<uint32_t*>(drow + 3 * x0);
        d[0] = px[0] | (px[1] << 24);
        d[1] = (px[1] >> 8) | (px[2] << 16);
        d[2] = (px[2] >> 16) | (px[3] << 8);
    }
    else
        for (int p = 0; p < npx; p++) { uint8_t* d = drow + 3 * (x0 + p); d[0] = (uint8_t)px[p]; d[1] = (uint8_t)(px[p] >> 8); d[2] = (uint8_t)(px[p] >> 16); }
}

// Exact 2x chroma upsampling (always the case for 4:2:0) without tables, byte loads, multiplications or left shifts.  For the 2x
// case the fixed-point INTER_LINEAR of the general kernel collapses: the horizontal pass (c_a a0 + c_b a1) >> 4 with (a0, a1) =
// (512, 1536) / (1536, 512) / (2048, 0 at the frame edge) is exactly 32 t with t = c_a + 3 c_b / 3 c_a + c_b / 4 c_a, the edge case
// being the general one with the edge sample replicated; and the vertical pass ((1536 h0 >> 16) + (512 h1 >> 16) + 2) >> 2 is
// ((3 t0 >> 2) + (t1 >> 2) + 2) >> 2.  Only additions, right shifts and ANDs remain -- the opcodes gfx950 issues at full rate
// (scripts/valu_peak.hip); the multiply / 64-bit-shift form this replaces was VALU-bound at 11.4 us for a 4K frame.
// A thread produces the 4 x 2 output pixels of the luma rows 2k - 1 and 2k: both interpolate between the SAME two chroma rows
// (k - 1, k) with mirrored weights.  Its chroma columns c0 - 1 .. c0 + 2 (c0 = x0 / 2) are one unaligned dword per plane and row
// (NV12: one 8-byte load, de-interleaved with v_perm_b32); the first / last thread of a row replicates the edge sample.
// Preconditions (checked by the launcher): Y and dst dword aligned incl. pitch, cols % 4 == 0, cols >= 16.
template <bool NV12>
__global__ __launch_bounds__(256)
void k_ingest_yuv420_x2(const uint8_t* __restrict__ yp, int y_step, const uint8_t* __restrict__ up, int u_step,
                        const uint8_t* __restrict__ vp, int v_step, int rows, int cols, uint8_t* __restrict__ dst, int dst_step)
{
    LVK_TL(0);
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int k = blockIdx.y * 4 + threadIdx.y;                   // luma rows 2k - 1 (odd) and 2k (even); k = 0 .. rows / 2
    const int cc = cols >> 1, cr = rows >> 1;
    if (x0 >= cols || k > cr) return;
    // vertical taps (rows clipped individually, coefficients unclamped -- resize.cpp resizeGeneric_Invoker)
    const int r0 = max(k - 1, 0), r1 = min(k, cr - 1);
    const int c0 = x0 >> 1;
    const bool left = x0 == 0, right = x0 == cols - 4;
    const int lc = left ? 0 : (right ? cc - 4 : c0 - 1);          // first chroma column of the dword that is loaded
    struct __attribute__((packed, aligned(1))) P4 { uint32_t w; };
    struct __attribute__((packed, aligned(1))) P8 { uint32_t w[2]; };
    uint32_t su0, su1, sv0, sv1;                                   // samples c0 - 1 .. c0 + 2 of (U, V) x (row r0, row r1), one per byte
    if (NV12)
    {
        const P8 a = *reinterpret_cast<const P8*>(up + (long)r0 * u_step + 2 * lc);
        const P8 b = *reinterpret_cast<const P8*>(up + (long)r1 * u_step + 2 * lc);
        su0 = __builtin_amdgcn_perm(a.w[1], a.w[0], 0x06040200u); sv0 = __builtin_amdgcn_perm(a.w[1], a.w[0], 0x07050301u);
        su1 = __builtin_amdgcn_perm(b.w[1], b.w[0], 0x06040200u); sv1 = __builtin_amdgcn_perm(b.w[1], b.w[0], 0x07050301u);
    }
    else
    {
        su0 = reinterpret_cast<const P4*>(up + (long)r0 * u_step + lc)->w;
        su1 = reinterpret_cast<const P4*>(up + (long)r1 * u_step + lc)->w;
        sv0 = reinterpret_cast<const P4*>(vp + (long)r0 * v_step + lc)->w;
        sv1 = reinterpret_cast<const P4*>(vp + (long)r1 * v_step + lc)->w;
    }
    if (left)  { su0 = (su0 << 8) | (su0 & 0xffu); su1 = (su1 << 8) | (su1 & 0xffu); sv0 = (sv0 << 8) | (sv0 & 0xffu); sv1 = (sv1 << 8) | (sv1 & 0xffu); }
    if (right) { su0 = (su0 >> 8) | (su0 & 0xff000000u); su1 = (su1 >> 8) | (su1 & 0xff000000u);
                 sv0 = (sv0 >> 8) | (sv0 & 0xff000000u); sv1 = (sv1 >> 8) | (sv1 & 0xff000000u); }
    // horizontal pass: t[p] for the 4 output columns of one window
    auto horizontal = [](uint32_t w, uint32_t (&t)[4]) {
        const uint32_t s0 = w & 0xffu, s1 = (w >> 8) & 0xffu, s2 = (w >> 16) & 0xffu, s3 = w >> 24;
        const uint32_t m1 = s1 + s1 + s1, m2 = s2 + s2 + s2;
        t[0] = s0 + m1; t[1] = m1 + s2; t[2] = s1 + m2; t[3] = m2 + s3;
    };
    uint32_t tu0[4], tu1[4], tv0[4], tv1[4];
    horizontal(su0, tu0); horizontal(su1, tu1); horizontal(sv0, tv0); horizontal(sv1, tv1);
    const int ya = 2 * k - 1, yb = 2 * k;
    const bool has_a = ya >= 0, has_b = yb < rows;
    const uint32_t ywa = has_a ? *reinterpret_cast<const uint32_t*>(yp + (long)ya * y_step + x0) : 0u;
    const uint32_t ywb = has_b ? *reinterpret_cast<const uint32_t*>(yp + (long)yb * y_step + x0) : 0u;
    uint32_t pa[4], pb[4];
#pragma unroll
    for (int p = 0; p < 4; p++)
    {
        // odd row 2k - 1: weights (3/4, 1/4) on chroma rows (k - 1, k); even row 2k: (1/4, 3/4)
        const uint32_t u0 = tu0[p], u1 = tu1[p], v0 = tv0[p], v1 = tv1[p];
        const uint32_t ua = ((((u0 + u0 + u0) >> 2) + (u1 >> 2) + 2u) >> 2), ub = (((u0 >> 2) + ((u1 + u1 + u1) >> 2) + 2u) >> 2);
        const uint32_t va = ((((v0 + v0 + v0) >> 2) + (v1 >> 2) + 2u) >> 2), vb = (((v0 >> 2) + ((v1 + v1 + v1) >> 2) + 2u) >> 2);
        pa[p] = ((ywa >> (8 * p)) & 0xffu) | (ua << 8) | (va << 16);
        pb[p] = ((ywb >> (8 * p)) & 0xffu) | (ub << 8) | (vb << 16);
    }
    if (has_a)
    {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + (long)ya * dst_step + 3 * x0);
        LVK_STREAM_STORE(d + 0, pa[0] | (pa[1] << 24)); LVK_STREAM_STORE(d + 1, (pa[1] >> 8) | (pa[2] << 16)); LVK_STREAM_STORE(d + 2, (pa[2] >> 16) | (pa[3] << 8));
    }
    if (has_b)
    {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + (long)yb * dst_step + 3 * x0);
        LVK_STREAM_STORE(d + 0, pb[0] | (pb[1] << 24)); LVK_STREAM_STORE(d + 1, (pb[1] >> 8) | (pb[2] << 16)); LVK_STREAM_STORE(d + 2, (pb[2] >> 16) | (pb[3] << 8));
    }
}

// thread = one 2x2 block of packed pixels -> 4 luma bytes + one (U, V) sample
template <bool NV12>
__global__ __launch_bounds__(256)
void k_egress_yuv420(const uint8_t* __restrict__ src, int src_step, int rows, int cols,
                     uint8_t* __restrict__ yp, int y_step, uint8_t* __restrict__ up, int u_step, uint8_t* __restrict__ vp, int v_step)
{
    const int cx = blockIdx.x * 64 + threadIdx.x;
    const int cy = blockIdx.y * 4 + threadIdx.y;
    if (cx >= cols / 2 || cy >= rows / 2) return;
    const uint8_t* p0 = src + (long)(2 * cy) * src_step + 6 * (long)cx;
    const uint8_t* p1 = p0 + src_step;
    uint8_t a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { a[i] = p0[i]; b[i] = p1[i]; }
    uint8_t* y0 = yp + (long)(2 * cy) * y_step + 2 * cx;
    y0[0] = a[0]; y0[1] = a[3]; y0[y_step] = b[0]; y0[y_step + 1] = b[3];
    const uint8_t u = (uint8_t)((a[1] + a[4] + b[1] + b[4] + 2) >> 2);
    const uint8_t v = (uint8_t)((a[2] + a[5] + b[2] + b[5] + 2) >> 2);
    if (NV12) { uint8_t* d = up + (long)cy * u_step + 2 * cx; d[0] = u; d[1] = v; }
    else { up[(long)cy * u_step + cx] = u; vp[(long)cy * v_step + cx] = v; }
}


// ---- the other OBS video formats of FrameIngest::Select (round 6; reference Modules/OBS-Plugin/Interop/FrameIngest.cpp:476-753) ----------------
// 4:2:2 (planar I422 / I42A, packed YUY2 / YVYU / UYVY) -> packed 444: cv::resize(chroma, frame size, INTER_LINEAR) where only the width doubles.
// The two-pass fixed point collapses for the exact 2x: horizontal (c_a 512 + c_b 1536) >> 4 = 32 (c_a + 3 c_b), vertical pass with (2048, 0):
// (2048 * 32 t) >> 16 = t, result (t + 2) >> 2; at the frame edge the single sample (4 c + 2) >> 2 = c, which is the same formula with the edge
// column replicated.  LAYOUT: 0 planar, 1 YUY2 (Y U Y V), 2 YVYU (Y V Y U), 3 UYVY (U Y V Y).  A thread = 4 output pixels.
template <int LAYOUT>
__global__ __launch_bounds__(256)
void k_ingest_422(const uint8_t* __restrict__ p0, int s0, const uint8_t* __restrict__ p1, int s1, const uint8_t* __restrict__ p2, int s2,
                  int rows, int cols, uint8_t* __restrict__ dst, int dst_step, int fast)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= cols || y >= rows) return;
    const int cc = cols >> 1;
    constexpr int YOFF = LAYOUT == 3 ? 1 : 0, COFF = 1 - YOFF;                 // P422Ingest::m_YFirst
    constexpr bool UFIRST = LAYOUT != 2;                                        // P422Ingest::m_UFirst
    const uint8_t* r0 = p0 + (long)y * s0;
    const uint8_t* r1 = LAYOUT == 0 ? p1 + (long)y * s1 : r0;
    const uint8_t* r2 = LAYOUT == 0 ? p2 + (long)y * s2 : r0;
    auto U = [&](int k) -> uint32_t { k = min(max(k, 0), cc - 1); return LAYOUT == 0 ? r1[k] : r0[4 * k + COFF + (UFIRST ? 0 : 2)]; };
    auto V = [&](int k) -> uint32_t { k = min(max(k, 0), cc - 1); return LAYOUT == 0 ? r2[k] : r0[4 * k + COFF + (UFIRST ? 2 : 0)]; };
    const int npx = min(4, cols - x0);
    uint32_t px[4];
#pragma unroll
    for (int p = 0; p < 4; p++)
    {
        px[p] = 0;
        if (p < npx)
        {
            const int x = x0 + p, k = x >> 1, o = (x & 1) ? k + 1 : k - 1;      // the nearer chroma column k (weight 3/4) and the other one (1/4)
            const uint32_t yy = LAYOUT == 0 ? r0[x] : r0[2 * x + YOFF];
            const uint32_t u = (U(o) + 3u * U(k) + 2u) >> 2, v = (V(o) + 3u * V(k) + 2u) >> 2;
            px[p] = yy | (u << 8) | (v << 16);
        }
    }
    uint8_t* drow = dst + (long)y * dst_step;
    if (fast && npx == 4)
    {
        uint32_t* d = reinterpret_cast<uint32_t*>(drow + 3 * x0);
        LVK_STREAM_STORE(d + 0, px[0] | (px[1] << 24)); LVK_STREAM_STORE(d + 1, (px[1] >> 8) | (px[2] << 16)); LVK_STREAM_STORE(d + 2, (px[2] >> 16) | (px[3] << 8));
    }
    else
        for (int p = 0; p < npx; p++) { uint8_t* d = drow + 3 * (x0 + p); d[0] = (uint8_t)px[p]; d[1] = (uint8_t)(px[p] >> 8); d[2] = (uint8_t)(px[p] >> 16); }
}

// 4:4:4 -> packed 444: cv::merge of the three planes (I444 / YUVA, FrameIngest.cpp:521) or the last three bytes of A Y U V (AYUV, :686).  LAYOUT 0 / 1.
template <int LAYOUT>
__global__ __launch_bounds__(256)
void k_ingest_444(const uint8_t* __restrict__ p0, int s0, const uint8_t* __restrict__ p1, int s1, const uint8_t* __restrict__ p2, int s2,
                  int rows, int cols, uint8_t* __restrict__ dst, int dst_step, int fast)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= cols || y >= rows) return;
    const int npx = min(4, cols - x0);
    uint32_t px[4];
#pragma unroll
    for (int p = 0; p < 4; p++)
    {
        px[p] = 0;
        if (p < npx)
        {
            const int x = x0 + p;
            if (LAYOUT == 0) px[p] = (uint32_t)p0[(long)y * s0 + x] | ((uint32_t)p1[(long)y * s1 + x] << 8) | ((uint32_t)p2[(long)y * s2 + x] << 16);
            else { const uint8_t* s = p0 + (long)y * s0 + 4 * x; px[p] = (uint32_t)s[1] | ((uint32_t)s[2] << 8) | ((uint32_t)s[3] << 16); }
        }
    }
    uint8_t* drow = dst + (long)y * dst_step;
    if (fast && npx == 4)
    {
        uint32_t* d = reinterpret_cast<uint32_t*>(drow + 3 * x0);
        LVK_STREAM_STORE(d + 0, px[0] | (px[1] << 24)); LVK_STREAM_STORE(d + 1, (px[1] >> 8) | (px[2] << 16)); LVK_STREAM_STORE(d + 2, (px[2] >> 16) | (px[3] << 8));
    }
    else
        for (int p = 0; p < npx; p++) { uint8_t* d = drow + 3 * (x0 + p); d[0] = (uint8_t)px[p]; d[1] = (uint8_t)(px[p] >> 8); d[2] = (uint8_t)(px[p] >> 16); }
}

// ---- dword paths of the four kernels above (cols % 4 == 0, >= 8 columns, planes and pitches 4-byte aligned: what OBS delivers): the same arithmetic on
// whole dwords -- a thread reads its 4 pixels' luma as one dword, its four chroma columns c0 - 1 .. c0 + 2 as one (unaligned) dword per plane or as
// four pixel-pair dwords of the packed layouts, and writes three packed dwords; no byte-sized memory instruction is left.
__device__ __forceinline__ void pack4(uint8_t* __restrict__ dst, uint32_t yw, const uint32_t (&u)[4], const uint32_t (&v)[4])
{
    uint32_t px[4];
#pragma unroll
    for (int p = 0; p < 4; p++) px[p] = ((yw >> (8 * p)) & 0xffu) | (u[p] << 8) | (v[p] << 16);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    LVK_STREAM_STORE(d + 0, px[0] | (px[1] << 24)); LVK_STREAM_STORE(d + 1, (px[1] >> 8) | (px[2] << 16)); LVK_STREAM_STORE(d + 2, (px[2] >> 16) | (px[3] << 8));
}

// chroma of the 4 output columns x0 .. x0 + 3 from the samples s0 .. s3 = columns c0 - 1 .. c0 + 2 (one per byte of w): (other + 3 nearer + 2) >> 2
__device__ __forceinline__ void up2x(uint32_t w, uint32_t (&o)[4])
{
    const uint32_t s0 = w & 0xffu, s1 = (w >> 8) & 0xffu, s2 = (w >> 16) & 0xffu, s3 = w >> 24;
    const uint32_t m1 = s1 + s1 + s1 + 2u, m2 = s2 + s2 + s2 + 2u;
    o[0] = (s0 + m1) >> 2; o[1] = (s2 + m1) >> 2; o[2] = (s1 + m2) >> 2; o[3] = (s3 + m2) >> 2;
}

template <int LAYOUT>
__global__ __launch_bounds__(256)
void k_ingest_422_dw(const uint8_t* __restrict__ p0, int s0, const uint8_t* __restrict__ p1, int s1, const uint8_t* __restrict__ p2, int s2,
                     int rows, int cols, uint8_t* __restrict__ dst, int dst_step)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= cols || y >= rows) return;
    const int cc = cols >> 1, c0 = x0 >> 1;
    struct __attribute__((packed, aligned(1))) P4 { uint32_t w; };
    uint32_t yw, uw, vw;
    if (LAYOUT == 0)
    {
        const bool left = x0 == 0, right = x0 == cols - 4;
        const int lc = left ? 0 : (right ? cc - 4 : c0 - 1);
        yw = *reinterpret_cast<const uint32_t*>(p0 + (long)y * s0 + x0);
        uw = reinterpret_cast<const P4*>(p1 + (long)y * s1 + lc)->w;
        vw = reinterpret_cast<const P4*>(p2 + (long)y * s2 + lc)->w;
        if (left)  { uw = (uw << 8) | (uw & 0xffu); vw = (vw << 8) | (vw & 0xffu); }
        if (right) { uw = (uw >> 8) | (uw & 0xff000000u); vw = (vw >> 8) | (vw & 0xff000000u); }
    }
    else
    {
        // the pixel pairs c0 - 1 .. c0 + 2, one dword each (Y C Y C or C Y C Y); clamping the pair index replicates the edge samples
        const uint32_t* r = reinterpret_cast<const uint32_t*>(p0 + (long)y * s0);
        const uint32_t a = r[max(c0 - 1, 0)], b = r[c0], c = r[c0 + 1], d = r[min(c0 + 2, cc - 1)];
        constexpr int YS = LAYOUT == 3 ? 8 : 0, CS = LAYOUT == 3 ? 0 : 8;          // bit offset of the first luma / chroma byte of a pair
        yw = ((b >> YS) & 0xffu) | (((b >> (YS + 16)) & 0xffu) << 8) | (((c >> YS) & 0xffu) << 16) | (((c >> (YS + 16)) & 0xffu) << 24);
        const uint32_t first = ((a >> CS) & 0xffu) | (((b >> CS) & 0xffu) << 8) | (((c >> CS) & 0xffu) << 16) | (((d >> CS) & 0xffu) << 24);
        const uint32_t second = ((a >> (CS + 16)) & 0xffu) | (((b >> (CS + 16)) & 0xffu) << 8) | (((c >> (CS + 16)) & 0xffu) << 16) | (((d >> (CS + 16)) & 0xffu) << 24);
        uw = LAYOUT != 2 ? first : second; vw = LAYOUT != 2 ? second : first;
    }
    uint32_t u[4], v[4];
    up2x(uw, u); up2x(vw, v);
    pack4(dst + (long)y * dst_step + 3 * x0, yw, u, v);
}

template <int LAYOUT>
__global__ __launch_bounds__(256)
void k_ingest_444_dw(const uint8_t* __restrict__ p0, int s0, const uint8_t* __restrict__ p1, int s1, const uint8_t* __restrict__ p2, int s2,
                     int rows, int cols, uint8_t* __restrict__ dst, int dst_step)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= cols || y >= rows) return;
    uint32_t yw, u[4], v[4];
    if (LAYOUT == 0)
    {
        yw = *reinterpret_cast<const uint32_t*>(p0 + (long)y * s0 + x0);
        const uint32_t uw = *reinterpret_cast<const uint32_t*>(p1 + (long)y * s1 + x0), vw = *reinterpret_cast<const uint32_t*>(p2 + (long)y * s2 + x0);
#pragma unroll
        for (int p = 0; p < 4; p++) { u[p] = (uw >> (8 * p)) & 0xffu; v[p] = (vw >> (8 * p)) & 0xffu; }
    }
    else
    {
        const uint4 q = *reinterpret_cast<const uint4*>(p0 + (long)y * s0 + 4 * (long)x0);      // A Y U V x 4
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        yw = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) { yw |= ((w[p] >> 8) & 0xffu) << (8 * p); u[p] = (w[p] >> 16) & 0xffu; v[p] = w[p] >> 24; }
    }
    pack4(dst + (long)y * dst_step + 3 * x0, yw, u, v);
}

// 4 packed pixels (12 bytes = 3 dwords) -> their channels, one per byte
__device__ __forceinline__ void unpack4(const uint8_t* __restrict__ src, uint32_t& yw, uint32_t& uw, uint32_t& vw)
{
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    const uint32_t a = s[0], b = s[1], c = s[2];            // y0 u0 v0 y1 | u1 v1 y2 u2 | v2 y3 u3 v3
    yw = (a & 0xffu) | ((a >> 24) << 8) | (((b >> 16) & 0xffu) << 16) | (((c >> 8) & 0xffu) << 24);
    uw = ((a >> 8) & 0xffu) | ((b & 0xffu) << 8) | ((b >> 24) << 16) | (((c >> 16) & 0xffu) << 24);
    vw = ((a >> 16) & 0xffu) | (((b >> 8) & 0xffu) << 8) | ((c & 0xffu) << 16) | ((c >> 24) << 24);
}

// packed 444 -> 4:2:2: cv::resize(Size(), 0.5, 1.0, INTER_AREA) on the chroma = saturate_cast<uchar>((a + b) * 0.5f), round half to EVEN
// (resizeAreaFast_'s generic loop; only the 2 x 2 case has the (s + 2) >> 2 vector kernel).  A thread = one pixel pair.
__device__ __forceinline__ uint32_t half_even(uint32_t s) { return (s + ((s >> 1) & 1u)) >> 1; }
template <int LAYOUT>
__global__ __launch_bounds__(256)
void k_egress_422(const uint8_t* __restrict__ src, int src_step, int rows, int cols,
                  uint8_t* __restrict__ p0, int s0, uint8_t* __restrict__ p1, int s1, uint8_t* __restrict__ p2, int s2)
{
    const int cx = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (cx >= cols / 2 || y >= rows) return;
    const uint8_t* s = src + (long)y * src_step + 6 * (long)cx;
    const uint32_t y0 = s[0], y1 = s[3], u = half_even((uint32_t)s[1] + s[4]), v = half_even((uint32_t)s[2] + s[5]);
    if (LAYOUT == 0)
    {
        uint8_t* d = p0 + (long)y * s0 + 2 * cx; d[0] = (uint8_t)y0; d[1] = (uint8_t)y1;
        p1[(long)y * s1 + cx] = (uint8_t)u; p2[(long)y * s2 + cx] = (uint8_t)v;
    }
    else
    {
        constexpr int YOFF = LAYOUT == 3 ? 1 : 0, COFF = 1 - YOFF;
        const uint32_t first = LAYOUT != 2 ? u : v, second = LAYOUT != 2 ? v : u;
        uint8_t* d = p0 + (long)y * s0 + 4 * (long)cx;
        d[YOFF] = (uint8_t)y0; d[2 + YOFF] = (uint8_t)y1; d[COFF] = (uint8_t)first; d[2 + COFF] = (uint8_t)second;
    }
}

// packed 444 -> three planes (cv::split, FrameIngest.cpp:531) or A Y U V with A = 255 (:694-701).  A thread = one pixel.
template <int LAYOUT>
__global__ __launch_bounds__(256)
void k_egress_444(const uint8_t* __restrict__ src, int src_step, int rows, int cols,
                  uint8_t* __restrict__ p0, int s0, uint8_t* __restrict__ p1, int s1, uint8_t* __restrict__ p2, int s2)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const uint8_t* s = src + (long)y * src_step + 3 * (long)x;
    if (LAYOUT == 0) { p0[(long)y * s0 + x] = s[0]; p1[(long)y * s1 + x] = s[1]; p2[(long)y * s2 + x] = s[2]; }
    else *reinterpret_cast<uint32_t*>(p0 + (long)y * s0 + 4 * (long)x) = 255u | ((uint32_t)s[0] << 8) | ((uint32_t)s[1] << 16) | ((uint32_t)s[2] << 24);
}

template <int LAYOUT>
__global__ __launch_bounds__(256)
void k_egress_422_dw(const uint8_t* __restrict__ src, int src_step, int rows, int cols,
                     uint8_t* __restrict__ p0, int s0, uint8_t* __restrict__ p1, int s1, uint8_t* __restrict__ p2, int s2)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= cols || y >= rows) return;
    uint32_t yw, uw, vw;
    unpack4(src + (long)y * src_step + 3 * (long)x0, yw, uw, vw);
    const uint32_t u0 = half_even((uw & 0xffu) + ((uw >> 8) & 0xffu)), u1 = half_even(((uw >> 16) & 0xffu) + (uw >> 24));
    const uint32_t v0 = half_even((vw & 0xffu) + ((vw >> 8) & 0xffu)), v1 = half_even(((vw >> 16) & 0xffu) + (vw >> 24));
    if (LAYOUT == 0)
    {
        *reinterpret_cast<uint32_t*>(p0 + (long)y * s0 + x0) = yw;
        *reinterpret_cast<uint16_t*>(p1 + (long)y * s1 + (x0 >> 1)) = (uint16_t)(u0 | (u1 << 8));
        *reinterpret_cast<uint16_t*>(p2 + (long)y * s2 + (x0 >> 1)) = (uint16_t)(v0 | (v1 << 8));
    }
    else
    {
        const uint32_t f0 = LAYOUT != 2 ? u0 : v0, g0 = LAYOUT != 2 ? v0 : u0, f1 = LAYOUT != 2 ? u1 : v1, g1 = LAYOUT != 2 ? v1 : u1;
        const uint32_t y0 = yw & 0xffu, y1 = (yw >> 8) & 0xffu, y2 = (yw >> 16) & 0xffu, y3 = yw >> 24;
        uint2 o;
        if (LAYOUT == 3) { o.x = f0 | (y0 << 8) | (g0 << 16) | (y1 << 24); o.y = f1 | (y2 << 8) | (g1 << 16) | (y3 << 24); }
        else             { o.x = y0 | (f0 << 8) | (y1 << 16) | (g0 << 24); o.y = y2 | (f1 << 8) | (y3 << 16) | (g1 << 24); }
        *reinterpret_cast<uint2*>(p0 + (long)y * s0 + 2 * (long)x0) = o;
    }
}

template <int LAYOUT>
__global__ __launch_bounds__(256)
void k_egress_444_dw(const uint8_t* __restrict__ src, int src_step, int rows, int cols,
                     uint8_t* __restrict__ p0, int s0, uint8_t* __restrict__ p1, int s1, uint8_t* __restrict__ p2, int s2)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= cols || y >= rows) return;
    uint32_t yw, uw, vw;
    unpack4(src + (long)y * src_step + 3 * (long)x0, yw, uw, vw);
    if (LAYOUT == 0)
    {
        *reinterpret_cast<uint32_t*>(p0 + (long)y * s0 + x0) = yw;
        *reinterpret_cast<uint32_t*>(p1 + (long)y * s1 + x0) = uw;
        *reinterpret_cast<uint32_t*>(p2 + (long)y * s2 + x0) = vw;
    }
    else
    {
        uint4 o;
        o.x = 255u | ((yw & 0xffu) << 8) | ((uw & 0xffu) << 16) | ((vw & 0xffu) << 24);
        o.y = 255u | (((yw >> 8) & 0xffu) << 8) | (((uw >> 8) & 0xffu) << 16) | (((vw >> 8) & 0xffu) << 24);
        o.z = 255u | (((yw >> 16) & 0xffu) << 8) | (((uw >> 16) & 0xffu) << 16) | (((vw >> 16) & 0xffu) << 24);
        o.w = 255u | ((yw >> 24) << 8) | ((uw >> 24) << 16) | ((vw >> 24) << 24);
        *reinterpret_cast<uint4*>(p0 + (long)y * s0 + 4 * (long)x0) = o;
    }
}

} // namespace

// cv::resize(8U, INTER_LINEAR) table (OpenCV 4.8 resize.cpp), coefficients as 11-bit shorts.
int lvk_get_lin8tab(lvk_hip_ctx* ctx, int ssize, int dsize, bool vertical, const Lin8Entry** d_out)
{
    const auto key = std::make_tuple(ssize, dsize, vertical ? 1 : 0);
    auto it = ctx->lin8tabs.find(key);
    if (it != ctx->lin8tabs.end()) { *d_out = it->second; return LVK_HIP_OK; }
    std::vector<Lin8Entry> tab((size_t)dsize);
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; d++)
    {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        Lin8Entry e;
        if (vertical)
        {
            e.s0 = std::min(std::max(s, 0), ssize - 1); e.s1 = std::min(std::max(s + 1, 0), ssize - 1);
            e.a0 = (int)lrintf((1.f - f) * 2048.f); e.a1 = (int)lrintf(f * 2048.f);
        }
        else
        {
            if (s < 0) { f = 0.f; s = 0; }
            bool single = false;
            if (s + 1 >= ssize) { single = true; if (s >= ssize - 1) { f = 0.f; s = ssize - 1; } }
            e.s0 = s; e.s1 = single ? s : s + 1;
            e.a0 = single ? 2048 : (int)lrintf((1.f - f) * 2048.f); e.a1 = single ? 0 : (int)lrintf(f * 2048.f);
        }
        tab[(size_t)d] = e;
    }
    Lin8Entry* d_tab = nullptr;
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_tab, tab.size() * sizeof(Lin8Entry)));
    LVK_HIP_CHECK(ctx, hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(Lin8Entry), hipMemcpyHostToDevice));
    ctx->lin8tabs[key] = d_tab;
    *d_out = d_tab;
    return LVK_HIP_OK;
}

int lvk_launch_ingest_yuv420(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_y, int y_step, const void* d_u, int u_step,
                             const void* d_v, int v_step, int nv12, int rows, int cols, void* d_dst, int dst_step)
{
    LVK_HIP_REQUIRE(ctx, d_y && d_u && (nv12 || d_v) && d_dst && rows > 0 && cols > 0);
    LVK_HIP_REQUIRE(ctx, (rows & 1) == 0 && (cols & 1) == 0);                       // 4:2:0 needs even dimensions
    LVK_HIP_REQUIRE(ctx, y_step >= cols && dst_step >= 3 * cols && u_step >= (nv12 ? cols : cols / 2));
    const Lin8Entry *xt, *yt;
    int rc;
    if ((rc = lvk_get_lin8tab(ctx, cols / 2, cols, false, &xt)) != LVK_HIP_OK) return rc;
    if ((rc = lvk_get_lin8tab(ctx, rows / 2, rows, true, &yt)) != LVK_HIP_OK) return rc;
    const int fast = ((reinterpret_cast<uintptr_t>(d_dst) | (uintptr_t)dst_step) & 3u) == 0 ? 1 : 0;
    const dim3 block(64, 4), grid1((cols / 4 + 63 + (cols % 4 ? 1 : 0)) / 64, (rows + 3) / 4);
    const bool x2 = fast && cols % 4 == 0 && cols >= 16 && rows >= 4 &&
                    ((reinterpret_cast<uintptr_t>(d_y) | (uintptr_t)y_step) & 3u) == 0;
    if (x2)
    {
        const dim3 grid(grid1.x, (rows / 2 + 1 + 3) / 4);                       // one thread row per chroma row pair: k = 0 .. rows / 2
        if (nv12) hipLaunchKernelGGL(k_ingest_yuv420_x2<true>, grid, block, 0, stream, (const uint8_t*)d_y, y_step, (const uint8_t*)d_u, u_step,
                                     (const uint8_t*)d_u, u_step, rows, cols, (uint8_t*)d_dst, dst_step);
        else hipLaunchKernelGGL(k_ingest_yuv420_x2<false>, grid, block, 0, stream, (const uint8_t*)d_y, y_step, (const uint8_t*)d_u, u_step,
                                (const uint8_t*)d_v, v_step, rows, cols, (uint8_t*)d_dst, dst_step);
        LVK_HIP_CHECK(ctx, hipGetLastError());
        return LVK_HIP_OK;
    }
    const dim3 grid = grid1;
    if (nv12)
        hipLaunchKernelGGL(k_ingest_yuv420<true>, grid, block, 0, stream, (const uint8_t*)d_y, y_step, (const uint8_t*)d_u, u_step, (const uint8_t*)d_u, u_step,
                           rows, cols, (uint8_t*)d_dst, dst_step, xt, yt, fast);
    else
        hipLaunchKernelGGL(k_ingest_yuv420<false>, grid, block, 0, stream, (const uint8_t*)d_y, y_step, (const uint8_t*)d_u, u_step, (const uint8_t*)d_v, v_step,
                           rows, cols, (uint8_t*)d_dst, dst_step, xt, yt, fast);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

int lvk_launch_egress_yuv420(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                             void* d_y, int y_step, void* d_u, int u_step, void* d_v, int v_step, int nv12)
{
    LVK_HIP_REQUIRE(ctx, d_src && d_y && d_u && (nv12 || d_v) && rows > 0 && cols > 0);
    LVK_HIP_REQUIRE(ctx, (rows & 1) == 0 && (cols & 1) == 0);
    LVK_HIP_REQUIRE(ctx, src_step >= 3 * cols && y_step >= cols && u_step >= (nv12 ? cols : cols / 2));
    const dim3 block(64, 4), grid((cols / 2 + 63) / 64, (rows / 2 + 3) / 4);
    if (nv12)
        hipLaunchKernelGGL(k_egress_yuv420<true>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, (uint8_t*)d_y, y_step, (uint8_t*)d_u, u_step, (uint8_t*)d_u, u_step);
    else
        hipLaunchKernelGGL(k_egress_yuv420<false>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, (uint8_t*)d_y, y_step, (uint8_t*)d_u, u_step, (uint8_t*)d_v, v_step);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

static inline bool aligned_to(const void* p, int step, unsigned a) { return ((reinterpret_cast<uintptr_t>(p) | (uintptr_t)step) & (a - 1)) == 0; }

// FrameIngest::Select's switch (FrameIngest.cpp:36-75) as one pair of launchers.  video_format = libobs' enum video_format (LVK_VIDEO_FORMAT_*).
int lvk_launch_ingest_obs(lvk_hip_ctx* ctx, hipStream_t stream, int video_format, const void* const d_planes[3], const int steps[3],
                          int rows, int cols, void* d_dst, int dst_step)
{
    LVK_HIP_REQUIRE(ctx, d_planes && steps && d_planes[0] && d_dst && rows > 0 && cols > 0);
    const uint8_t* p0 = (const uint8_t*)d_planes[0]; const uint8_t* p1 = (const uint8_t*)d_planes[1]; const uint8_t* p2 = (const uint8_t*)d_planes[2];
    const int fast = ((reinterpret_cast<uintptr_t>(d_dst) | (uintptr_t)dst_step) & 3u) == 0 ? 1 : 0;
    const dim3 block(64, 4), grid((cols + 255) / 256, (rows + 3) / 4);
    switch (video_format)
    {
    case LVK_VIDEO_FORMAT_I420: case LVK_VIDEO_FORMAT_I40A:
        return lvk_launch_ingest_yuv420(ctx, stream, p0, steps[0], p1, steps[1], p2, steps[2], 0, rows, cols, d_dst, dst_step);
    case LVK_VIDEO_FORMAT_NV12:
        return lvk_launch_ingest_yuv420(ctx, stream, p0, steps[0], p1, steps[1], nullptr, 0, 1, rows, cols, d_dst, dst_step);
    case LVK_VIDEO_FORMAT_I422: case LVK_VIDEO_FORMAT_I42A:
        LVK_HIP_REQUIRE(ctx, p1 && p2 && (cols & 1) == 0 && steps[0] >= cols && steps[1] >= cols / 2 && steps[2] >= cols / 2 && dst_step >= 3 * cols);
        if (fast && cols % 4 == 0 && cols >= 8 && aligned_to(p0, steps[0], 4))
            hipLaunchKernelGGL(k_ingest_422_dw<0>, grid, block, 0, stream, p0, steps[0], p1, steps[1], p2, steps[2], rows, cols, (uint8_t*)d_dst, dst_step);
        else
            hipLaunchKernelGGL(k_ingest_422<0>, grid, block, 0, stream, p0, steps[0], p1, steps[1], p2, steps[2], rows, cols, (uint8_t*)d_dst, dst_step, fast);
        break;
    case LVK_VIDEO_FORMAT_YUY2: case LVK_VIDEO_FORMAT_YVYU: case LVK_VIDEO_FORMAT_UYVY:
        LVK_HIP_REQUIRE(ctx, (cols & 1) == 0 && steps[0] >= 2 * cols && dst_step >= 3 * cols);
        if (fast && cols % 4 == 0 && aligned_to(p0, steps[0], 4))
        {
            if (video_format == LVK_VIDEO_FORMAT_YUY2) hipLaunchKernelGGL(k_ingest_422_dw<1>, grid, block, 0, stream, p0, steps[0], p0, 0, p0, 0, rows, cols, (uint8_t*)d_dst, dst_step);
            else if (video_format == LVK_VIDEO_FORMAT_YVYU) hipLaunchKernelGGL(k_ingest_422_dw<2>, grid, block, 0, stream, p0, steps[0], p0, 0, p0, 0, rows, cols, (uint8_t*)d_dst, dst_step);
            else hipLaunchKernelGGL(k_ingest_422_dw<3>, grid, block, 0, stream, p0, steps[0], p0, 0, p0, 0, rows, cols, (uint8_t*)d_dst, dst_step);
        }
        else if (video_format == LVK_VIDEO_FORMAT_YUY2) hipLaunchKernelGGL(k_ingest_422<1>, grid, block, 0, stream, p0, steps[0], p0, 0, p0, 0, rows, cols, (uint8_t*)d_dst, dst_step, fast);
        else if (video_format == LVK_VIDEO_FORMAT_YVYU) hipLaunchKernelGGL(k_ingest_422<2>, grid, block, 0, stream, p0, steps[0], p0, 0, p0, 0, rows, cols, (uint8_t*)d_dst, dst_step, fast);
        else hipLaunchKernelGGL(k_ingest_422<3>, grid, block, 0, stream, p0, steps[0], p0, 0, p0, 0, rows, cols, (uint8_t*)d_dst, dst_step, fast);
        break;
    case LVK_VIDEO_FORMAT_I444: case LVK_VIDEO_FORMAT_YUVA:
        LVK_HIP_REQUIRE(ctx, p1 && p2 && steps[0] >= cols && steps[1] >= cols && steps[2] >= cols && dst_step >= 3 * cols);
        if (fast && cols % 4 == 0 && aligned_to(p0, steps[0], 4) && aligned_to(p1, steps[1], 4) && aligned_to(p2, steps[2], 4))
            hipLaunchKernelGGL(k_ingest_444_dw<0>, grid, block, 0, stream, p0, steps[0], p1, steps[1], p2, steps[2], rows, cols, (uint8_t*)d_dst, dst_step);
        else
            hipLaunchKernelGGL(k_ingest_444<0>, grid, block, 0, stream, p0, steps[0], p1, steps[1], p2, steps[2], rows, cols, (uint8_t*)d_dst, dst_step, fast);
        break;
    case LVK_VIDEO_FORMAT_AYUV:
        LVK_HIP_REQUIRE(ctx, steps[0] >= 4 * cols && dst_step >= 3 * cols);
        if (fast && cols % 4 == 0 && aligned_to(p0, steps[0], 16))
            hipLaunchKernelGGL(k_ingest_444_dw<1>, grid, block, 0, stream, p0, steps[0], p0, 0, p0, 0, rows, cols, (uint8_t*)d_dst, dst_step);
        else
            hipLaunchKernelGGL(k_ingest_444<1>, grid, block, 0, stream, p0, steps[0], p0, 0, p0, 0, rows, cols, (uint8_t*)d_dst, dst_step, fast);
        break;
    case LVK_VIDEO_FORMAT_Y800:                                   // DirectIngest: upload_planes(src, 1).copyTo(dst)
        LVK_HIP_REQUIRE(ctx, steps[0] >= cols && dst_step >= cols);
        LVK_HIP_CHECK(ctx, hipMemcpy2DAsync(d_dst, (size_t)dst_step, p0, (size_t)steps[0], (size_t)cols, (size_t)rows, hipMemcpyDeviceToDevice, stream));
        return LVK_HIP_OK;
    case LVK_VIDEO_FORMAT_BGR3:
        LVK_HIP_REQUIRE(ctx, steps[0] >= 3 * cols && dst_step >= 3 * cols);
        LVK_HIP_CHECK(ctx, hipMemcpy2DAsync(d_dst, (size_t)dst_step, p0, (size_t)steps[0], 3 * (size_t)cols, (size_t)rows, hipMemcpyDeviceToDevice, stream));
        return LVK_HIP_OK;
    case LVK_VIDEO_FORMAT_RGBA: case LVK_VIDEO_FORMAT_BGRA: case LVK_VIDEO_FORMAT_BGRX:
        // DirectIngest::to_ocl as written (FrameIngest.cpp:743-747): rows * cols * 3 BYTES of the tightly packed 4-byte pixels, viewed as 3-byte pixels
        LVK_HIP_REQUIRE(ctx, steps[0] == 4 * cols && dst_step >= 3 * cols);
        LVK_HIP_CHECK(ctx, hipMemcpy2DAsync(d_dst, (size_t)dst_step, p0, 3 * (size_t)cols, 3 * (size_t)cols, (size_t)rows, hipMemcpyDeviceToDevice, stream));
        return LVK_HIP_OK;
    default:
        return ctx->fail(LVK_HIP_ERR_ARG, "lvk_hip_ingest_obs: video format " + std::to_string(video_format) + " is not one FrameIngest::Select knows");
    }
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

int lvk_launch_egress_obs(lvk_hip_ctx* ctx, hipStream_t stream, int video_format, const void* d_src, int src_step, int rows, int cols,
                          void* const d_planes[3], const int steps[3])
{
    LVK_HIP_REQUIRE(ctx, d_planes && steps && d_planes[0] && d_src && rows > 0 && cols > 0);
    uint8_t* p0 = (uint8_t*)d_planes[0]; uint8_t* p1 = (uint8_t*)d_planes[1]; uint8_t* p2 = (uint8_t*)d_planes[2];
    const uint8_t* src = (const uint8_t*)d_src;
    const dim3 block(64, 4);
    switch (video_format)
    {
    case LVK_VIDEO_FORMAT_I420: case LVK_VIDEO_FORMAT_I40A:
        return lvk_launch_egress_yuv420(ctx, stream, d_src, src_step, rows, cols, p0, steps[0], p1, steps[1], p2, steps[2], 0);
    case LVK_VIDEO_FORMAT_NV12:
        return lvk_launch_egress_yuv420(ctx, stream, d_src, src_step, rows, cols, p0, steps[0], p1, steps[1], nullptr, 0, 1);
    case LVK_VIDEO_FORMAT_I422: case LVK_VIDEO_FORMAT_I42A:
        LVK_HIP_REQUIRE(ctx, p1 && p2 && (cols & 1) == 0 && steps[0] >= cols && steps[1] >= cols / 2 && steps[2] >= cols / 2 && src_step >= 3 * cols);
        if (cols % 4 == 0 && aligned_to(src, src_step, 4) && aligned_to(p0, steps[0], 4) && aligned_to(p1, steps[1], 2) && aligned_to(p2, steps[2], 2))
            hipLaunchKernelGGL(k_egress_422_dw<0>, dim3((cols + 255) / 256, (rows + 3) / 4), block, 0, stream, src, src_step, rows, cols, p0, steps[0], p1, steps[1], p2, steps[2]);
        else
            hipLaunchKernelGGL(k_egress_422<0>, dim3((cols / 2 + 63) / 64, (rows + 3) / 4), block, 0, stream, src, src_step, rows, cols, p0, steps[0], p1, steps[1], p2, steps[2]);
        break;
    case LVK_VIDEO_FORMAT_YUY2: case LVK_VIDEO_FORMAT_YVYU: case LVK_VIDEO_FORMAT_UYVY:
    {
        LVK_HIP_REQUIRE(ctx, (cols & 1) == 0 && steps[0] >= 2 * cols && src_step >= 3 * cols);
        const dim3 grid((cols / 2 + 63) / 64, (rows + 3) / 4), grid4((cols + 255) / 256, (rows + 3) / 4);
        if (cols % 4 == 0 && aligned_to(src, src_step, 4) && aligned_to(p0, steps[0], 8))
        {
            if (video_format == LVK_VIDEO_FORMAT_YUY2) hipLaunchKernelGGL(k_egress_422_dw<1>, grid4, block, 0, stream, src, src_step, rows, cols, p0, steps[0], p0, 0, p0, 0);
            else if (video_format == LVK_VIDEO_FORMAT_YVYU) hipLaunchKernelGGL(k_egress_422_dw<2>, grid4, block, 0, stream, src, src_step, rows, cols, p0, steps[0], p0, 0, p0, 0);
            else hipLaunchKernelGGL(k_egress_422_dw<3>, grid4, block, 0, stream, src, src_step, rows, cols, p0, steps[0], p0, 0, p0, 0);
        }
        else if (video_format == LVK_VIDEO_FORMAT_YUY2) hipLaunchKernelGGL(k_egress_422<1>, grid, block, 0, stream, src, src_step, rows, cols, p0, steps[0], p0, 0, p0, 0);
        else if (video_format == LVK_VIDEO_FORMAT_YVYU) hipLaunchKernelGGL(k_egress_422<2>, grid, block, 0, stream, src, src_step, rows, cols, p0, steps[0], p0, 0, p0, 0);
        else hipLaunchKernelGGL(k_egress_422<3>, grid, block, 0, stream, src, src_step, rows, cols, p0, steps[0], p0, 0, p0, 0);
        break;
    }
    case LVK_VIDEO_FORMAT_I444: case LVK_VIDEO_FORMAT_YUVA:
        LVK_HIP_REQUIRE(ctx, p1 && p2 && steps[0] >= cols && steps[1] >= cols && steps[2] >= cols && src_step >= 3 * cols);
        if (cols % 4 == 0 && aligned_to(src, src_step, 4) && aligned_to(p0, steps[0], 4) && aligned_to(p1, steps[1], 4) && aligned_to(p2, steps[2], 4))
            hipLaunchKernelGGL(k_egress_444_dw<0>, dim3((cols + 255) / 256, (rows + 3) / 4), block, 0, stream, src, src_step, rows, cols, p0, steps[0], p1, steps[1], p2, steps[2]);
        else
            hipLaunchKernelGGL(k_egress_444<0>, dim3((cols + 63) / 64, (rows + 3) / 4), block, 0, stream, src, src_step, rows, cols, p0, steps[0], p1, steps[1], p2, steps[2]);
        break;
    case LVK_VIDEO_FORMAT_AYUV:
        LVK_HIP_REQUIRE(ctx, steps[0] >= 4 * cols && (steps[0] & 3) == 0 && (reinterpret_cast<uintptr_t>(p0) & 3u) == 0 && src_step >= 3 * cols);
        if (cols % 4 == 0 && aligned_to(src, src_step, 4) && aligned_to(p0, steps[0], 16))
            hipLaunchKernelGGL(k_egress_444_dw<1>, dim3((cols + 255) / 256, (rows + 3) / 4), block, 0, stream, src, src_step, rows, cols, p0, steps[0], p0, 0, p0, 0);
        else
            hipLaunchKernelGGL(k_egress_444<1>, dim3((cols + 63) / 64, (rows + 3) / 4), block, 0, stream, src, src_step, rows, cols, p0, steps[0], p0, 0, p0, 0);
        break;
    case LVK_VIDEO_FORMAT_Y800:
        LVK_HIP_REQUIRE(ctx, steps[0] >= cols && src_step >= cols);
        LVK_HIP_CHECK(ctx, hipMemcpy2DAsync(p0, (size_t)steps[0], src, (size_t)src_step, (size_t)cols, (size_t)rows, hipMemcpyDeviceToDevice, stream));
        return LVK_HIP_OK;
    case LVK_VIDEO_FORMAT_BGR3:
        LVK_HIP_REQUIRE(ctx, steps[0] >= 3 * cols && src_step >= 3 * cols);
        LVK_HIP_CHECK(ctx, hipMemcpy2DAsync(p0, (size_t)steps[0], src, (size_t)src_step, 3 * (size_t)cols, (size_t)rows, hipMemcpyDeviceToDevice, stream));
        return LVK_HIP_OK;
    case LVK_VIDEO_FORMAT_RGBA: case LVK_VIDEO_FORMAT_BGRA: case LVK_VIDEO_FORMAT_BGRX:
        // DirectIngest::to_obs: download_planes(src, dst) writes rows * cols * 3 bytes at data[0] (FrameIngest.cpp:751-753); the last quarter stays
        LVK_HIP_REQUIRE(ctx, steps[0] == 4 * cols && src_step >= 3 * cols);
        LVK_HIP_CHECK(ctx, hipMemcpy2DAsync(p0, 3 * (size_t)cols, src, (size_t)src_step, 3 * (size_t)cols, (size_t)rows, hipMemcpyDeviceToDevice, stream));
        return LVK_HIP_OK;
    default:
        return ctx->fail(LVK_HIP_ERR_ARG, "lvk_hip_egress_obs: video format " + std::to_string(video_format) + " is not one FrameIngest::Select knows");
    }
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}


extern "C" {

int lvk_hip_ingest_yuv420(lvk_hip_ctx* ctx, const void* d_y, int y_step, const void* d_u, int u_step, const void* d_v, int v_step, int nv12,
                          int rows, int cols, void* d_dst, int dst_step)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_ingest_yuv420(ctx, ctx->stream, d_y, y_step, d_u, u_step, d_v, v_step, nv12, rows, cols, d_dst, dst_step);
}

int lvk_hip_egress_yuv420(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols,
                          void* d_y, int y_step, void* d_u, int u_step, void* d_v, int v_step, int nv12)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_egress_yuv420(ctx, ctx->stream, d_src, src_step, rows, cols, d_y, y_step, d_u, u_step, d_v, v_step, nv12);
}

int lvk_hip_ingest_obs(lvk_hip_ctx* ctx, int video_format, const void* const d_planes[3], const int steps[3], int rows, int cols, void* d_dst, int dst_step)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_ingest_obs(ctx, ctx->stream, video_format, d_planes, steps, rows, cols, d_dst, dst_step);
}

int lvk_hip_egress_obs(lvk_hip_ctx* ctx, int video_format, const void* d_src, int src_step, int rows, int cols, void* const d_planes[3], const int steps[3])
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_egress_obs(ctx, ctx->stream, video_format, d_src, src_step, rows, cols, d_planes, steps);
}

int lvk_hip_obs_frame_format(int video_format)
{
    switch (video_format)                                         // FrameIngest.cpp:476-490,562,604-611,672,715-726: the VideoFrame::Format each ingest declares
    {
    case LVK_VIDEO_FORMAT_I420: case LVK_VIDEO_FORMAT_NV12: case LVK_VIDEO_FORMAT_I422: case LVK_VIDEO_FORMAT_I444: case LVK_VIDEO_FORMAT_I40A:
    case LVK_VIDEO_FORMAT_I42A: case LVK_VIDEO_FORMAT_YUVA: case LVK_VIDEO_FORMAT_YUY2: case LVK_VIDEO_FORMAT_YVYU: case LVK_VIDEO_FORMAT_UYVY:
    case LVK_VIDEO_FORMAT_AYUV: return LVK_FORMAT_YUV;
    case LVK_VIDEO_FORMAT_Y800: return LVK_FORMAT_GRAY;
    case LVK_VIDEO_FORMAT_RGBA: return LVK_FORMAT_RGB;
    case LVK_VIDEO_FORMAT_BGRX: case LVK_VIDEO_FORMAT_BGRA: case LVK_VIDEO_FORMAT_BGR3: return LVK_FORMAT_BGR;
    default: return LVK_HIP_ERR_ARG;
    }
}

} // extern "C"

LVK_TL_EXPORT(ingest)
