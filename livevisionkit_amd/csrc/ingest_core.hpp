// (ingest_core.hpp: the per-thread body of the 4:2:0 -> packed 4:4:4 conversion, shared by ingest.hip -- the conversion as a kernel of its own -- and
//  remap.hip -- the same conversion as side work of the VALU-bound remap kernel, k_remap_*_420_ingest.  Anonymous namespace: one copy per translation unit.)
#pragma once
#include "lvk_hip_internal.hpp"

namespace {

// streaming stores: the converted frame is read N pushes later, see remap_core.hpp
#ifndef LVK_STREAM_STORE
#define LVK_STREAM_STORE(ptr, v) __builtin_nontemporal_store((uint32_t)(v), (ptr))
#endif

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
// (x0, k): the thread's first output column and its chroma row pair -- luma rows 2k - 1 (odd) and 2k (even), k = 0 .. rows / 2.
template <bool NV12>
__device__ __forceinline__ void ingest420_x2_thread(const uint8_t* __restrict__ yp, int y_step, const uint8_t* __restrict__ up, int u_step,
                                                    const uint8_t* __restrict__ vp, int v_step, int rows, int cols, uint8_t* __restrict__ dst, int dst_step,
                                                    int x0, int k)
{
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

// the conversion's work units: 64 x 4 threads (256 output columns x 4 chroma row pairs) each, row-major
struct Ingest420Args
{
    const uint8_t* __restrict__ y; int y_step; const uint8_t* __restrict__ u; int u_step; const uint8_t* __restrict__ v; int v_step;
    int rows, cols; uint8_t* __restrict__ dst; int dst_step;
    int units_x, units;                 // ceil(cols / 256), units_x * ceil((rows / 2 + 1) / 4); units == 0: no side work
};
inline Ingest420Args ingest420_args(const void* y, int y_step, const void* u, int u_step, const void* v, int v_step, int rows, int cols, void* dst, int dst_step)
{
    Ingest420Args a{(const uint8_t*)y, y_step, (const uint8_t*)u, u_step, (const uint8_t*)v, v_step, rows, cols, (uint8_t*)dst, dst_step, 0, 0};
    a.units_x = (cols / 4 + 63) / 64;
    a.units = a.units_x * ((rows / 2 + 1 + 3) / 4);
    return a;
}
template <bool NV12>
__device__ __forceinline__ void ingest420_x2_unit(const Ingest420Args& a, int unit)
{
    const int uy = unit / a.units_x, ux = unit - uy * a.units_x;
    ingest420_x2_thread<NV12>(a.y, a.y_step, a.u, a.u_step, a.v, a.v_step, a.rows, a.cols, a.dst, a.dst_step,
                              (ux * 64 + (int)(threadIdx.x & 63)) * 4, uy * 4 + (int)(threadIdx.x >> 6));
}

} // namespace
