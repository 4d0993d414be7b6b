// (remap_core.hpp: the device code and host helpers shared by remap.hip -- the packed and 4:2:0 kernels and their launchers -- and remap_obs.hip -- the
//  fused remap + egress kernels of the other OBS video formats.  Everything lives in an anonymous namespace: each translation unit has its own copy.)
//
// Dense frame remap for gfx950: the EASU (edge adaptive, 12-tap) resampler driven either by a 3x3
// homography or by a warp mesh that is interpolated inside the kernel.
//
// Replaces lvk::remap x2 (reference: LiveVisionKit/Functions/Image.cpp:28-151) and the OpenCL kernels
// easu_remap / easu_remap_homography (LiveVisionKit/Functions/OpenCL/Sources/FSR.cl:362-452) including
// WarpMesh::apply's map construction (LiveVisionKit/Math/WarpMesh.cpp:183-223).
//
// Arithmetic contract (must stay in lock-step with the specification the tests check against): the binary32 operation
// sequence the reference's OpenCL source compiles to for this device (the GPU tests check bit-identical
// output): no implicit contraction (the file is compiled with -ffp-contract=off), fused multiply-adds exactly where clang's
// FP_CONTRACT ON forms them (written as fma()), and native_recip / `1.0f / x` = the device reciprocal v_rcp_f32.
//
// Work decomposition: see remap_strip() -- 256 x 4 output strips, 4 pixels per thread, taps gathered with unaligned
// dwordx2/x4 loads, XCD-aware strip order.
#pragma once
#include "lvk_hip_internal.hpp"

#include <climits>
#include <cstdlib>
#include <cstring>
#include <cmath>

namespace {

struct HomographyArgs { float h[9]; };

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float min_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float max_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float abs_(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ float rcp_lo(float a) { return __uint_as_float(0x7ef07ebbu - __float_as_uint(a)); }   // FSR.cl:65
__device__ __forceinline__ float rsq_lo(float a) { return __uint_as_float(0x5f347d74u - (__float_as_uint(a) >> 1)); } // FSR.cl:60
// FSR.cl:79 ASatF1 = clamp(x, 0, 1), for a FINITE argument (every use below: a product of finite, non-negative factors): the median of
// (x, 0, 1), which the backend turns into the `clamp` output modifier of the instruction that produces x -- the multiply itself -- instead
// of a v_min_f32 / v_max_f32 pair behind it (16 instructions of the slow issue class per output pixel)
__device__ __forceinline__ float sat_(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
// min(hi, max(lo, x)) for lo <= hi and x not NaN: one v_med3_f32
__device__ __forceinline__ float clamp3_(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// native_recip(x) and OpenCL's `1.0f / x` as the reference's kernels compute them on gfx950: v_frexp_mant, v_rcp_f32, v_frexp_exp,
// v_ldexp (denormal-safe).  For a normal argument with a normal result the plain instruction gives the same bits (all 2^32 inputs
// checked, scripts/rcp_probe.hip): rcp_native() is used where the argument's range is known, rcp_cl() where the caller controls it.
__device__ __forceinline__ float rcp_native(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rcp_cl(float x)
{
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(x)), -__builtin_amdgcn_frexp_expf(x));
}

struct F3 { float x, y, z; };

// (Round 5, measured and rejected: the 256 products convert_float(uchar) * norm_factor in an LDS table, one ds_read_b32 per channel behind a
//  byte-select shift instead of v_cvt_f32_ubyteN + v_mul_f32 -- 36 VALU instructions per pixel fewer (1 949 instead of 2 066 static per
//  4-pixel thread, same bits), and 94.5 instead of 82.5 us alone / 117.5 instead of 109.3 us next to the tracker: 36 data-dependent LDS reads
//  per pixel cost more in the LDS pipe than the 36 instructions they save in the VALU.  profiles/r05_ab_remap_lds_table.txt.)
// (Round 6, measured and rejected: the conversion without v_cvt_f32_ubyteN -- the byte OR-ed into the mantissa of 2^23 by an SDWA byte select, then
//  ONE fused multiply-add (2^23 + b) * n - 2^23 * n, which is float(b) * n bit for bit because 2^23 * n is exact: 144 instructions of the
//  4-cycle class per thread fewer, same bits (68 parity tests), and 109 instead of 88 us: an SDWA operand costs a full issue round and, unlike
//  a conversion, does not share it with a neighbouring fma (scripts/valu_peak.hip: fma : or_sdwa 1:1 = 3.65 cycles per instruction, fma :
//  cvt_ubyte 1:1 = 2.3).  profiles/r06_ab_remap_variants.txt.)
__device__ __forceinline__ F3 unpack3(uint32_t lo_bytes)   // bytes 0,1,2 of the dword
{
    const float norm_factor = 0.00392156862f;               // FSR.cl:205
    F3 r;
#if defined(LVK_EASU_TOLERANT) && LVK_EASU_TOLERANT >= 2
    // (tolerance-mode A / B partner, level 2, never the product build: the colours stay in 0 .. 255, only the luma is normalised -- easu_core below)
    (void)norm_factor;
    r.x = (float)(lo_bytes & 0xffu);
    r.y = (float)((lo_bytes >> 8) & 0xffu);
    r.z = (float)((lo_bytes >> 16) & 0xffu);
#else
    r.x = (float)(lo_bytes & 0xffu) * norm_factor;
    r.y = (float)((lo_bytes >> 8) & 0xffu) * norm_factor;
    r.z = (float)((lo_bytes >> 16) & 0xffu) * norm_factor;
#endif
    return r;
}

template <bool YUV>
__device__ __forceinline__ float luma(const F3& p)
{
    // FSR.cl:229-241 (the YUV program is the one that uses the 3-channel pseudo luma)
#if defined(LVK_EASU_TOLERANT) && LVK_EASU_TOLERANT >= 2
    // the direction analysis keeps the reference's scale: its bit-trick reciprocals (rcp_lo / rsq_lo) have a mantissa-dependent error of several
    // percent, so an analysis on 255 x the luma would shape other kernels (> 1 LSB on edges); 0.5 x + y + 0.5 z is exact on bytes, one rounding follows
    return (YUV ? fma_(p.z, 0.5f, fma_(p.x, 0.5f, p.y)) : p.x) * 0.00392156862f;
#else
    return YUV ? fma_(p.z, 0.5f, fma_(p.x, 0.5f, p.y)) : p.x;
#endif
}

__device__ __forceinline__ void accumulate(float& dirx, float& diry, float& len, float w,
                                           float lA, float lB, float lC, float lD, float lE)
{
    // FSR.cl:131-176
    const float dc = lD - lC, cb = lC - lB;
    float lenX = rcp_lo(max_(abs_(dc), abs_(cb)));
    const float dirX = lD - lB;
    dirx = fma_(dirX, w, dirx);
    lenX = sat_(abs_(dirX) * lenX);
    lenX *= lenX;
    len = fma_(lenX, w, len);
    const float ec = lE - lC, ca = lC - lA;
    float lenY = rcp_lo(max_(abs_(ec), abs_(ca)));
    const float dirY = lE - lA;
    diry = fma_(dirY, w, diry);
    lenY = sat_(abs_(dirY) * lenY);
    lenY *= lenY;
    len = fma_(lenY, w, len);
}

__device__ __forceinline__ void tap(F3& aC, float& aW, float offx, float offy, float dirx, float diry,
                                    float lenx, float leny, float lob, float clp, const F3& c)
{
    // FSR.cl:98-126
    float vx = fma_(offx, dirx, offy * diry);
    float vy = fma_(offx, -diry, offy * dirx);
    vx *= lenx;
    vy *= leny;
    const float d2 = min_(fma_(vx, vx, vy * vy), clp);
    float wA = fma_(lob, d2, -1.0f);
    float wB = fma_(2.0f / 5.0f, d2, -1.0f);
    wA *= wA;
    wB = fma_(25.0f / 16.0f, wB * wB, -(25.0f / 16.0f - 1.0f));
    const float w = wB * wA;
    aC.x = fma_(c.x, w, aC.x);
    aC.y = fma_(c.y, w, aC.y);
    aC.z = fma_(c.z, w, aC.z);
    aW += w;
}

// Unaligned little-endian loads straight from global memory (gfx950 runs in unaligned access mode;
// these lower to single global_load_dword / dwordx2 / dwordx4).
struct __attribute__((packed, aligned(1))) U16B { uint32_t w[4]; };
struct __attribute__((packed, aligned(1))) U8B { uint32_t w[2]; };
struct __attribute__((packed, aligned(1))) U4B { uint32_t w; };

__device__ __forceinline__ uint32_t byte_window(uint32_t lo, uint32_t hi, int shift_bytes)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * shift_bytes));
}

// One source pixel as the kernel consumes it: the three normalised channels and the EASU luma.
template <bool YUV>
__device__ __forceinline__ float4 make_tap(uint32_t lo_bytes)
{
    const F3 p = unpack3(lo_bytes);
    return make_float4(p.x, p.y, p.z, luma<YUV>(p));
}

// FSR.cl:181-318 on the 12 taps  b c / e f g h / i j k l / n o  (order of the array below).  Returns 0x00ZZYYXX.
enum { TB, TC, TE, TF, TG, TH_, TI, TJ, TK, TL, TN, TO };
__device__ __forceinline__ uint32_t easu_core(const float4 t[12], float ppx, float ppy)
{
    // FSR.cl:244-249
    float len = 0.0f, dirx = 0.0f, diry = 0.0f;
    const float omx = 1.0f - ppx, omy = 1.0f - ppy;
    accumulate(dirx, diry, len, omx * omy, t[TB].w, t[TE].w, t[TF].w, t[TG].w, t[TJ].w);
    accumulate(dirx, diry, len, ppx * omy, t[TC].w, t[TF].w, t[TG].w, t[TH_].w, t[TK].w);
    accumulate(dirx, diry, len, omx * ppy, t[TF].w, t[TI].w, t[TJ].w, t[TK].w, t[TN].w);
    accumulate(dirx, diry, len, ppx * ppy, t[TG].w, t[TJ].w, t[TK].w, t[TL].w, t[TO].w);

    // FSR.cl:252-258
    float dirR = dirx * dirx + diry * diry;          // two statements in FSR.cl (dir2 = dir * dir; dir2.x + dir2.y): not contracted
    const bool zro = dirR < (1.0f / 32768.0f);
    dirR = rsq_lo(dirR);
    dirR = zro ? 1.0f : dirR;
    dirx = zro ? 1.0f : dirx;
    dirx *= dirR;
    diry *= dirR;

    // FSR.cl:261-277
    len = len * 0.5f;
    len *= len;
    const float stretch = fma_(dirx, dirx, diry * diry) * rcp_lo(max_(abs_(dirx), abs_(diry)));
    const float len2x = fma_(stretch - 1.0f, len, 1.0f);
    const float len2y = fma_(-0.5f, len, 1.0f);
    const float lob = fma_((1.0f / 4.0f - 0.04f) - 0.5f, len, 0.5f);
    const float clp = rcp_lo(lob);

    // FSR.cl:284-296
    const float4 &f = t[TF], &g = t[TG], &j = t[TJ], &k = t[TK];
    const F3 mi4{ min_(f.x, min_(g.x, min_(j.x, k.x))), min_(f.y, min_(g.y, min_(j.y, k.y))), min_(f.z, min_(g.z, min_(j.z, k.z))) };
    const F3 ma4{ max_(f.x, max_(g.x, max_(j.x, k.x))), max_(f.y, max_(g.y, max_(j.y, k.y))), max_(f.z, max_(g.z, max_(j.z, k.z))) };

#ifdef LVK_EASU_TOLERANT
    // Tolerance-mode A / B partner (SURVEY 8c allows <= 1 LSB / PSNR >= 50 dB for the remap), measured in round 6 and NOT shipped: level 1 (the two regroupings
    // below) is within 1 LSB everywhere and 6-7 % faster, level 2 (raw colours as well) 10 % faster and up to 22 LSB off where the reference's analysis
    // amplifies the rounding noise of its own normalisation (profiles/r06_ab_remap_tolerant.txt).  Same 12 taps, same window,
    // algebraically regrouped -- not the reference's operation sequence:
    //   * a tap's distance d2 = |diag(len2) R(dir) off|^2 is a quadratic form in off = (i - ppx, j - ppy): d2 = (q11 ox + 2 q12 oy) ox + q22 oy^2, two
    //     fused multiply-adds per tap over per-row / per-pixel terms instead of rotate (2) + scale (2) + square (2);
    //   * the window (25/16 (2/5 u - 1)^2 - 9/16) (lob u - 1)^2 = ((u / 4 - 5 / 4) u + 1) (lob u - 1)^2: five operations instead of six;
    //   * the colours stay in 0 .. 255 (36 normalising multiplies fewer, 12 for the luma more); the closing x 255 becomes x (norm_factor x 255), which
    //     maps every integer exactly as the reference's round trip does (k -> k - 1 for k >= 1: 0.00392156862f is below 1 / 255).
    F3 aC{0.0f, 0.0f, 0.0f};
    float aW = 0.0f;
    {
        const float a = len2x * dirx, b = len2x * diry, c = len2y * diry, d = len2y * dirx;      // rows of diag(len2) R: (a, b), (-c, d)
        const float q11 = fma_(a, a, c * c), q22 = fma_(b, b, d * d);
        const float q12 = fma_(a, b, -(c * d));
        const float q12x2 = q12 + q12;
        const float ox[4] = {-1.0f - ppx, 0.0f - ppx, 1.0f - ppx, 2.0f - ppx};
        const float oy[4] = {-1.0f - ppy, 0.0f - ppy, 1.0f - ppy, 2.0f - ppy};
        float E[4], B[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { E[j] = q12x2 * oy[j]; B[j] = (q22 * oy[j]) * oy[j]; }
#define LVK_TAP(I, J, T)                                                                   \
        {                                                                                  \
            const float u = min_(fma_(fma_(q11, ox[I], E[J]), ox[I], B[J]), clp);          \
            const float sA = fma_(lob, u, -1.0f);                                          \
            const float qB = fma_(fma_(u, 0.25f, -1.25f), u, 1.0f);                        \
            const float w = qB * (sA * sA);                                                \
            aC.x = fma_(t[T].x, w, aC.x);                                                  \
            aC.y = fma_(t[T].y, w, aC.y);                                                  \
            aC.z = fma_(t[T].z, w, aC.z);                                                  \
            aW += w;                                                                       \
        }
        LVK_TAP(1, 0, TB) LVK_TAP(2, 0, TC)
        LVK_TAP(0, 1, TE) LVK_TAP(1, 1, TF) LVK_TAP(2, 1, TG) LVK_TAP(3, 1, TH_)
        LVK_TAP(0, 2, TI) LVK_TAP(1, 2, TJ) LVK_TAP(2, 2, TK) LVK_TAP(3, 2, TL)
        LVK_TAP(1, 3, TN) LVK_TAP(2, 3, TO)
#undef LVK_TAP
    }
    const float rW = rcp_native(aW);
#if LVK_EASU_TOLERANT >= 2
    const float out_scale = 0.99999994f;            // float(0.00392156862f * 255): trunc(k * out_scale) == trunc((k * norm_factor) * 255) for every byte k
#else
    const float out_scale = 255.0f;                 // level 1: the taps are normalised as in the reference
#endif
    const float px = clamp3_(aC.x * rW, mi4.x, ma4.x);
    const float py = clamp3_(aC.y * rW, mi4.y, ma4.y);
    const float pz = clamp3_(aC.z * rW, mi4.z, ma4.z);
    const uint32_t ux = (uint32_t)(int)(px * out_scale) & 0xffu;
    const uint32_t uy = (uint32_t)(int)(py * out_scale) & 0xffu;
    const uint32_t uz = (uint32_t)(int)(pz * out_scale) & 0xffu;
#else
    // FSR.cl:299-313
    F3 aC{0.0f, 0.0f, 0.0f};
    float aW = 0.0f;
#define LVK_TAP(ox, oy, T) tap(aC, aW, (ox) - ppx, (oy) - ppy, dirx, diry, len2x, len2y, lob, clp, F3{t[T].x, t[T].y, t[T].z})
    LVK_TAP( 0.0f, -1.0f, TB);
    LVK_TAP( 1.0f, -1.0f, TC);
    LVK_TAP(-1.0f,  1.0f, TI);
    LVK_TAP( 0.0f,  1.0f, TJ);
    LVK_TAP( 0.0f,  0.0f, TF);
    LVK_TAP(-1.0f,  0.0f, TE);
    LVK_TAP( 1.0f,  1.0f, TK);
    LVK_TAP( 2.0f,  1.0f, TL);
    LVK_TAP( 2.0f,  0.0f, TH_);
    LVK_TAP( 1.0f,  0.0f, TG);
    LVK_TAP( 0.0f,  2.0f, TN);
    LVK_TAP( 1.0f,  2.0f, TO);
#undef LVK_TAP

    // FSR.cl:316-317
    // aW: the centre taps alone contribute > 0.5 and no tap reaches 2, so 1/aW and aW are normal numbers
    const float rW = rcp_native(aW);
    // min(ma4, max(mi4, v)) with mi4 <= ma4 and v finite (aW > 0.5, |aC| bounded): the median of the three
    const float px = clamp3_(aC.x * rW, mi4.x, ma4.x);
    const float py = clamp3_(aC.y * rW, mi4.y, ma4.y);
    const float pz = clamp3_(aC.z * rW, mi4.z, ma4.z);
    const uint32_t ux = (uint32_t)(int)(px * 255.0f) & 0xffu;
    const uint32_t uy = (uint32_t)(int)(py * 255.0f) & 0xffu;
    const uint32_t uz = (uint32_t)(int)(pz * 255.0f) & 0xffu;
#endif
    return ux | (uy << 8) | (uz << 16);
}

// EASU with the 12 taps gathered straight from global memory (8 + 16 + 16 + 8 byte loads like FSR.cl:196-202).
// Addressing: ONE 32-bit byte offset per pixel (a frame is < 4 GB: the launchers assert it) against four block-uniform row bases
// (src, src + step - 3, src + 2 step - 3, src + 3 step: scalar registers, computed once per kernel) -- the loads take the form
// global_load_dwordx2/x4 v, v_off, s[base:base+1], and the per-pixel address arithmetic is one v_mad_u32_u24-class instruction instead of
// the 2 x v_mad_u64_u32 + 4-5 x v_lshl_add_u64 (slow issue class) that 64-bit per-row pointers cost (round-4 VERDICT, item 5).
struct TapBases { const uint8_t* __restrict__ r0; const uint8_t* __restrict__ r1; const uint8_t* __restrict__ r2; const uint8_t* __restrict__ r3; };
__device__ __forceinline__ TapBases tap_bases(const uint8_t* __restrict__ src, int step)
{
    return TapBases{src, src + step - 3, src + 2 * (long)step - 3, src + 3 * (long)step};
}

template <bool YUV>
__device__ __forceinline__ uint32_t easu_gather(const TapBases& tb, int step, int sx, int sy, float ppx, float ppy)
{
    // sy >= 1, sx >= 1 here (interior pixels only): the offset of tap b (row sy - 1, column sx) is non-negative
    // (24-bit operands: rows and the row pitch are far below 2^24 -- v_mad_u32_u24, not the quarter-rate v_mad_u64_u32 of a full 32-bit product)
    const uint32_t off = __umul24((uint32_t)(sy - 1), (uint32_t)step) + 3u * (uint32_t)sx;
    const U8B r0 = *reinterpret_cast<const U8B*>(tb.r0 + off);      // b, c
    const U16B r1 = *reinterpret_cast<const U16B*>(tb.r1 + off);    // e, f, g, h
    const U16B r2 = *reinterpret_cast<const U16B*>(tb.r2 + off);    // i, j, k, l
    const U8B r3 = *reinterpret_cast<const U8B*>(tb.r3 + off);      // n, o
    float4 t[12];
    t[TB] = make_tap<YUV>(r0.w[0]);                                t[TC] = make_tap<YUV>(byte_window(r0.w[0], r0.w[1], 3));
    t[TE] = make_tap<YUV>(r1.w[0]);                                t[TF] = make_tap<YUV>(byte_window(r1.w[0], r1.w[1], 3));
    t[TG] = make_tap<YUV>(byte_window(r1.w[1], r1.w[2], 2));       t[TH_] = make_tap<YUV>(byte_window(r1.w[2], 0u, 1));
    t[TI] = make_tap<YUV>(r2.w[0]);                                t[TJ] = make_tap<YUV>(byte_window(r2.w[0], r2.w[1], 3));
    t[TK] = make_tap<YUV>(byte_window(r2.w[1], r2.w[2], 2));       t[TL] = make_tap<YUV>(byte_window(r2.w[2], 0u, 1));
    t[TN] = make_tap<YUV>(r3.w[0]);                                t[TO] = make_tap<YUV>(byte_window(r3.w[0], r3.w[1], 3));
    return easu_core(t, ppx, ppy);
}

// the object at `base` + a 32-bit byte offset (base block-uniform: global_load v, v_off, s[base:base+1])
template <class T>
__device__ __forceinline__ T at_byte(const void* __restrict__ base, uint32_t byte_off)
{
    return *reinterpret_cast<const T*>(static_cast<const uint8_t*>(base) + byte_off);
}

// ---- coordinate generators: destination pixel -> source coordinate -------------------------------------------
struct HomographyCoord      // FSR.cl:422-430
{
    HomographyArgs H; int off_x, off_y;
    __device__ __forceinline__ void operator()(int x, int y, float& subx, float& suby) const
    {
        const float fx = (float)x, fy = (float)y;
        // `r.x * fx + r.y * fy + r.z` = ((r.x * fx) + (r.y * fy)) + r.z: clang fuses the first product only
        const float dz = rcp_cl(fma_(H.h[6], fx, H.h[7] * fy) + H.h[8]);
        const float ox = (fma_(H.h[0], fx, H.h[1] * fy) + H.h[2]) * dz - fx;
        const float oy = (fma_(H.h[3], fx, H.h[4] * fy) + H.h[5]) * dz - fy;
        subx = (float)(x + off_x) + ox;
        suby = (float)(y + off_y) + oy;
    }
};

// The mesh in LDS (meshes up to MESH_LDS_FLOATS values: the 16 x 16 preset is 512): a pixel's coordinate is table entry -> 4 mesh vertices ->
// tap rows, three DEPENDENT memory round trips where the homography kernels have one; with the vertices a ds_read away the mesh kernel runs
// 110.5 -> 106.6 us alone, 131 -> 122 us next to the vector-field tracker, four concurrent field streams 5 810 -> 6 030 frames/s
// (profiles/r05_ab_mesh_in_lds.txt).  Filled once per block by mesh_to_lds().
#ifndef LVK_MESH_LDS_FLOATS
#define LVK_MESH_LDS_FLOATS 2048
#endif
constexpr int MESH_LDS_FLOATS = LVK_MESH_LDS_FLOATS;               // 0: always the global-memory path (A / B partner)
__shared__ float s_mesh_lds[MESH_LDS_FLOATS > 0 ? MESH_LDS_FLOATS : 1];
__device__ __forceinline__ bool mesh_to_lds(const float* __restrict__ mesh, int mesh_floats)
{
    if (mesh_floats > MESH_LDS_FLOATS) return false;                   // block-uniform
    for (int i = (int)threadIdx.x; i < mesh_floats; i += (int)blockDim.x) s_mesh_lds[i] = mesh[i];
    __syncthreads();
    return true;
}

template <bool LDS>
struct MeshCoordT           // WarpMesh.cpp:190-191 per pixel (HResizeLinear, VResizeLinear, * (cols, rows)) + FSR.cl:381
{
    const float* __restrict__ mesh; int mesh_cols;
    const LinTabEntry* __restrict__ xtab; const LinTabEntry* __restrict__ ytab;
    float sw, sh;
    __device__ __forceinline__ float vertex(uint32_t byte_off) const
    {
        if constexpr (LDS) return *reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(s_mesh_lds) + byte_off);
        else return at_byte<float>(mesh, byte_off);
    }
    __device__ __forceinline__ void operator()(int x, int y, float& subx, float& suby) const
    {
        // (32-bit BYTE offsets against the block-uniform bases: the table and mesh loads take the scalar-base + 32-bit-offset form, no 64-bit
        //  address arithmetic per pixel)
        const LinTabEntry ty = at_byte<LinTabEntry>(ytab, (uint32_t)y << 4), tx = at_byte<LinTabEntry>(xtab, (uint32_t)x << 4);
        static_assert(sizeof(LinTabEntry) == 16, "table entries are addressed by a shift");
        const uint32_t r0 = __umul24((uint32_t)ty.s0, (uint32_t)mesh_cols) << 3, r1 = __umul24((uint32_t)ty.s1, (uint32_t)mesh_cols) << 3;
        const uint32_t c0 = (uint32_t)tx.s0 << 3, c1 = (uint32_t)tx.s1 << 3;
        float off[2];
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
        {
            const uint32_t b = 4u * (uint32_t)ch;
            const float h0 = (tx.s1 == tx.s0) ? vertex(r0 + c0 + b) * 1.0f : vertex(r0 + c0 + b) * tx.a0 + vertex(r0 + c1 + b) * tx.a1;
            const float h1 = (tx.s1 == tx.s0) ? vertex(r1 + c0 + b) * 1.0f : vertex(r1 + c0 + b) * tx.a0 + vertex(r1 + c1 + b) * tx.a1;
            off[ch] = (h0 * ty.a0 + h1 * ty.a1) * (ch == 0 ? sw : sh);
        }
        subx = (float)x + off[0];
        suby = (float)y + off[1];
    }
};

struct MapCoord             // FSR.cl:376-381: a materialised offset map (pixels), e.g. the lens-correction warp
{
    const uint8_t* __restrict__ map; int map_step;
    __device__ __forceinline__ void operator()(int x, int y, float& subx, float& suby) const
    {
        const float2 o = *reinterpret_cast<const float2*>(map + (__umul24((uint32_t)y, (uint32_t)map_step) + 8u * (uint32_t)x));
        subx = (float)x + o.x;
        suby = (float)y + o.y;
    }
};

struct ScaleCoord           // easu_scale, FSR.cl:334-338: dst_coord * rscale (rscale = src size / dst size, Image.cpp:192-195)
{
    float rsx, rsy;
    __device__ __forceinline__ void operator()(int x, int y, float& subx, float& suby) const
    {
        subx = (float)x * rsx;
        suby = (float)y * rsy;
    }
};

// Fused lens pre-warp (BASELINE config 5): the inner functor yields the position (u, v) the
// stabilizing warp asks for in the LENS-CORRECTED frame; the closed-form Brown-Conrady map (LCFilter.cpp:133-171 reduced by
// lens.hip to 17 floats) carries it on to the raw frame, so the chain LC -> VS costs one EASU resampling and no map traffic.
// (u, v) outside the corrected frame is background, as the second pass of the reference chain would decide.
template <class Inner>
struct LensCoord
{
    Inner inner; LensArgs L; int rows, cols;
    __device__ __forceinline__ void operator()(int px, int py, float& subx, float& suby) const
    {
        float u, v;
        inner(px, py, u, v);
        const int ux = (int)u, vy = (int)v;
        if (ux < 0 || ux >= cols || vy < 0 || vy >= rows) { subx = -16.0f; suby = -16.0f; return; }
        const float* f = L.f;
        const float x = (u - f[2]) * f[0], y = (v - f[3]) * f[1];
        const float r2 = __builtin_fmaf(x, x, y * y);
        const float kr = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(f[12], r2, f[9]), r2, f[8]), r2, 1.0f);
        const float xy2 = (x + x) * y;
        const float xd = __builtin_fmaf(x, kr, __builtin_fmaf(f[10], xy2, f[11] * __builtin_fmaf(x + x, x, r2)));
        const float yd = __builtin_fmaf(y, kr, __builtin_fmaf(f[10], __builtin_fmaf(y + y, y, r2), f[11] * xy2));
        subx = __builtin_fmaf(f[4], xd, f[6]) + __builtin_fmaf(u, f[13], f[14]);
        suby = __builtin_fmaf(f[5], yd, f[7]) + __builtin_fmaf(v, f[15], f[16]);
    }
};

// ---- kernel body ---------------------------------------------------------------------------------------------------
// rocprofv3 (profiles/r06_remap_stalls.txt: the SQ_* / GRBM counters of THIS kernel, round 6) shows it is VALU-issue bound, not memory bound:
// 488 VALU instructions per output pixel, one leaving each SIMD every 2.94 cycles (0.68 of the 2-cycle issue slots; 0.85 of what its mix of
// 2- and 4-cycle opcodes allows), 2.4 of the 4.4 resident waves per SIMD ready and not issued at any moment (SQ_WAIT_INST_ANY 0.52-0.56 of the
// waves' time), 0.5 waiting for memory (SQ_WAIT_ANY 0.11-0.17).  Alternatives that were built and
// measured on MI355X at 4K and rejected: (a) staging the source window in LDS as pre-converted float4 (saves the
// 72 unpack + 24 luma ops per pixel but needs 4 block barriers and 31-36 KB LDS -> 4 waves/SIMD: 126 us), and
// (b) two pixels per lane on packed v_pk_fma/mul/add_f32 (those issue at half rate on gfx950, scripts/valu_peak.hip:
// 126 us), (c) hoisting the tap gathers of all 4 (or 2) pixels of a thread ahead of the arithmetic so that one memory round
// trip is exposed per thread instead of four (113 VGPRs, 4 waves/SIMD: 114-117 us; forcing 5-6 waves spills: 125-159 us).
// None of load batching, occupancy or LDS staging moves the time: the kernel runs at the rate its ~500 VALU
// instructions per pixel issue.  The barrier-free gather below keeps 5 waves/SIMD resident and runs at 108 us.
//
// A thread produces PXT horizontally adjacent output pixels so that the packed 3-byte pixels leave as three aligned
// dwords; a 256-thread block covers a 256 x 4 strip.  Strips are handed out so that the blocks one XCD receives
// (block b -> XCD b mod 8) form a contiguous band of the frame: vertically adjacent strips re-read 3 of their 7
// source rows, and with this order those re-reads hit that XCD's L2 instead of going back to HBM.
constexpr int PXT = 4, STRIP_W = 64 * PXT, STRIP_H = 4;
// Overlap mode: the remap shares the GPU with the next frame's tracker, whose small latency-bound kernels must be PLACED at once when
// they are launched.  The `_co` launches therefore run a persistent grid of LVK_CO_WAVES blocks per CU (one wave per SIMD each, every
// block walking several strips), which holds the remap at 4 waves per SIMD WITHOUT inflating its register allocation: the first
// version capped the occupancy with amdgpu_waves_per_eu(4, 4), which the compiler implements by padding the kernel to 104 VGPRs --
// 416 of a SIMD's 512 registers, so that the tracker's 1024-thread compaction block and its 250-VGPR RANSAC finalize block did not fit
// next to it and waited ~11 us each for remap workgroups to retire (in-kernel timeline, scripts/timeline_free.py).  With 72 VGPRs per
// remap wave 224 stay free on every SIMD.  The tracker kernels also raise their issue priority (s_setprio, LVK_TRACKER_PRIORITY)
// so that they are not starved by the VALU-bound remap waves they share a SIMD with.
#ifndef LVK_CO_WAVES
#define LVK_CO_WAVES 4
#endif
#define LVK_CO_SCHEDULED
// occupancy experiments (scripts/variant_build.sh waves8 -DLVK_REMAP_WAVES=8; round 6: 6 waves per SIMD = the default's time, 8 spill and
// are 8 % slower, profiles/r06_ab_remap_variants.txt): empty in the product build
#ifdef LVK_REMAP_WAVES
#define LVK_REMAP_ATTR __attribute__((amdgpu_waves_per_eu(LVK_REMAP_WAVES, LVK_REMAP_WAVES)))
#else
#define LVK_REMAP_ATTR
#endif
constexpr int NUM_XCD = 8;

// Output pixels are written once and not read again by this GPU for N frames: streaming (non-temporal) stores keep them from sitting
// dirty in the L2s that the tracker's kernels release at every kernel boundary (+1 % frames/s, -2 % latency next to the tracker)
#define LVK_STREAM_STORE(ptr, v) __builtin_nontemporal_store((uint32_t)(v), (ptr))
__device__ __forceinline__ void store_pixels(uint8_t* __restrict__ drow, int x0, int npx, const uint32_t px[PXT], bool aligned)
{
    if (npx == PXT && aligned)
    {
        uint32_t* d = reinterpret_cast<uint32_t*>(drow + 3 * x0);      // 4 packed pixels = 12 bytes = 3 dwords
        LVK_STREAM_STORE(d + 0, px[0] | (px[1] << 24));
        LVK_STREAM_STORE(d + 1, (px[1] >> 8) | (px[2] << 16));
        LVK_STREAM_STORE(d + 2, (px[2] >> 16) | (px[3] << 8));
    }
    else
        for (int p = 0; p < npx; p++)
        {
            uint8_t* d = drow + 3 * (x0 + p);
            d[0] = (uint8_t)px[p]; d[1] = (uint8_t)(px[p] >> 8); d[2] = (uint8_t)(px[p] >> 16);
        }
}

// Where a thread's PXT output pixels go.  PackedSink: the packed 8UC3 frame (three aligned dwords per thread).
struct PackedSink
{
    uint8_t* __restrict__ dst; int dst_step;
    __device__ __forceinline__ void store(int x0, int y, int npx, const uint32_t px[PXT], bool active, int /*parity*/) const
    {
        if (!active) return;
        // (32-bit row offset against the block-uniform base, like the tap loads: a frame is < 4 GB)
        uint8_t* drow = dst + __umul24((uint32_t)y, (uint32_t)dst_step);
        store_pixels(drow, x0, npx, px, ((reinterpret_cast<uintptr_t>(drow) & 3u) == 0));
    }
};

// Planar 4:2:0 sink (I420 or NV12): the remap and the OBS egress (split + cv::resize(0.5, INTER_AREA) = (a + b + c + d + 2) >> 2 on the
// chroma planes, FrameIngest.cpp:540-557,590-602) in one kernel.  Luma leaves as one dword per thread; the two rows of a chroma
// sample belong to threads of neighbouring waves of the strip (rows y, y + 1), which meet through 2 KB of LDS and one barrier.
template <bool NV12>
struct Sink420
{
    uint8_t* __restrict__ yp; int y_step; uint8_t* __restrict__ up; int u_step; uint8_t* __restrict__ vp; int v_step;
    __device__ __forceinline__ void store(int x0, int y, int npx, const uint32_t px[PXT], bool active, int parity) const
    {
        // two buffers, alternating per strip of a block: a wave that is already writing the next strip's sums cannot overwrite what a
        // slower wave of the block still has to read for this one (the one barrier per strip orders everything else)
        __shared__ uint2 s_uv2[2][256];
        uint2* s_uv = s_uv2[parity & 1];
        const int t = (int)threadIdx.x;
        // (U, V) of the four pixels, two 16-bit sums per word: u0 + u1 | u2 + u3 and v0 + v1 | v2 + v3 (horizontal pairs pre-added)
        const uint32_t u01 = ((px[0] >> 8) & 0xffu) + ((px[1] >> 8) & 0xffu), u23 = ((px[2] >> 8) & 0xffu) + ((px[3] >> 8) & 0xffu);
        const uint32_t v01 = ((px[0] >> 16) & 0xffu) + ((px[1] >> 16) & 0xffu), v23 = ((px[2] >> 16) & 0xffu) + ((px[3] >> 16) & 0xffu);
        s_uv[t] = make_uint2(u01 | (u23 << 16), v01 | (v23 << 16));
        if (active)
        {
            uint8_t* yr = yp + (__umul24((uint32_t)y, (uint32_t)y_step) + (uint32_t)x0);
            const uint32_t yy = (px[0] & 0xffu) | ((px[1] & 0xffu) << 8) | ((px[2] & 0xffu) << 16) | ((px[3] & 0xffu) << 24);
            if (npx == PXT && ((reinterpret_cast<uintptr_t>(yr) & 3u) == 0)) LVK_STREAM_STORE(reinterpret_cast<uint32_t*>(yr), yy);
            else for (int p = 0; p < npx; p++) yr[p] = (uint8_t)(yy >> (8 * p));
        }
        __syncthreads();
        if (active && ((t >> 6) & 1) == 0)                 // even row of the pair: partner = same lane, next wave
        {
            const uint2 a = s_uv[t], b = s_uv[t + 64];
            const uint32_t u0 = ((a.x & 0xffffu) + (b.x & 0xffffu) + 2u) >> 2, u1 = ((a.x >> 16) + (b.x >> 16) + 2u) >> 2;
            const uint32_t v0 = ((a.y & 0xffffu) + (b.y & 0xffffu) + 2u) >> 2, v1 = ((a.y >> 16) + (b.y >> 16) + 2u) >> 2;
            const int cx = x0 >> 1, cy = y >> 1, nc = npx >> 1;
            if (NV12)
            {
                uint8_t* d = up + (__umul24((uint32_t)cy, (uint32_t)u_step) + 2u * (uint32_t)cx);
                d[0] = (uint8_t)u0; d[1] = (uint8_t)v0;
                if (nc > 1) { d[2] = (uint8_t)u1; d[3] = (uint8_t)v1; }
            }
            else
            {
                uint8_t* du = up + (__umul24((uint32_t)cy, (uint32_t)u_step) + (uint32_t)cx); uint8_t* dv = vp + (__umul24((uint32_t)cy, (uint32_t)v_step) + (uint32_t)cx);
                du[0] = (uint8_t)u0; dv[0] = (uint8_t)v0;
                if (nc > 1) { du[1] = (uint8_t)u1; dv[1] = (uint8_t)v1; }
            }
        }
    }
};

template <bool YUV, class Coord, class Sink>
__device__ __forceinline__ void remap_one_strip(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                                                const Sink& sink, int dst_rows, int dst_cols, const Coord& coord, uint32_t bg,
                                                int strip, int nstrips, int strips_x, int parity)
{
    const int sy_ = strip / strips_x, sx_ = strip - sy_ * strips_x;
    const int x0 = sx_ * STRIP_W + (int)(threadIdx.x & 63) * PXT;
    const int y = sy_ * STRIP_H + (int)(threadIdx.x >> 6);
    const bool active = strip < nstrips && x0 < dst_cols && y < dst_rows;
    const int npx = active ? min(PXT, dst_cols - x0) : 0;
    const TapBases tb = tap_bases(src, src_step);
    uint32_t px[PXT];
#pragma unroll
    for (int p = 0; p < PXT; p++)
    {
        px[p] = 0;
        if (p < npx)
        {
            float subx, suby;
            coord(x0 + p, y, subx, suby);
            // shared tail of FSR.cl:380-402 / 429-451
            const int sx = (int)subx;                 // v_cvt_i32_f32: truncates, saturates, NaN -> 0
            const int sy = (int)suby;
            // `coord - floor(coord)` (FSR.cl:383,432) as ONE v_fract_f32: for every coordinate that reaches the EASU path (1 <= coord < size - 4) x and
            // floor(x) lie in the same or neighbouring binades and the difference is exact in either form, so the bits are the same; the border
            // and background paths do not use pp (where a tiny negative coordinate would round x - floor(x) up to 1.0 and v_fract stays below it)
            const float ppx = __builtin_amdgcn_fractf(subx);
            const float ppy = __builtin_amdgcn_fractf(suby);
            if (sx < 1 || sy < 1 || sx >= src_cols - 4 || sy >= src_rows - 4)
            {
                if (sx >= 0 && sx < src_cols && sy >= 0 && sy < src_rows)
                {
                    const uint8_t* s = src + (__umul24((uint32_t)sy, (uint32_t)src_step) + 3u * (uint32_t)sx);
                    px[p] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
                }
                else px[p] = bg;
            }
            else px[p] = easu_gather<YUV>(tb, src_step, sx, sy, ppx, ppy);
        }
    }
    sink.store(x0, y, npx, px, active, parity);
}

template <bool YUV, class Coord, class Sink>
__device__ __forceinline__ void remap_strip(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                                            const Sink& sink, int dst_rows, int dst_cols,
                                            const Coord& coord, uint32_t bg)
{
    const int strips_x = (dst_cols + STRIP_W - 1) / STRIP_W, strips_y = (dst_rows + STRIP_H - 1) / STRIP_H;
    const int nstrips = strips_x * strips_y;
    const int band = (nstrips + NUM_XCD - 1) / NUM_XCD;
    // A block walks the band of its XCD with the stride of the launch: one strip per block for a full grid, several for the persistent
    // grid of the overlap mode (lvk_co_grid).  Everything up to the sink is block-uniform.  (Drawing the strips dynamically from a
    // per-XCD counter instead -- L2-local atomics keyed by HW_REG_XCC_ID, one draw kept in flight -- was built and measured: bit-exact,
    // but 143 us instead of 103: the returning atomic sits in the same in-order vmcnt queue as the strip's first tap loads.)
    const int xcd = (int)(blockIdx.x % NUM_XCD);
    const int kstride = (int)(gridDim.x / NUM_XCD);
    int parity = 0;
    for (int k = (int)(blockIdx.x / NUM_XCD); k < band; k += kstride, parity ^= 1)
    {
        const int strip = xcd * band + k;
        if (strip >= nstrips) break;                                    // block-uniform (only the last band is short)
        remap_one_strip<YUV>(src, src_step, src_rows, src_cols, sink, dst_rows, dst_cols, coord, bg, strip, nstrips, strips_x, parity);
    }
}
inline dim3 remap_grid(int dst_rows, int dst_cols)
{
    const int nstrips = ((dst_cols + STRIP_W - 1) / STRIP_W) * ((dst_rows + STRIP_H - 1) / STRIP_H);
    return dim3((unsigned)(((nstrips + NUM_XCD - 1) / NUM_XCD) * NUM_XCD));
}

// Persistent grid of the overlap mode: LVK_CO_WAVES blocks (of 4 waves, one per SIMD) per CU, a multiple of the XCD count.
inline dim3 lvk_co_grid(lvk_hip_ctx* ctx, int dst_rows, int dst_cols)
{
    const int n = ctx->cu_count > 0 ? ctx->cu_count : 256;      // read once per context (no process-wide cache: filters run on many threads)
    const unsigned full = remap_grid(dst_rows, dst_cols).x;
    // co_blocks_per_cu: k > 0 = k blocks per CU, -k = one block per k CUs (a remap whose stores cross the host link does not need the chip)
    const int per_cu = ctx->co_blocks_per_cu != 0 ? ctx->co_blocks_per_cu : LVK_CO_WAVES;
    const unsigned persistent = (unsigned)std::max(NUM_XCD, ((per_cu > 0 ? n * per_cu : n / -per_cu) / NUM_XCD) * NUM_XCD);
    return dim3(full < persistent ? full : persistent);
}

// The kernels address a frame with ONE 32-bit byte offset per pixel built from 24-bit factors (easu_gather, the sinks): the whole frame
// must lie within 4 GB of its base and rows / pitch below 2^24 (an 8K packed frame is 100 MB with a 23 KB pitch).
inline bool fits_u32(int step, int rows) { return step > 0 && rows > 0 && step < (1 << 24) && rows < (1 << 24) && (uint64_t)step * (uint64_t)rows < (1ull << 32); }

inline uint32_t pack_bg(const uint8_t bg[3]) { return (uint32_t)bg[0] | ((uint32_t)bg[1] << 8) | ((uint32_t)bg[2] << 16); }

// cv::getPerspectiveTransform (OpenCV 4.8 imgproc; call site Math/WarpMesh.cpp:214): 8x8 double system,
// LU with partial pivoting.  Maps src[i] -> dst[i].
bool perspective_transform(const float src[8], const float dst[8], double M[9])
{
    double A[8][8], B[8];
    for (int i = 0; i < 4; i++)
    {
        const double x = src[2 * i], y = src[2 * i + 1], u = dst[2 * i], v = dst[2 * i + 1];
        A[i][0] = A[i + 4][3] = x;  A[i][1] = A[i + 4][4] = y;  A[i][2] = A[i + 4][5] = 1.0;
        A[i][3] = A[i][4] = A[i][5] = A[i + 4][0] = A[i + 4][1] = A[i + 4][2] = 0.0;
        A[i][6] = -x * u;  A[i][7] = -y * u;  A[i + 4][6] = -x * v;  A[i + 4][7] = -y * v;
        B[i] = u;  B[i + 4] = v;
    }
    for (int i = 0; i < 8; i++)
    {
        int piv = i;
        for (int j = i + 1; j < 8; j++) if (std::fabs(A[j][i]) > std::fabs(A[piv][i])) piv = j;
        if (std::fabs(A[piv][i]) < 2.220446049250313e-16 * 100) return false;
        if (piv != i) { for (int j = i; j < 8; j++) std::swap(A[i][j], A[piv][j]); std::swap(B[i], B[piv]); }
        const double d = -1.0 / A[i][i];
        for (int j = i + 1; j < 8; j++)
        {
            const double alpha = A[j][i] * d;
            for (int q = i + 1; q < 8; q++) A[j][q] += alpha * A[i][q];
            B[j] += alpha * B[i];
        }
    }
    for (int i = 7; i >= 0; i--)
    {
        double s = B[i];
        for (int q = i + 1; q < 8; q++) s -= A[i][q] * B[q];
        B[i] = s / A[i][i];
    }
    for (int q = 0; q < 8; q++) M[q] = B[q];
    M[8] = 1.0;
    return true;
}


} // namespace
