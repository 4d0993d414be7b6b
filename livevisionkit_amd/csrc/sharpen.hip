// FSR robust contrast adaptive sharpening (RCAS) for gfx950.
//
// Replaces lvk::sharpen (reference: LiveVisionKit/Functions/Image.cpp:206-233) and the OpenCL kernel `rcas`
// (LiveVisionKit/Functions/OpenCL/Sources/FSR.cl:460-535), the second half of ScalingFilter::filter
// (LiveVisionKit/Filters/ScalingFilter.cpp:52-59).
//
// Arithmetic contract (must stay in lock-step with the specification the tests check against): binary32 IEEE ops, no implicit contraction, fused
// multiply-adds exactly where written as fma(), native_recip = v_rcp_f32 as in the reference's compiled kernel (see k_rcas), min/max with fmin/fmax semantics
// (a NaN operand loses), truncating byte conversion.  Out of place; border pixels are copied.
//
// Shape: 6 B of traffic and ~90 VALU instructions per pixel put this kernel near the point where the HBM and VALU rooflines
// meet (4K: 49.8 MB = 6.2 us at 8 TB/s, ~9 us of VALU issue).  A thread owns a 4 x RCAS_ROWS block of output pixels (three aligned
// dwords per row), walks it top to bottom with a rolling 3-row window in registers, and takes the six limiter reciprocals per pixel
// from two 256-entry LDS tables: the ring minimum / maximum is one of the 256 values k * norm_factor, so `1 / (4 max)` and
// `1 / (4 min - 4)` have 256 possible correctly rounded results each, computed once per block with the IEEE division.
#include "lvk_hip_internal.hpp"

#include <cmath>

namespace {

constexpr int RCAS_PXT = 4, RCAS_STRIP_W = 64 * RCAS_PXT;

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float min_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float max_(float a, float b) { return __builtin_fmaxf(a, b); }

struct __attribute__((packed, aligned(1))) U12B { uint32_t w[3]; };
struct __attribute__((packed, aligned(1))) U4B { uint32_t w; };

struct Px { float c[3]; };

__device__ __forceinline__ Px unpack(uint32_t lo_bytes)
{
    const float norm_factor = 0.00392156862f;               // FSR.cl:484
    Px r;
    r.c[0] = (float)(lo_bytes & 0xffu) * norm_factor;
    r.c[1] = (float)((lo_bytes >> 8) & 0xffu) * norm_factor;
    r.c[2] = (float)((lo_bytes >> 16) & 0xffu) * norm_factor;
    return r;
}

// One source row as a thread sees it: the pixel left of its four, its four, and the pixel right of them.
struct RawRow { uint32_t w0, w1, w2, wl, wr; };
struct Row { Px p[RCAS_PXT + 2]; };

// Unconditional loads (no branches, so that all rows of a thread are in flight together): the row index is clamped into the image
// and the two side loads fall back to in-row addresses at the frame edge -- whatever they return there belongs to border pixels,
// which are copied.  Requires x0 + 4 <= cols.
__device__ __forceinline__ RawRow load_row(const uint8_t* __restrict__ src, int step, int y, int rows, int x0, int cols)
{
    const uint8_t* rp = src + (long)min(max(y, 0), rows - 1) * step + 3 * (long)x0;
    const U12B v = *reinterpret_cast<const U12B*>(rp);
    RawRow r;
    r.w0 = v.w[0]; r.w1 = v.w[1]; r.w2 = v.w[2];
    r.wl = reinterpret_cast<const U4B*>(x0 > 0 ? rp - 4 : rp)->w >> 8;                        // bytes -3..-1
    r.wr = reinterpret_cast<const U4B*>(x0 + RCAS_PXT < cols ? rp + 11 : rp + 8)->w >> 8;     // bytes 12..14
    return r;
}

__device__ __forceinline__ Row unpack_row(const RawRow& r)
{
    Row o;
    o.p[0] = unpack(r.wl);
    o.p[1] = unpack(r.w0);
    o.p[2] = unpack((r.w0 >> 24) | (r.w1 << 8));
    o.p[3] = unpack((r.w1 >> 16) | (r.w2 << 16));
    o.p[4] = unpack(r.w2 >> 8);
    o.p[5] = unpack(r.wr);
    return o;
}

// FSR.cl:486-534 for one pixel: b above, h below, d left, f right, e centre.  Returns 0x00ZZYYXX.
__device__ __forceinline__ uint32_t rcas_pixel(const Px& b, const Px& d, const Px& e, const Px& f, const Px& h, float sharp,
                                               const float* __restrict__ s_rmin, const float* __restrict__ s_rmax)
{
    float lobe_c[3];
#pragma unroll
    for (int c = 0; c < 3; c++)                              // FSR.cl:503-521
    {
        const float mn4 = min_(b.c[c], min_(d.c[c], min_(f.c[c], h.c[c])));
        const float mx4 = max_(b.c[c], max_(d.c[c], max_(f.c[c], h.c[c])));
        // Byte offset 4 k of an extremum k * norm_factor in the tables, without a conversion or a left shift (both half-rate VALU
        // classes on gfx950, scripts/valu_peak.hip): 1020 (k norm) + 2 = 4 k + 2 - 8e-9 k, and adding 2^23 leaves that integer,
        // rounded to 4 k + 2 for every k in 0..255, in the low mantissa bits.
        const uint32_t omx = __float_as_uint(fma_(mx4, 1020.0f, 8388610.0f)) & 0x3fcu;
        const uint32_t omn = __float_as_uint(fma_(mn4, 1020.0f, 8388610.0f)) & 0x3fcu;
        const float hitMin = min_(mn4, e.c[c]) * *reinterpret_cast<const float*>(reinterpret_cast<const char*>(s_rmin) + omx);
        const float hitMax = (1.0f - max_(mx4, e.c[c])) * *reinterpret_cast<const float*>(reinterpret_cast<const char*>(s_rmax) + omn);
        lobe_c[c] = max_(-hitMin, hitMax);
    }
    const float lobe = min_(max_(max_(lobe_c[2], max_(lobe_c[1], lobe_c[0])), -0.1875f), 0.0f) * sharp;   // FSR.cl:525
    const float a = fma_(4.0f, lobe, 1.0f);                  // FSR.cl:528, APrxMedRcpF1 (FSR.cl:70)
    const float rb = __uint_as_float(0x7ef19fffu - __float_as_uint(a));
    const float rcpL = rb * fma_(-rb, a, 2.0f);
    uint32_t px = 0;
#pragma unroll
    for (int c = 0; c < 3; c++)                              // FSR.cl:529-531
    {
        const float v = fma_(((b.c[c] + d.c[c]) + h.c[c]) + f.c[c], lobe, e.c[c]) * rcpL;
        px |= ((uint32_t)(int)(v * 255.0f) & 0xffu) << (8 * c);
    }
    return px;
}

__device__ __forceinline__ uint32_t load_px_raw(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }

template <int ROWS>
__global__ __launch_bounds__(256)
void k_rcas(const uint8_t* __restrict__ src, int src_step, int rows, int cols, uint8_t* __restrict__ dst, int dst_step, float sharp)
{
    // limiter reciprocals of the 256 possible ring extrema (FSR.cl:513-518, "these need to be high precision RCPs")
    __shared__ float s_rmin[256], s_rmax[256];
    {
        const float v = (float)threadIdx.x * 0.00392156862f;
        // What the reference's kernel computes when it is compiled for this device: LLVM folds `-hitMin` into
        // min * (-1.0f / (4 mx4)), a correctly rounded divide; the hitMax reciprocal stays the device's v_rcp_f32.
        s_rmin[threadIdx.x] = 1.0f / (4.0f * v);                                   // native_recip(4 * mx4), negation folded in
        s_rmax[threadIdx.x] = __builtin_amdgcn_rcpf(fma_(4.0f, v, -4.0f));         // native_recip(4 * mn4 + peakC.y)
    }
    __syncthreads();

    const int x0 = (int)blockIdx.x * RCAS_STRIP_W + (int)(threadIdx.x & 63) * RCAS_PXT;
    const int y0 = ((int)blockIdx.y * 4 + (int)(threadIdx.x >> 6)) * ROWS;
    if (x0 >= cols || y0 >= rows) return;

    if (x0 + RCAS_PXT > cols)                                        // ragged right edge (cols % 4 != 0): one pixel at a time
    {
        for (int y = y0; y < min(y0 + ROWS, rows); y++)
            for (int x = x0; x < cols; x++)
            {
                const uint8_t* pe = src + (long)y * src_step + 3 * (long)x;
                uint32_t px = load_px_raw(pe);
                if (!(x == 0 || x >= cols - 1 || y == 0 || y >= rows - 1))
                    px = rcas_pixel(unpack(load_px_raw(pe - src_step)), unpack(load_px_raw(pe - 3)), unpack(px), unpack(load_px_raw(pe + 3)),
                                    unpack(load_px_raw(pe + src_step)), sharp, s_rmin, s_rmax);
                uint8_t* o = dst + (long)y * dst_step + 3 * (long)x;
                o[0] = (uint8_t)px; o[1] = (uint8_t)(px >> 8); o[2] = (uint8_t)(px >> 16);
            }
        return;
    }

    RawRow raw[ROWS + 2];                                            // rows y0 - 1 .. y0 + ROWS, all in flight together
#pragma unroll
    for (int r = 0; r < ROWS + 2; r++) raw[r] = load_row(src, src_step, y0 - 1 + r, rows, x0, cols);
    Row win[3];
    win[0] = unpack_row(raw[0]);
    win[1] = unpack_row(raw[1]);
#pragma unroll
    for (int r = 0; r < ROWS; r++)
    {
        const int y = y0 + r;
        if (y >= rows) break;
        const Row& up = win[r % 3];
        const Row& mid = win[(r + 1) % 3];
        Row& down = win[(r + 2) % 3];
        down = unpack_row(raw[r + 2]);
        const RawRow& m = raw[r + 1];
        const uint32_t centre[RCAS_PXT] = { m.w0 & 0xffffffu, (m.w0 >> 24) | ((m.w1 & 0xffffu) << 8), (m.w1 >> 16) | ((m.w2 & 0xffu) << 16), m.w2 >> 8 };
        const bool border_row = y == 0 || y >= rows - 1;
        uint32_t out[RCAS_PXT];
#pragma unroll
        for (int p = 0; p < RCAS_PXT; p++)
        {
            const int x = x0 + p;
            const bool border = border_row || x == 0 || x >= cols - 1;              // FSR.cl:475-481: copied
            const uint32_t px = rcas_pixel(up.p[p + 1], mid.p[p], mid.p[p + 1], mid.p[p + 2], down.p[p + 1], sharp, s_rmin, s_rmax);
            out[p] = border ? centre[p] : px;
        }
        uint8_t* drow = dst + (long)y * dst_step + 3 * (long)x0;
        if ((reinterpret_cast<uintptr_t>(drow) & 3u) == 0)
        {
            uint32_t* o = reinterpret_cast<uint32_t*>(drow);
            o[0] = out[0] | (out[1] << 24);
            o[1] = (out[1] >> 8) | (out[2] << 16);
            o[2] = (out[2] >> 16) | (out[3] << 8);
        }
        else
            for (int p = 0; p < RCAS_PXT; p++) { drow[3 * p] = (uint8_t)out[p]; drow[3 * p + 1] = (uint8_t)(out[p] >> 8); drow[3 * p + 2] = (uint8_t)(out[p] >> 16); }
    }
}

} // namespace

// lvk::sharpen(src, dst, sharpness) (Functions/Image.cpp:206-233)
int lvk_launch_sharpen(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                       void* d_dst, int dst_step, float sharpness)
{
    LVK_HIP_REQUIRE(ctx, d_src != nullptr && d_dst != nullptr && d_src != d_dst);
    LVK_HIP_REQUIRE(ctx, cols > 0 && rows > 0);                                  // Image.cpp:208
    LVK_HIP_REQUIRE(ctx, sharpness >= 0.0f && sharpness <= 1.0f);                // LVK_ASSERT_01, Image.cpp:210
    LVK_HIP_REQUIRE(ctx, src_step >= 3 * cols && dst_step >= 3 * cols);
    const float sharp = exp2f(-2.0f * (1.0f - sharpness));                       // Image.cpp:227
    constexpr int ROWS = 4;                                                      // rows per thread: 1 / 2 / 4 / 8 measure 24.3 / 23.0 / 23.3 / 25.7 us at 4K
    const dim3 block(256), grid((unsigned)((cols + RCAS_STRIP_W - 1) / RCAS_STRIP_W), (unsigned)((rows + 4 * ROWS - 1) / (4 * ROWS)));
    hipLaunchKernelGGL(k_rcas<ROWS>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, (uint8_t*)d_dst, dst_step, sharp);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

extern "C" int lvk_hip_sharpen(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step, float sharpness)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_sharpen(ctx, ctx->stream, d_src, src_step, rows, cols, d_dst, dst_step, sharpness);
}

// native_recip of FSR.cl on this device (lvk_hip.h): what rcp_native() / the table of k_rcas evaluate.
namespace {
__global__ __launch_bounds__(256) void k_native_rcp(const float* __restrict__ in, float* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = __builtin_amdgcn_rcpf(in[i]);
}
}

extern "C" int lvk_hip_native_rcp(lvk_hip_ctx* ctx, const float* d_in, float* d_out, size_t n)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, d_in != nullptr && d_out != nullptr);
    if (n == 0) return LVK_HIP_OK;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_native_rcp, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, ctx->stream, d_in, d_out, n);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}
