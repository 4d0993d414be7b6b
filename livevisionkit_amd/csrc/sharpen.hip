// FSR robust contrast adaptive sharpening (RCAS) for gfx950.
//
// Replaces lvk::sharpen (reference: LiveVisionKit/Functions/Image.cpp:206-233) and the OpenCL kernel `rcas`
// (LiveVisionKit/Functions/OpenCL/Sources/FSR.cl:460-535), the second half of ScalingFilter::filter
// (LiveVisionKit/Filters/ScalingFilter.cpp:52-59).
//
// Arithmetic contract (must stay in lock-step with the specification the tests check against): binary32 IEEE ops, no implicit contraction, fused
// multiply-adds exactly where written as fma(), native_recip = the correctly rounded 1.0f / x, min/max with fmin/fmax semantics
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

constexpr int RCAS_PXT = 4, RCAS_ROWS = 4, RCAS_STRIP_W = 64 * RCAS_PXT, RCAS_STRIP_H = 4 * RCAS_ROWS;

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

// One source row as a thread sees it: the pixel left of its four, its four, and the pixel right of them (raw bytes kept for the
// border copy).
struct Row
{
    Px p[RCAS_PXT + 2];
    uint32_t raw[3];
};

__device__ __forceinline__ void load_row(Row& r, const uint8_t* __restrict__ src, int step, int y, int x0, int cols, bool full)
{
    const uint8_t* rp = src + (long)y * step + 3 * (long)x0;
    uint32_t w0, w1, w2, wl = 0u, wr = 0u;
    if (full)
    {
        const U12B v = *reinterpret_cast<const U12B*>(rp);
        w0 = v.w[0]; w1 = v.w[1]; w2 = v.w[2];
    }
    else
    {
        uint8_t b[12];
#pragma unroll
        for (int i = 0; i < 12; i++) b[i] = (x0 + i / 3 < cols) ? rp[i] : (uint8_t)0;
        w0 = b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24);
        w1 = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
        w2 = b[8] | (b[9] << 8) | (b[10] << 16) | ((uint32_t)b[11] << 24);
    }
    if (x0 > 0) wl = reinterpret_cast<const U4B*>(rp - 4)->w >> 8;                       // bytes -3..-1
    if (x0 + RCAS_PXT < cols) wr = (uint32_t)rp[12] | ((uint32_t)rp[13] << 8) | ((uint32_t)rp[14] << 16);
    r.raw[0] = w0; r.raw[1] = w1; r.raw[2] = w2;
    r.p[0] = unpack(wl);
    r.p[1] = unpack(w0);
    r.p[2] = unpack((w0 >> 24) | (w1 << 8));
    r.p[3] = unpack((w1 >> 16) | (w2 << 16));
    r.p[4] = unpack(w2 >> 8);
    r.p[5] = unpack(wr);
}

__global__ __launch_bounds__(256)
void k_rcas(const uint8_t* __restrict__ src, int src_step, int rows, int cols, uint8_t* __restrict__ dst, int dst_step, float sharp)
{
    // limiter reciprocals of the 256 possible ring extrema (FSR.cl:513-518, "these need to be high precision RCPs")
    __shared__ float s_rmin[256], s_rmax[256];
    {
        const float v = (float)threadIdx.x * 0.00392156862f;
        s_rmin[threadIdx.x] = 1.0f / (4.0f * v);                    // native_recip(4 * mx4)
        s_rmax[threadIdx.x] = 1.0f / fma_(4.0f, v, -4.0f);           // native_recip(4 * mn4 + peakC.y)
    }
    __syncthreads();

    const int x0 = (int)blockIdx.x * RCAS_STRIP_W + (int)(threadIdx.x & 63) * RCAS_PXT;
    const int y0 = (int)blockIdx.y * RCAS_STRIP_H + (int)(threadIdx.x >> 6) * RCAS_ROWS;
    if (x0 >= cols || y0 >= rows) return;
    const int npx = min(RCAS_PXT, cols - x0);
    const bool full = npx == RCAS_PXT;

    Row win[3] = {};                                                 // rows y - 1, y, y + 1 (rows outside the image stay 0: unused)
    if (y0 > 0) load_row(win[0], src, src_step, y0 - 1, x0, cols, full);
    load_row(win[1], src, src_step, y0, x0, cols, full);
#pragma unroll
    for (int r = 0; r < RCAS_ROWS; r++)
    {
        const int y = y0 + r;
        if (y >= rows) break;
        Row& up = win[r % 3];
        Row& mid = win[(r + 1) % 3];
        Row& down = win[(r + 2) % 3];
        if (y + 1 < rows) load_row(down, src, src_step, y + 1, x0, cols, full);
        uint32_t out[RCAS_PXT];
        const bool border_row = y == 0 || y >= rows - 1;
#pragma unroll
        for (int p = 0; p < RCAS_PXT; p++)
        {
            const int x = x0 + p;
            // raw bytes of pixel p of the middle row
            const uint32_t raw = p == 0 ? (mid.raw[0] & 0xffffffu)
                               : p == 1 ? ((mid.raw[0] >> 24) | ((mid.raw[1] & 0xffffu) << 8))
                               : p == 2 ? ((mid.raw[1] >> 16) | ((mid.raw[2] & 0xffu) << 16))
                               : (mid.raw[2] >> 8);
            const bool border = border_row || x == 0 || x >= cols - 1;              // FSR.cl:475-481: copied (selected below)
            const Px &b = up.p[p + 1], &h = down.p[p + 1], &d = mid.p[p], &e = mid.p[p + 1], &f = mid.p[p + 2];
            float lobe_c[3];
#pragma unroll
            for (int c = 0; c < 3; c++)                              // FSR.cl:503-521
            {
                const float mn4 = min_(b.c[c], min_(d.c[c], min_(f.c[c], h.c[c])));
                const float mx4 = max_(b.c[c], max_(d.c[c], max_(f.c[c], h.c[c])));
                // k of an extremum k * norm_factor: k * norm * 255 = k (1 - 2e-9), + 0.5 truncates to k for every k in 0..255
                const int kmx = (int)fma_(mx4, 255.0f, 0.5f), kmn = (int)fma_(mn4, 255.0f, 0.5f);
                const float hitMin = min_(mn4, e.c[c]) * s_rmin[kmx];
                const float hitMax = (1.0f - max_(mx4, e.c[c])) * s_rmax[kmn];
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
            out[p] = border ? raw : px;
        }
        uint8_t* drow = dst + (long)y * dst_step + 3 * (long)x0;
        if (full && ((reinterpret_cast<uintptr_t>(drow) & 3u) == 0))
        {
            uint32_t* o = reinterpret_cast<uint32_t*>(drow);
            o[0] = out[0] | (out[1] << 24);
            o[1] = (out[1] >> 8) | (out[2] << 16);
            o[2] = (out[2] >> 16) | (out[3] << 8);
        }
        else
            for (int p = 0; p < npx; p++) { drow[3 * p] = (uint8_t)out[p]; drow[3 * p + 1] = (uint8_t)(out[p] >> 8); drow[3 * p + 2] = (uint8_t)(out[p] >> 16); }
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
    const dim3 block(256), grid((unsigned)((cols + RCAS_STRIP_W - 1) / RCAS_STRIP_W), (unsigned)((rows + RCAS_STRIP_H - 1) / RCAS_STRIP_H));
    hipLaunchKernelGGL(k_rcas, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, (uint8_t*)d_dst, dst_step, sharp);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

extern "C" int lvk_hip_sharpen(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step, float sharpness)
{
    if (!ctx) return LVK_HIP_ERR_ARG;
    return lvk_launch_sharpen(ctx, ctx->stream, d_src, src_step, rows, cols, d_dst, dst_step, sharpness);
}
