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

} // extern "C"

LVK_TL_EXPORT(ingest)
