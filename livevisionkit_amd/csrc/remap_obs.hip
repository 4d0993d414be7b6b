// Remap + egress in one kernel for the OBS video formats that are not 4:2:0 (lvk_hip_stab_push_obs): the EASU remap of remap_core.hpp with a sink that
// writes the frame the way FrameIngest::to_obs would (reference: Modules/OBS-Plugin/Interop/FrameIngest.cpp:526-557 I4XXIngest, :640-666 P422Ingest,
// :690-703 P444Ingest) -- planar or packed 4:2:2 (chroma = cv::resize(0.5, 1.0, INTER_AREA) = the pixel pair's mean, round half to even), planar 4:4:4,
// AYUV.  Same bytes as lvk_launch_warpmesh_apply_lens followed by lvk_launch_egress_obs (tests/test_ingest_obs_gpu.py holds the two routes together);
// what it saves is the packed intermediate (2 x 3 W H bytes), one kernel and one launch per frame.
#include "remap_core.hpp"

namespace {

__device__ __forceinline__ uint32_t half_even_u32(uint32_t s) { return (s + ((s >> 1) & 1u)) >> 1; }      // cvRound(s * 0.5f)

// LAYOUT 0: planes Y, U, V (I422 / I42A); 1 YUY2 (Y U Y V); 2 YVYU; 3 UYVY.  A thread holds 4 horizontally adjacent pixels = 2 chroma pairs; the frame's
// width is even, so a thread at the right edge holds 2 or 4.
template <int LAYOUT>
struct Sink422
{
    uint8_t* __restrict__ p0; int s0; uint8_t* __restrict__ p1; int s1; uint8_t* __restrict__ p2; int s2;
    __device__ __forceinline__ void store(int x0, int y, int npx, const uint32_t px[PXT], bool active, int /*parity*/) const
    {
        if (!active) return;
        const uint32_t u0 = half_even_u32(((px[0] >> 8) & 0xffu) + ((px[1] >> 8) & 0xffu)), u1 = half_even_u32(((px[2] >> 8) & 0xffu) + ((px[3] >> 8) & 0xffu));
        const uint32_t v0 = half_even_u32(((px[0] >> 16) & 0xffu) + ((px[1] >> 16) & 0xffu)), v1 = half_even_u32(((px[2] >> 16) & 0xffu) + ((px[3] >> 16) & 0xffu));
        const uint32_t y0 = px[0] & 0xffu, y1 = px[1] & 0xffu, y2 = px[2] & 0xffu, y3 = px[3] & 0xffu;
        if (LAYOUT == 0)
        {
            uint8_t* yr = p0 + (__umul24((uint32_t)y, (uint32_t)s0) + (uint32_t)x0);
            uint8_t* ur = p1 + (__umul24((uint32_t)y, (uint32_t)s1) + (uint32_t)(x0 >> 1));
            uint8_t* vr = p2 + (__umul24((uint32_t)y, (uint32_t)s2) + (uint32_t)(x0 >> 1));
            const uint32_t yy = y0 | (y1 << 8) | (y2 << 16) | (y3 << 24);
            if (npx == PXT && ((reinterpret_cast<uintptr_t>(yr) & 3u) == 0)) LVK_STREAM_STORE(reinterpret_cast<uint32_t*>(yr), yy);
            else for (int p = 0; p < npx; p++) yr[p] = (uint8_t)(yy >> (8 * p));
            ur[0] = (uint8_t)u0; vr[0] = (uint8_t)v0;
            if (npx > 2) { ur[1] = (uint8_t)u1; vr[1] = (uint8_t)v1; }
        }
        else
        {
            const uint32_t f0 = LAYOUT != 2 ? u0 : v0, g0 = LAYOUT != 2 ? v0 : u0, f1 = LAYOUT != 2 ? u1 : v1, g1 = LAYOUT != 2 ? v1 : u1;
            const uint32_t a = LAYOUT == 3 ? (f0 | (y0 << 8) | (g0 << 16) | (y1 << 24)) : (y0 | (f0 << 8) | (y1 << 16) | (g0 << 24));
            const uint32_t b = LAYOUT == 3 ? (f1 | (y2 << 8) | (g1 << 16) | (y3 << 24)) : (y2 | (f1 << 8) | (y3 << 16) | (g1 << 24));
            uint8_t* d = p0 + (__umul24((uint32_t)y, (uint32_t)s0) + 2u * (uint32_t)x0);
            if ((reinterpret_cast<uintptr_t>(d) & 3u) == 0)
            {
                LVK_STREAM_STORE(reinterpret_cast<uint32_t*>(d), a);
                if (npx > 2) LVK_STREAM_STORE(reinterpret_cast<uint32_t*>(d) + 1, b);
            }
            else
                for (int k = 0; k < 2 * npx; k++) d[k] = (uint8_t)((k < 4 ? a : b) >> (8 * (k & 3)));
        }
    }
};

// LAYOUT 0: planes Y, U, V (I444 / YUVA); 1: A Y U V with A = 255 (AYUV)
template <int LAYOUT>
struct Sink444
{
    uint8_t* __restrict__ p0; int s0; uint8_t* __restrict__ p1; int s1; uint8_t* __restrict__ p2; int s2;
    __device__ __forceinline__ void store(int x0, int y, int npx, const uint32_t px[PXT], bool active, int /*parity*/) const
    {
        if (!active) return;
        if (LAYOUT == 0)
        {
            uint8_t* r[3] = {p0 + (__umul24((uint32_t)y, (uint32_t)s0) + (uint32_t)x0), p1 + (__umul24((uint32_t)y, (uint32_t)s1) + (uint32_t)x0),
                             p2 + (__umul24((uint32_t)y, (uint32_t)s2) + (uint32_t)x0)};
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                const uint32_t w = ((px[0] >> (8 * ch)) & 0xffu) | (((px[1] >> (8 * ch)) & 0xffu) << 8) | (((px[2] >> (8 * ch)) & 0xffu) << 16) | (((px[3] >> (8 * ch)) & 0xffu) << 24);
                if (npx == PXT && ((reinterpret_cast<uintptr_t>(r[ch]) & 3u) == 0)) LVK_STREAM_STORE(reinterpret_cast<uint32_t*>(r[ch]), w);
                else for (int p = 0; p < npx; p++) r[ch][p] = (uint8_t)(w >> (8 * p));
            }
        }
        else
        {
            uint8_t* d = p0 + (__umul24((uint32_t)y, (uint32_t)s0) + 4u * (uint32_t)x0);
            const bool al = (reinterpret_cast<uintptr_t>(d) & 3u) == 0;
            for (int p = 0; p < npx; p++)
            {
                const uint32_t w = 255u | (px[p] << 8);
                if (al) LVK_STREAM_STORE(reinterpret_cast<uint32_t*>(d) + p, w);
                else { d[4 * p] = 255; d[4 * p + 1] = (uint8_t)px[p]; d[4 * p + 2] = (uint8_t)(px[p] >> 8); d[4 * p + 3] = (uint8_t)(px[p] >> 16); }
            }
        }
    }
};

template <class Sink>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_homography_planes(const uint8_t* __restrict__ src, int src_step, int rows, int cols, Sink sink, HomographyArgs H, uint32_t bg)
{
    remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, HomographyCoord{H, 0, 0}, bg);
}

template <class Sink>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_homography_lens_planes(const uint8_t* __restrict__ src, int src_step, int rows, int cols, Sink sink, HomographyArgs H, LensArgs L, uint32_t bg)
{
    remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, LensCoord<HomographyCoord>{HomographyCoord{H, 0, 0}, L, rows, cols}, bg);
}

template <class Sink>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_mesh_planes(const uint8_t* __restrict__ src, int src_step, int rows, int cols, Sink sink,
                         const float* __restrict__ mesh, int mesh_cols, int mesh_floats, const LinTabEntry* __restrict__ xtab, const LinTabEntry* __restrict__ ytab, uint32_t bg)
{
    if (mesh_to_lds(mesh, mesh_floats)) remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, MeshCoordT<true>{mesh, mesh_cols, xtab, ytab, (float)cols, (float)rows}, bg);
    else remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, MeshCoordT<false>{mesh, mesh_cols, xtab, ytab, (float)cols, (float)rows}, bg);
}

template <class Sink>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_mesh_lens_planes(const uint8_t* __restrict__ src, int src_step, int rows, int cols, Sink sink,
                              const float* __restrict__ mesh, int mesh_cols, int mesh_floats, const LinTabEntry* __restrict__ xtab, const LinTabEntry* __restrict__ ytab,
                              LensArgs L, uint32_t bg)
{
    if (mesh_to_lds(mesh, mesh_floats))
        remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, LensCoord<MeshCoordT<true>>{MeshCoordT<true>{mesh, mesh_cols, xtab, ytab, (float)cols, (float)rows}, L, rows, cols}, bg);
    else
        remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, LensCoord<MeshCoordT<false>>{MeshCoordT<false>{mesh, mesh_cols, xtab, ytab, (float)cols, (float)rows}, L, rows, cols}, bg);
}

template <class Sink>
int launch_planes(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols, const Sink& sink,
                  const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], const LensArgs* lens, bool co)
{
    const dim3 block(256), grid = co ? lvk_co_grid(ctx, rows, cols) : remap_grid(rows, cols);
    int stage_slot = -1;
    if (mesh_rows == 2 && mesh_cols == 2)
    {
        const float w = (float)cols, h = (float)rows;                 // WarpMesh.cpp:194-217, as in lvk_launch_warpmesh_apply_lens
        const float dstp[8] = { 0, 0, w, 0, 0, h, w, h };
        float srcp[8];
        for (int i = 0; i < 4; i++)
        {
            srcp[2 * i] = dstp[2 * i] + (float)((double)mesh[2 * i] * (double)cols);
            srcp[2 * i + 1] = dstp[2 * i + 1] + (float)((double)mesh[2 * i + 1] * (double)rows);
        }
        double M[9];
        if (!perspective_transform(dstp, srcp, M))
            for (int q = 0; q < 9; q++) M[q] = (q % 4 == 0) ? 1.0 : 0.0;
        HomographyArgs args;
        for (int q = 0; q < 9; q++) args.h[q] = (float)M[q];
        if (lens) hipLaunchKernelGGL(k_remap_homography_lens_planes<Sink>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, sink, args, *lens, pack_bg(bg));
        else hipLaunchKernelGGL(k_remap_homography_planes<Sink>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, sink, args, pack_bg(bg));
    }
    else
    {
        const size_t mesh_bytes = (size_t)mesh_rows * mesh_cols * 2 * sizeof(float);
        LVK_HIP_REQUIRE(ctx, mesh_bytes <= lvk_hip_ctx::kStageBytes);
        const LinTabEntry *xtab = nullptr, *ytab = nullptr;          // before the mesh is staged (see lvk_launch_remap_mesh)
        int rc;
        if ((rc = lvk_get_lintab(ctx, mesh_cols, cols, false, &xtab)) != LVK_HIP_OK) return rc;
        if ((rc = lvk_get_lintab(ctx, mesh_rows, rows, true, &ytab)) != LVK_HIP_OK) return rc;
        void* d_mesh = nullptr;
        if ((rc = lvk_stage_params(ctx, stream, mesh, mesh_bytes, &d_mesh, &stage_slot)) != LVK_HIP_OK) return rc;
        if (lens) hipLaunchKernelGGL(k_remap_mesh_lens_planes<Sink>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, sink, (const float*)d_mesh, mesh_cols,
                                     mesh_rows * mesh_cols * 2, xtab, ytab, *lens, pack_bg(bg));
        else hipLaunchKernelGGL(k_remap_mesh_planes<Sink>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, sink, (const float*)d_mesh, mesh_cols,
                                mesh_rows * mesh_cols * 2, xtab, ytab, pack_bg(bg));
    }
    const hipError_t le = hipGetLastError();
    const int src = stage_slot >= 0 ? lvk_stage_consumed(ctx, stage_slot, stream) : LVK_HIP_OK;
    if (le != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(le));
    return src;
}

} // namespace

bool lvk_remap_obs_fusable(int video_format)
{
    switch (video_format)
    {
    case LVK_VIDEO_FORMAT_I422: case LVK_VIDEO_FORMAT_I42A: case LVK_VIDEO_FORMAT_YUY2: case LVK_VIDEO_FORMAT_YVYU: case LVK_VIDEO_FORMAT_UYVY:
    case LVK_VIDEO_FORMAT_I444: case LVK_VIDEO_FORMAT_YUVA: case LVK_VIDEO_FORMAT_AYUV: return true;
    default: return false;
    }
}

// WarpMesh::apply + FrameIngest::to_obs of `video_format` in one kernel; the planes' geometry has been checked by the caller (lvk_stab_push_planes).
int lvk_launch_warpmesh_apply_obs(lvk_hip_ctx* ctx, hipStream_t stream, int video_format, const void* d_src, int src_step, int rows, int cols,
                                  void* const planes[3], const int steps[3], const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3],
                                  const LensArgs* lens, bool co)
{
    LVK_HIP_REQUIRE(ctx, d_src && planes && steps && planes[0] && mesh && bg && mesh_rows >= 2 && mesh_cols >= 2 && rows > 0 && cols > 0 && src_step >= 3 * cols);
    LVK_HIP_REQUIRE(ctx, fits_u32(src_step, rows) && fits_u32(steps[0], rows));
    uint8_t* p0 = (uint8_t*)planes[0]; uint8_t* p1 = (uint8_t*)planes[1]; uint8_t* p2 = (uint8_t*)planes[2];
    const bool planar = video_format == LVK_VIDEO_FORMAT_I422 || video_format == LVK_VIDEO_FORMAT_I42A || video_format == LVK_VIDEO_FORMAT_I444 || video_format == LVK_VIDEO_FORMAT_YUVA;
    if (planar) LVK_HIP_REQUIRE(ctx, p1 && p2 && fits_u32(steps[1], rows) && fits_u32(steps[2], rows));
    switch (video_format)
    {
    case LVK_VIDEO_FORMAT_I422: case LVK_VIDEO_FORMAT_I42A:
        LVK_HIP_REQUIRE(ctx, (cols & 1) == 0 && steps[0] >= cols && steps[1] >= cols / 2 && steps[2] >= cols / 2);
        return launch_planes(ctx, stream, d_src, src_step, rows, cols, Sink422<0>{p0, steps[0], p1, steps[1], p2, steps[2]}, mesh, mesh_rows, mesh_cols, bg, lens, co);
    case LVK_VIDEO_FORMAT_YUY2:
        LVK_HIP_REQUIRE(ctx, (cols & 1) == 0 && steps[0] >= 2 * cols);
        return launch_planes(ctx, stream, d_src, src_step, rows, cols, Sink422<1>{p0, steps[0], p0, 0, p0, 0}, mesh, mesh_rows, mesh_cols, bg, lens, co);
    case LVK_VIDEO_FORMAT_YVYU:
        LVK_HIP_REQUIRE(ctx, (cols & 1) == 0 && steps[0] >= 2 * cols);
        return launch_planes(ctx, stream, d_src, src_step, rows, cols, Sink422<2>{p0, steps[0], p0, 0, p0, 0}, mesh, mesh_rows, mesh_cols, bg, lens, co);
    case LVK_VIDEO_FORMAT_UYVY:
        LVK_HIP_REQUIRE(ctx, (cols & 1) == 0 && steps[0] >= 2 * cols);
        return launch_planes(ctx, stream, d_src, src_step, rows, cols, Sink422<3>{p0, steps[0], p0, 0, p0, 0}, mesh, mesh_rows, mesh_cols, bg, lens, co);
    case LVK_VIDEO_FORMAT_I444: case LVK_VIDEO_FORMAT_YUVA:
        LVK_HIP_REQUIRE(ctx, steps[0] >= cols && steps[1] >= cols && steps[2] >= cols);
        return launch_planes(ctx, stream, d_src, src_step, rows, cols, Sink444<0>{p0, steps[0], p1, steps[1], p2, steps[2]}, mesh, mesh_rows, mesh_cols, bg, lens, co);
    case LVK_VIDEO_FORMAT_AYUV:
        LVK_HIP_REQUIRE(ctx, steps[0] >= 4 * cols);
        return launch_planes(ctx, stream, d_src, src_step, rows, cols, Sink444<1>{p0, steps[0], p0, 0, p0, 0}, mesh, mesh_rows, mesh_cols, bg, lens, co);
    }
    return ctx->fail(LVK_HIP_ERR_ARG, "lvk_launch_warpmesh_apply_obs: no fused sink for video format " + std::to_string(video_format));
}
