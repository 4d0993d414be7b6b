// Dense frame remap for gfx950, kernels and launchers: lvk::remap x2 (LiveVisionKit/Functions/Image.cpp:28-151), WarpMesh::apply
// (Math/WarpMesh.cpp:183-223), lvk::upscale (Image.cpp:155-202) and the fused remap + 4:2:0 egress of the plugin's path.  The EASU arithmetic, the
// coordinate generators, the sinks and the strip walk are in remap_core.hpp.
#include "remap_core.hpp"
#ifndef LVK_CO_LDS_PAD
#define LVK_CO_LDS_PAD 0
#endif

namespace {


template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR
void k_remap_homography(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                        uint8_t* __restrict__ dst, int dst_step, int dst_rows, int dst_cols,
                        int off_x, int off_y, HomographyArgs H, uint32_t bg)
{
    const HomographyCoord coord{H, off_x, off_y};
    remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, dst_rows, dst_cols, coord, bg);
}

template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_homography_co(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                        uint8_t* __restrict__ dst, int dst_step, int dst_rows, int dst_cols,
                        int off_x, int off_y, HomographyArgs H, uint32_t bg)
{
    const HomographyCoord coord{H, off_x, off_y};
    remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, dst_rows, dst_cols, coord, bg);
}

template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR
void k_remap_mesh(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                  uint8_t* __restrict__ dst, int dst_step,
                  const float* __restrict__ mesh, int mesh_cols, int mesh_floats,
                  const LinTabEntry* __restrict__ xtab, const LinTabEntry* __restrict__ ytab, uint32_t bg)
{
    if (mesh_to_lds(mesh, mesh_floats))
        remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, src_rows, src_cols, MeshCoordT<true>{mesh, mesh_cols, xtab, ytab, (float)src_cols, (float)src_rows}, bg);
    else
        remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, src_rows, src_cols, MeshCoordT<false>{mesh, mesh_cols, xtab, ytab, (float)src_cols, (float)src_rows}, bg);
}

template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_mesh_co(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                  uint8_t* __restrict__ dst, int dst_step,
                  const float* __restrict__ mesh, int mesh_cols, int mesh_floats,
                  const LinTabEntry* __restrict__ xtab, const LinTabEntry* __restrict__ ytab, uint32_t bg)
{
    if (mesh_to_lds(mesh, mesh_floats))
        remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, src_rows, src_cols, MeshCoordT<true>{mesh, mesh_cols, xtab, ytab, (float)src_cols, (float)src_rows}, bg);
    else
        remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, src_rows, src_cols, MeshCoordT<false>{mesh, mesh_cols, xtab, ytab, (float)src_cols, (float)src_rows}, bg);
}

template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR
void k_remap_homography_lens(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                             uint8_t* __restrict__ dst, int dst_step, int dst_rows, int dst_cols,
                             int off_x, int off_y, HomographyArgs H, LensArgs L, uint32_t bg)
{
    const LensCoord<HomographyCoord> coord{HomographyCoord{H, off_x, off_y}, L, src_rows, src_cols};
    remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, dst_rows, dst_cols, coord, bg);
}

template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_homography_lens_co(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                             uint8_t* __restrict__ dst, int dst_step, int dst_rows, int dst_cols,
                             int off_x, int off_y, HomographyArgs H, LensArgs L, uint32_t bg)
{
    const LensCoord<HomographyCoord> coord{HomographyCoord{H, off_x, off_y}, L, src_rows, src_cols};
    remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, dst_rows, dst_cols, coord, bg);
}

template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR
void k_remap_mesh_lens(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                       uint8_t* __restrict__ dst, int dst_step,
                       const float* __restrict__ mesh, int mesh_cols, int mesh_floats,
                       const LinTabEntry* __restrict__ xtab, const LinTabEntry* __restrict__ ytab, LensArgs L, uint32_t bg)
{
    if (mesh_to_lds(mesh, mesh_floats))
        remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, src_rows, src_cols,
                         LensCoord<MeshCoordT<true>>{MeshCoordT<true>{mesh, mesh_cols, xtab, ytab, (float)src_cols, (float)src_rows}, L, src_rows, src_cols}, bg);
    else
        remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, src_rows, src_cols,
                         LensCoord<MeshCoordT<false>>{MeshCoordT<false>{mesh, mesh_cols, xtab, ytab, (float)src_cols, (float)src_rows}, L, src_rows, src_cols}, bg);
}

template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_mesh_lens_co(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                       uint8_t* __restrict__ dst, int dst_step,
                       const float* __restrict__ mesh, int mesh_cols, int mesh_floats,
                       const LinTabEntry* __restrict__ xtab, const LinTabEntry* __restrict__ ytab, LensArgs L, uint32_t bg)
{
    if (mesh_to_lds(mesh, mesh_floats))
        remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, src_rows, src_cols,
                         LensCoord<MeshCoordT<true>>{MeshCoordT<true>{mesh, mesh_cols, xtab, ytab, (float)src_cols, (float)src_rows}, L, src_rows, src_cols}, bg);
    else
        remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, src_rows, src_cols,
                         LensCoord<MeshCoordT<false>>{MeshCoordT<false>{mesh, mesh_cols, xtab, ytab, (float)src_cols, (float)src_rows}, L, src_rows, src_cols}, bg);
}

template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR
void k_remap_map(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                 uint8_t* __restrict__ dst, int dst_step, const uint8_t* __restrict__ map, int map_step, uint32_t bg)
{
    const MapCoord coord{map, map_step};
    remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, src_rows, src_cols, coord, bg);
}

// lvk::upscale (Image.cpp:155-202): the same strip body; the source coordinate never leaves the image, so the border band is the
// nearest copy of FSR.cl:342-351 and the background is unreachable.
template <bool YUV>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR
void k_easu_scale(const uint8_t* __restrict__ src, int src_step, int src_rows, int src_cols,
                  uint8_t* __restrict__ dst, int dst_step, int dst_rows, int dst_cols, float rsx, float rsy)
{
    const ScaleCoord coord{rsx, rsy};
    remap_strip<YUV>(src, src_step, src_rows, src_cols, PackedSink{dst, dst_step}, dst_rows, dst_cols, coord, 0u);
}

// ---- remap + 4:2:0 egress in one kernel (lvk_hip_stab_push_yuv420): YUV frames only, same size in and out, occupancy-capped like
//      the other kernels the overlap mode runs next to the tracker
struct Planes420 { uint8_t* y; int y_step; uint8_t* u; int u_step; uint8_t* v; int v_step; };

template <bool NV12>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_homography_420(const uint8_t* __restrict__ src, int src_step, int rows, int cols, Planes420 o, HomographyArgs H, uint32_t bg)
{
    LVK_TL(0);
    const HomographyCoord coord{H, 0, 0};
    remap_strip<true>(src, src_step, rows, cols, Sink420<NV12>{o.y, o.y_step, o.u, o.u_step, o.v, o.v_step}, rows, cols, coord, bg);
}

template <bool NV12>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_homography_lens_420(const uint8_t* __restrict__ src, int src_step, int rows, int cols, Planes420 o, HomographyArgs H, LensArgs L, uint32_t bg)
{
    const LensCoord<HomographyCoord> coord{HomographyCoord{H, 0, 0}, L, rows, cols};
    remap_strip<true>(src, src_step, rows, cols, Sink420<NV12>{o.y, o.y_step, o.u, o.u_step, o.v, o.v_step}, rows, cols, coord, bg);
}

template <bool NV12>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_mesh_420(const uint8_t* __restrict__ src, int src_step, int rows, int cols, Planes420 o,
                      const float* __restrict__ mesh, int mesh_cols, int mesh_floats, const LinTabEntry* __restrict__ xtab, const LinTabEntry* __restrict__ ytab, uint32_t bg)
{
    const Sink420<NV12> sink{o.y, o.y_step, o.u, o.u_step, o.v, o.v_step};
    if (mesh_to_lds(mesh, mesh_floats)) remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, MeshCoordT<true>{mesh, mesh_cols, xtab, ytab, (float)cols, (float)rows}, bg);
    else remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, MeshCoordT<false>{mesh, mesh_cols, xtab, ytab, (float)cols, (float)rows}, bg);
}

template <bool NV12>
__global__ __launch_bounds__(256) LVK_REMAP_ATTR LVK_CO_SCHEDULED
void k_remap_mesh_lens_420(const uint8_t* __restrict__ src, int src_step, int rows, int cols, Planes420 o,
                           const float* __restrict__ mesh, int mesh_cols, int mesh_floats, const LinTabEntry* __restrict__ xtab, const LinTabEntry* __restrict__ ytab,
                           LensArgs L, uint32_t bg)
{
    const Sink420<NV12> sink{o.y, o.y_step, o.u, o.u_step, o.v, o.v_step};
    if (mesh_to_lds(mesh, mesh_floats))
        remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, LensCoord<MeshCoordT<true>>{MeshCoordT<true>{mesh, mesh_cols, xtab, ytab, (float)cols, (float)rows}, L, rows, cols}, bg);
    else
        remap_strip<true>(src, src_step, rows, cols, sink, rows, cols, LensCoord<MeshCoordT<false>>{MeshCoordT<false>{mesh, mesh_cols, xtab, ytab, (float)cols, (float)rows}, L, rows, cols}, bg);
}


} // namespace

int lvk_launch_remap_homography(lvk_hip_ctx* ctx, hipStream_t stream,
                                const void* d_src, int src_step, int src_rows, int src_cols,
                                void* d_dst, int dst_step, int dst_rows, int dst_cols,
                                int off_x, int off_y, const float H[9], const uint8_t bg[3], int yuv, const LensArgs* lens, bool co)
{
    // Image.cpp:93-98
    LVK_HIP_REQUIRE(ctx, d_src != nullptr && d_dst != nullptr && H != nullptr && bg != nullptr);
    LVK_HIP_REQUIRE(ctx, src_cols > 0 && src_rows > 0 && dst_cols > 0 && dst_rows > 0);
    LVK_HIP_REQUIRE(ctx, src_step >= 3 * src_cols && dst_step >= 3 * dst_cols);
    LVK_HIP_REQUIRE(ctx, fits_u32(src_step, src_rows) && fits_u32(dst_step, dst_rows));
    HomographyArgs args;
    std::memcpy(args.h, H, sizeof(args.h));
    const dim3 block(256), grid = remap_grid(dst_rows, dst_cols), cogrid = lvk_co_grid(ctx, dst_rows, dst_cols);
#define LVK_LAUNCH_REMAP(K, ...)                                                                                          \
    do {                                                                                                                  \
        if (yuv) { if (co) hipLaunchKernelGGL(K##_co<true>, cogrid, block, 0, stream, __VA_ARGS__);                       \
                   else hipLaunchKernelGGL(K<true>, grid, block, 0, stream, __VA_ARGS__); }                               \
        else { if (co) hipLaunchKernelGGL(K##_co<false>, cogrid, block, 0, stream, __VA_ARGS__);                          \
               else hipLaunchKernelGGL(K<false>, grid, block, 0, stream, __VA_ARGS__); }                                  \
    } while (0)
    if (lens)
        LVK_LAUNCH_REMAP(k_remap_homography_lens, (const uint8_t*)d_src, src_step, src_rows, src_cols, (uint8_t*)d_dst, dst_step, dst_rows, dst_cols,
                         off_x, off_y, args, *lens, pack_bg(bg));
    else
        LVK_LAUNCH_REMAP(k_remap_homography, (const uint8_t*)d_src, src_step, src_rows, src_cols, (uint8_t*)d_dst, dst_step, dst_rows, dst_cols,
                         off_x, off_y, args, pack_bg(bg));
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

int lvk_launch_remap_mesh(lvk_hip_ctx* ctx, hipStream_t stream,
                          const void* d_src, int src_step, int src_rows, int src_cols,
                          void* d_dst, int dst_step,
                          const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv, const LensArgs* lens, bool co)
{
    // Image.cpp:30-34
    LVK_HIP_REQUIRE(ctx, d_src != nullptr && d_dst != nullptr && mesh != nullptr && bg != nullptr);
    LVK_HIP_REQUIRE(ctx, src_cols > 0 && src_rows > 0);
    LVK_HIP_REQUIRE(ctx, mesh_rows >= 2 && mesh_cols >= 2);           // WarpMesh::MinimumSize
    LVK_HIP_REQUIRE(ctx, src_step >= 3 * src_cols && dst_step >= 3 * src_cols);
    LVK_HIP_REQUIRE(ctx, fits_u32(src_step, src_rows) && fits_u32(dst_step, src_rows));
    const size_t mesh_bytes = (size_t)mesh_rows * mesh_cols * 2 * sizeof(float);
    LVK_HIP_REQUIRE(ctx, mesh_bytes <= lvk_hip_ctx::kStageBytes);

    // the tables first: once the mesh is staged nothing may fail before lvk_stage_consumed (a slot whose event was never re-recorded
    // could be rewritten while its copy is still in flight)
    const LinTabEntry *xtab = nullptr, *ytab = nullptr;
    int rc;
    if ((rc = lvk_get_lintab(ctx, mesh_cols, src_cols, false, &xtab)) != LVK_HIP_OK) return rc;
    if ((rc = lvk_get_lintab(ctx, mesh_rows, src_rows, true, &ytab)) != LVK_HIP_OK) return rc;
    void* d_mesh = nullptr; int stage_slot = 0;
    if ((rc = lvk_stage_params(ctx, stream, mesh, mesh_bytes, &d_mesh, &stage_slot)) != LVK_HIP_OK) return rc;

    const dim3 block(256), grid = remap_grid(src_rows, src_cols), cogrid = lvk_co_grid(ctx, src_rows, src_cols);
    if (lens)
        LVK_LAUNCH_REMAP(k_remap_mesh_lens, (const uint8_t*)d_src, src_step, src_rows, src_cols, (uint8_t*)d_dst, dst_step, (const float*)d_mesh, mesh_cols, mesh_rows * mesh_cols * 2,
                         xtab, ytab, *lens, pack_bg(bg));
    else
        LVK_LAUNCH_REMAP(k_remap_mesh, (const uint8_t*)d_src, src_step, src_rows, src_cols, (uint8_t*)d_dst, dst_step, (const float*)d_mesh, mesh_cols, mesh_rows * mesh_cols * 2,
                         xtab, ytab, pack_bg(bg));
#undef LVK_LAUNCH_REMAP
    const hipError_t le = hipGetLastError();
    rc = lvk_stage_consumed(ctx, stage_slot, stream);               // the slot is free again once this kernel has read the mesh (also after a failed launch)
    if (le != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(le));
    return rc;
}

// lvk::remap(src, dst, offset_map, background) with the map resident in HBM (Functions/Image.cpp:28-81): dst and map have
// the size of src (the path never uses map ROIs).  d_map: rows x cols float2 offsets in pixels, pitch map_step bytes.
int lvk_launch_remap_map(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                         void* d_dst, int dst_step, const void* d_map, int map_step, const uint8_t bg[3], int yuv)
{
    LVK_HIP_REQUIRE(ctx, d_src != nullptr && d_dst != nullptr && d_map != nullptr && bg != nullptr);     // Image.cpp:30-34
    LVK_HIP_REQUIRE(ctx, cols > 0 && rows > 0 && src_step >= 3 * cols && dst_step >= 3 * cols && map_step >= 8 * cols);
    LVK_HIP_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_map) | (uintptr_t)map_step) & 7u) == 0);
    LVK_HIP_REQUIRE(ctx, fits_u32(src_step, rows) && fits_u32(dst_step, rows) && fits_u32(map_step, rows));
    const dim3 block(256), grid = remap_grid(rows, cols);
    if (yuv) hipLaunchKernelGGL(k_remap_map<true>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, (uint8_t*)d_dst, dst_step, (const uint8_t*)d_map, map_step, pack_bg(bg));
    else hipLaunchKernelGGL(k_remap_map<false>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, rows, cols, (uint8_t*)d_dst, dst_step, (const uint8_t*)d_map, map_step, pack_bg(bg));
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

// lvk::upscale(src, dst, size, yuv) (Functions/Image.cpp:155-202)
int lvk_launch_upscale(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int src_rows, int src_cols,
                       void* d_dst, int dst_step, int dst_rows, int dst_cols, int yuv)
{
    LVK_HIP_REQUIRE(ctx, d_src != nullptr && d_dst != nullptr && d_src != d_dst);
    LVK_HIP_REQUIRE(ctx, src_cols > 0 && src_rows > 0);                                         // Image.cpp:158
    LVK_HIP_REQUIRE(ctx, dst_cols >= src_cols && dst_rows >= src_rows);                         // Image.cpp:157
    LVK_HIP_REQUIRE(ctx, src_step >= 3 * src_cols && dst_step >= 3 * dst_cols);
    LVK_HIP_REQUIRE(ctx, fits_u32(src_step, src_rows) && fits_u32(dst_step, dst_rows));
    if (dst_cols == src_cols && dst_rows == src_rows)                                           // Image.cpp:162-166
    {
        LVK_HIP_CHECK(ctx, hipMemcpy2DAsync(d_dst, (size_t)dst_step, d_src, (size_t)src_step, 3 * (size_t)src_cols, (size_t)src_rows,
                                            hipMemcpyDeviceToDevice, stream));
        return LVK_HIP_OK;
    }
    const float rsx = (float)src_cols / (float)dst_cols, rsy = (float)src_rows / (float)dst_rows;
    const dim3 block(256), grid = remap_grid(dst_rows, dst_cols);
    if (yuv) hipLaunchKernelGGL(k_easu_scale<true>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, src_rows, src_cols, (uint8_t*)d_dst, dst_step, dst_rows, dst_cols, rsx, rsy);
    else hipLaunchKernelGGL(k_easu_scale<false>, grid, block, 0, stream, (const uint8_t*)d_src, src_step, src_rows, src_cols, (uint8_t*)d_dst, dst_step, dst_rows, dst_cols, rsx, rsy);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

int lvk_launch_warpmesh_apply(lvk_hip_ctx* ctx, hipStream_t stream,
                              const void* d_src, int src_step, int rows, int cols,
                              void* d_dst, int dst_step,
                              const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv)
{
    return lvk_launch_warpmesh_apply_lens(ctx, stream, d_src, src_step, rows, cols, d_dst, dst_step, mesh, mesh_rows, mesh_cols, bg, yuv, nullptr, false);
}

int lvk_launch_warpmesh_apply_lens(lvk_hip_ctx* ctx, hipStream_t stream,
                                   const void* d_src, int src_step, int rows, int cols,
                                   void* d_dst, int dst_step,
                                   const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv, const LensArgs* lens, bool co)
{
    LVK_HIP_REQUIRE(ctx, mesh != nullptr && mesh_rows >= 2 && mesh_cols >= 2);
    if (mesh_rows == 2 && mesh_cols == 2)
    {
        // WarpMesh.cpp:194-217: corners + offsets * (cols, rows) -> getPerspectiveTransform(destination, source)
        const float w = (float)cols, h = (float)rows;
        const float dstp[8] = { 0, 0, w, 0, 0, h, w, h };
        float srcp[8];
        for (int i = 0; i < 4; i++)
        {
            // Point2f * Scalar: float * double, rounded back to float (Functions/Extensions.cpp operator*(Point2f, Scalar))
            const float mx = (float)((double)mesh[2 * i] * (double)cols);
            const float my = (float)((double)mesh[2 * i + 1] * (double)rows);
            srcp[2 * i] = dstp[2 * i] + mx;
            srcp[2 * i + 1] = dstp[2 * i + 1] + my;
        }
        double M[9];
        if (!perspective_transform(dstp, srcp, M))
            for (int q = 0; q < 9; q++) M[q] = (q % 4 == 0) ? 1.0 : 0.0;
        float H[9];
        for (int q = 0; q < 9; q++) H[q] = (float)M[q];              // Image.cpp:137-139
        return lvk_launch_remap_homography(ctx, stream, d_src, src_step, rows, cols, d_dst, dst_step, rows, cols, 0, 0, H, bg, yuv, lens, co);
    }
    return lvk_launch_remap_mesh(ctx, stream, d_src, src_step, rows, cols, d_dst, dst_step, mesh, mesh_rows, mesh_cols, bg, yuv, lens, co);
}

// WarpMesh::apply + I4XXIngest / NV12Ingest::to_obs in one launch: d_src packed YUV 8UC3, output planar 4:2:0 (I420: y, u, v; NV12: y, uv).
int lvk_launch_warpmesh_apply_420(lvk_hip_ctx* ctx, hipStream_t stream, const void* d_src, int src_step, int rows, int cols,
                                  void* o_y, int oy_step, void* o_u, int ou_step, void* o_v, int ov_step, int nv12,
                                  const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], const LensArgs* lens, bool co)
{
    LVK_HIP_REQUIRE(ctx, d_src && o_y && o_u && (nv12 || o_v) && mesh && bg && mesh_rows >= 2 && mesh_cols >= 2);
    LVK_HIP_REQUIRE(ctx, rows > 0 && cols > 0 && (rows & 1) == 0 && (cols & 1) == 0 && src_step >= 3 * cols);
    LVK_HIP_REQUIRE(ctx, oy_step >= cols && ou_step >= (nv12 ? cols : cols / 2) && (nv12 || ov_step >= cols / 2));
    LVK_HIP_REQUIRE(ctx, fits_u32(src_step, rows) && fits_u32(oy_step, rows) && fits_u32(ou_step, rows / 2) && (nv12 || fits_u32(ov_step, rows / 2)));
    const Planes420 o{(uint8_t*)o_y, oy_step, (uint8_t*)o_u, ou_step, (uint8_t*)(nv12 ? o_u : o_v), nv12 ? ou_step : ov_step};
    const dim3 block(256), grid = co ? lvk_co_grid(ctx, rows, cols) : remap_grid(rows, cols);
    const size_t lds_pad = co ? (size_t)LVK_CO_LDS_PAD : 0;          // A / B switch (scripts/variant_build.sh): unused LDS that caps the persistent grid's blocks per CU
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
        if (lens)
        {
            if (nv12) hipLaunchKernelGGL(k_remap_homography_lens_420<true>, grid, block, lds_pad, stream, (const uint8_t*)d_src, src_step, rows, cols, o, args, *lens, pack_bg(bg));
            else hipLaunchKernelGGL(k_remap_homography_lens_420<false>, grid, block, lds_pad, stream, (const uint8_t*)d_src, src_step, rows, cols, o, args, *lens, pack_bg(bg));
        }
        else
        {
            if (nv12) hipLaunchKernelGGL(k_remap_homography_420<true>, grid, block, lds_pad, stream, (const uint8_t*)d_src, src_step, rows, cols, o, args, pack_bg(bg));
            else hipLaunchKernelGGL(k_remap_homography_420<false>, grid, block, lds_pad, stream, (const uint8_t*)d_src, src_step, rows, cols, o, args, pack_bg(bg));
        }
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
        if (lens)
        {
            if (nv12) hipLaunchKernelGGL(k_remap_mesh_lens_420<true>, grid, block, lds_pad, stream, (const uint8_t*)d_src, src_step, rows, cols, o, (const float*)d_mesh, mesh_cols, mesh_rows * mesh_cols * 2, xtab, ytab, *lens, pack_bg(bg));
            else hipLaunchKernelGGL(k_remap_mesh_lens_420<false>, grid, block, lds_pad, stream, (const uint8_t*)d_src, src_step, rows, cols, o, (const float*)d_mesh, mesh_cols, mesh_rows * mesh_cols * 2, xtab, ytab, *lens, pack_bg(bg));
        }
        else
        {
            if (nv12) hipLaunchKernelGGL(k_remap_mesh_420<true>, grid, block, lds_pad, stream, (const uint8_t*)d_src, src_step, rows, cols, o, (const float*)d_mesh, mesh_cols, mesh_rows * mesh_cols * 2, xtab, ytab, pack_bg(bg));
            else hipLaunchKernelGGL(k_remap_mesh_420<false>, grid, block, lds_pad, stream, (const uint8_t*)d_src, src_step, rows, cols, o, (const float*)d_mesh, mesh_cols, mesh_rows * mesh_cols * 2, xtab, ytab, pack_bg(bg));
        }
    }
    const hipError_t le = hipGetLastError();
    const int src = stage_slot >= 0 ? lvk_stage_consumed(ctx, stage_slot, stream) : LVK_HIP_OK;
    if (le != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(le));
    return src;
}

extern "C" {

int lvk_hip_remap_homography(lvk_hip_ctx* ctx,
                             const void* d_src, int src_step, int src_rows, int src_cols,
                             void* d_dst, int dst_step, int dst_rows, int dst_cols,
                             int off_x, int off_y, const float H[9], const uint8_t bg[3], int yuv)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_remap_homography(ctx, ctx->stream, d_src, src_step, src_rows, src_cols, d_dst, dst_step, dst_rows, dst_cols, off_x, off_y, H, bg, yuv);
}

int lvk_hip_remap_mesh(lvk_hip_ctx* ctx,
                       const void* d_src, int src_step, int src_rows, int src_cols,
                       void* d_dst, int dst_step,
                       const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_remap_mesh(ctx, ctx->stream, d_src, src_step, src_rows, src_cols, d_dst, dst_step, mesh, mesh_rows, mesh_cols, bg, yuv);
}

int lvk_hip_remap_map(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step,
                      const void* d_map, int map_step, const uint8_t bg[3], int yuv)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_remap_map(ctx, ctx->stream, d_src, src_step, rows, cols, d_dst, dst_step, d_map, map_step, bg, yuv);
}

int lvk_hip_warpmesh_apply_lens(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols, void* d_dst, int dst_step,
                                const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv, const lvk_camera_params* lens)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, lens != nullptr && rows > 1 && cols > 1);
    LensModel m; LensArgs a;
    const int rc = lvk_lens_model_build(*lens, rows, cols, m);
    if (rc != LVK_HIP_OK) return ctx->fail(rc, "invalid camera profile");
    std::memcpy(a.f, m.f, sizeof(a.f));
    return lvk_launch_warpmesh_apply_lens(ctx, ctx->stream, d_src, src_step, rows, cols, d_dst, dst_step, mesh, mesh_rows, mesh_cols, bg, yuv, &a, false);
}

int lvk_hip_warpmesh_apply_yuv420(lvk_hip_ctx* ctx, const void* d_src, int src_step, int rows, int cols,
                                  void* o_y, int oy_step, void* o_u, int ou_step, void* o_v, int ov_step, int nv12,
                                  const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3])
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_warpmesh_apply_420(ctx, ctx->stream, d_src, src_step, rows, cols, o_y, oy_step, o_u, ou_step, o_v, ov_step, nv12,
                                         mesh, mesh_rows, mesh_cols, bg, nullptr, false);
}

int lvk_hip_upscale(lvk_hip_ctx* ctx, const void* d_src, int src_step, int src_rows, int src_cols,
                    void* d_dst, int dst_step, int dst_rows, int dst_cols, int yuv)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_upscale(ctx, ctx->stream, d_src, src_step, src_rows, src_cols, d_dst, dst_step, dst_rows, dst_cols, yuv);
}

int lvk_hip_warpmesh_apply(lvk_hip_ctx* ctx,
                           const void* d_src, int src_step, int rows, int cols,
                           void* d_dst, int dst_step,
                           const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3], int yuv)
{
    LVK_HIP_ENTRY(ctx);
    return lvk_launch_warpmesh_apply(ctx, ctx->stream, d_src, src_step, rows, cols, d_dst, dst_step, mesh, mesh_rows, mesh_cols, bg, yuv);
}

} // extern "C"

LVK_TL_EXPORT(remap)
