// Pyramidal Lucas-Kanade sparse optical flow for gfx950: one 64-lane wavefront per feature, all pyramid levels
// inside one launch, patch windows staged in LDS, integer fixed-point bilinear sampling and exact integer
// wave reductions for the 2x2 system.
//
// Replaces cv::SparsePyrLKOpticalFlow::calc as the reference configures it (reference call:
// LiveVisionKit/Vision/FrameTracker.cpp:33-35,42-48,140-146; arithmetic: OpenCV 4.8.0 video/lkpyramid.cpp
// LKTrackerInvoker, CPU fixed-point path).  The covariance / mismatch sums are exact
// integers converted to binary32 once, every other float op is the one written in lkpyramid.cpp (no contraction),
// so results are bit-identical to the specification regardless of lane order.
#include "lvk_hip_internal.hpp"

namespace {

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { p = (p < 0) ? -p : 2 * (len - 1) - p; }
    return p;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ __forceinline__ long long wave_sum(long long v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11)
{
    w00 = (int)__builtin_rintf((1.f - a) * (1.f - b) * 16384.0f);
    w01 = (int)__builtin_rintf(a * (1.f - b) * 16384.0f);
    w10 = (int)__builtin_rintf((1.f - a) * b * 16384.0f);
    w11 = 16384 - w00 - w01 - w10;
}

// LDS layout per block (one wave): tile[(win_h+1) * tw] u8 image samples, dtile[... ] short2 derivative samples,
// then the cached patches I, Ix, Iy (int16 each, win_w * win_h).
__global__ __launch_bounds__(64)
void k_pyrlk(PyrArgs prev, PyrArgs next, const float2* __restrict__ prev_pts, int n,
             float2* __restrict__ next_pts, uint8_t* __restrict__ status,
             int win_w, int win_h, int max_count, double epsilon_sq, float min_eig_threshold)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int pt = blockIdx.x;
    if (pt >= n) return;
    const int lane = threadIdx.x;
    const int tw = win_w + 1, th = win_h + 1, area = win_w * win_h, tarea = tw * th;
    short2* dtile = reinterpret_cast<short2*>(smem);                          // tarea * 4 B
    short* Iw = reinterpret_cast<short*>(smem + (size_t)tarea * 4);           // area * 2 B
    short* Ixw = Iw + area;
    short* Iyw = Ixw + area;
    uint8_t* tile = reinterpret_cast<uint8_t*>(Iyw + area);                   // tarea B

    const float2 p0 = prev_pts[pt];
    const float halfx = (win_w - 1) * 0.5f, halfy = (win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    float outx = 0.f, outy = 0.f;
    bool ok = true;                                                           // status (initialised to 1 by calc())
    const int top = prev.nlevels - 1;

    for (int level = top; level >= 0; level--)
    {
        const PyrLevel I = prev.lv[level];
        const PyrLevel J = next.lv[level];
        float px = p0.x * (float)(1. / (1 << level));
        float py = p0.y * (float)(1. / (1 << level));
        float nx, ny;
        if (level == top) { nx = px; ny = py; }
        else { nx = outx * 2.f; ny = outy * 2.f; }
        outx = nx; outy = ny;

        px -= halfx; py -= halfy;
        const int ipx = (int)__builtin_floorf(px), ipy = (int)__builtin_floorf(py);
        if (ipx < -win_w || ipx >= I.cols || ipy < -win_h || ipy >= I.rows)
        {
            if (level == 0) ok = false;
            continue;
        }
        float a = px - ipx, b = py - ipy;
        int w00, w01, w10, w11;
        bilinear_weights(a, b, w00, w01, w10, w11);

        // stage the (win+1)^2 window of the previous image (reflect-101 border) and its derivatives (zero border)
        __syncthreads();
        for (int i = lane; i < tarea; i += 64)
        {
            const int ty = i / tw, tx = i - ty * tw;
            const int yy = ipy + ty, xx = ipx + tx;
            tile[i] = I.img[(long)reflect101(yy, I.rows) * I.step + reflect101(xx, I.cols)];
            short2 d = make_short2(0, 0);
            if (xx >= 0 && yy >= 0 && xx < I.cols && yy < I.rows) d = I.deriv[(long)yy * I.cols + xx];
            dtile[i] = d;
        }
        __syncthreads();
        long long sA11 = 0, sA12 = 0, sA22 = 0;
        for (int p = lane; p < area; p += 64)
        {
            const int y = p / win_w, x = p - y * win_w;
            const int i00 = y * tw + x, i01 = i00 + 1, i10 = i00 + tw, i11 = i10 + 1;
            const int ival = descale(tile[i00] * w00 + tile[i01] * w01 + tile[i10] * w10 + tile[i11] * w11, 14 - 5);
            const int ixval = descale(dtile[i00].x * w00 + dtile[i01].x * w01 + dtile[i10].x * w10 + dtile[i11].x * w11, 14);
            const int iyval = descale(dtile[i00].y * w00 + dtile[i01].y * w01 + dtile[i10].y * w10 + dtile[i11].y * w11, 14);
            Iw[p] = (short)ival; Ixw[p] = (short)ixval; Iyw[p] = (short)iyval;
            sA11 += (long long)ixval * ixval;
            sA12 += (long long)ixval * iyval;
            sA22 += (long long)iyval * iyval;
        }
        sA11 = wave_sum(sA11); sA12 = wave_sum(sA12); sA22 = wave_sum(sA22);
        const float A11 = (float)(double)sA11 * FLT_SCALE, A12 = (float)(double)sA12 * FLT_SCALE, A22 = (float)(double)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - __builtin_sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win_w * win_h);
        if (minEig < min_eig_threshold || D < 1.1920928955078125e-07f)
        {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.f / D;
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
            a = nx - inx; b = ny - iny;
            bilinear_weights(a, b, w00, w01, w10, w11);
            __syncthreads();
            for (int i = lane; i < tarea; i += 64)
            {
                const int ty = i / tw, tx = i - ty * tw;
                tile[i] = J.img[(long)reflect101(iny + ty, J.rows) * J.step + reflect101(inx + tx, J.cols)];
            }
            __syncthreads();
            long long sb1 = 0, sb2 = 0;
            for (int p = lane; p < area; p += 64)
            {
                const int y = p / win_w, x = p - y * win_w;
                const int i00 = y * tw + x, i01 = i00 + 1, i10 = i00 + tw, i11 = i10 + 1;
                const int diff = descale(tile[i00] * w00 + tile[i01] * w01 + tile[i10] * w10 + tile[i11] * w11, 14 - 5) - Iw[p];
                sb1 += (long long)diff * Ixw[p];
                sb2 += (long long)diff * Iyw[p];
            }
            sb1 = wave_sum(sb1); sb2 = wave_sum(sb2);
            const float b1 = (float)(double)sb1 * FLT_SCALE, b2 = (float)(double)sb2 * FLT_SCALE;
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
        }
    }
    if (lane == 0)
    {
        next_pts[pt] = make_float2(outx, outy);
        status[pt] = ok ? 1 : 0;
    }
}

} // namespace

size_t lvk_pyrlk_lds_bytes(int win_w, int win_h)
{
    const size_t tarea = (size_t)(win_w + 1) * (win_h + 1), area = (size_t)win_w * win_h;
    return ((tarea * 4 + area * 6 + tarea) + 15) & ~(size_t)15;
}

int lvk_launch_pyrlk(lvk_hip_ctx* ctx, const PyrArgs& prev, const PyrArgs& next, const float2* d_prev_pts, int n,
                     float2* d_next_pts, uint8_t* d_status, int win_w, int win_h, int max_count, double epsilon, double min_eig)
{
    LVK_HIP_REQUIRE(ctx, prev.nlevels >= 1 && prev.nlevels == next.nlevels && prev.nlevels <= LVK_MAX_PYR_LEVELS);
    LVK_HIP_REQUIRE(ctx, win_w >= 3 && win_h >= 3 && win_w <= 31 && win_h <= 31);
    if (n <= 0) return LVK_HIP_OK;
    // SparsePyrLKOpticalFlowImpl: criteria clamp, epsilon squared
    max_count = std::min(std::max(max_count, 0), 100);
    epsilon = std::min(std::max(epsilon, 0.), 10.);
    epsilon *= epsilon;
    hipLaunchKernelGGL(k_pyrlk, dim3(n), dim3(64), lvk_pyrlk_lds_bytes(win_w, win_h), ctx->stream, prev, next, d_prev_pts, n,
                       d_next_pts, d_status, win_w, win_h, max_count, epsilon, (float)min_eig);
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
int DevicePyramid::build(lvk_hip_ctx* ctx)
{
    int rc;
    for (int i = 1; i < args.nlevels; i++)
        if ((rc = lvk_launch_pyr_down(ctx, args.lv[i - 1].img, args.lv[i - 1].step, args.lv[i - 1].rows, args.lv[i - 1].cols,
                                      const_cast<uint8_t*>(args.lv[i].img), args.lv[i].step)) != LVK_HIP_OK) return rc;
    for (int i = 0; i < args.nlevels; i++)
        if ((rc = lvk_launch_scharr(ctx, args.lv[i].img, args.lv[i].step, args.lv[i].rows, args.lv[i].cols,
                                    const_cast<short2*>(args.lv[i].deriv))) != LVK_HIP_OK) return rc;
    return LVK_HIP_OK;
}

extern "C" {

// Synchronous test entry point mirroring calc(prevImg, nextImg, prevPts, nextPts, status): device images of the
// tracking resolution, host point arrays.
int lvk_hip_pyrlk(lvk_hip_ctx* ctx, const void* d_prev, int prev_step, const void* d_next, int next_step, int rows, int cols,
                  const float* prev_pts, int n, float* next_pts, uint8_t* status,
                  int win_w, int win_h, int max_level, int max_count, double epsilon, double min_eig_threshold)
{
    if (!ctx) return LVK_HIP_ERR_ARG;
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

} // extern "C"
