// ORACLE (test infrastructure only -- see lvk_oracle.h).
// CPU restatement of the reference's dense remap: Functions/OpenCL/Sources/FSR.cl:55-452 (kernels
// easu_remap / easu_remap_homography and helpers), launched by Functions/Image.cpp:28-151 from
// Math/WarpMesh.cpp:183-223.  Scalar, one output pixel at a time, in the reference's own order.
//
// Arithmetic definition.  The reference is OpenCL C, compiled at run time for the device it runs on; the oracle follows what the
// reference's own source compiles to for gfx950 with the image's ROCm clang (oracle/_ref/fsr_*.hsaco, `make -C oracle ref`; the
// GPU tests check oracle == that code object == the HIP kernels bit for bit, tests/test_ref_pin_gpu.py):
//   * contraction is clang's FP_CONTRACT ON: a multiply feeding an add INSIDE ONE EXPRESSION is one fused fmaf (`x * y + z`,
//     `z + x * y`, `acc += x * y`); a sum of two products `a*b + c*d` is fmaf(a, b, c*d); `a*b + c*d + e` is fmaf(a, b, c*d) + e.
//     Everything else is a separately rounded binary32 op.
//   * native_recip(x) and `1.0f / x` (OpenCL's 2.5 ulp divide) both compile to v_frexp_mant / v_rcp_f32 / v_ldexp: the device's
//     reciprocal, within 1 ulp of 1/x (exact for 89 % of the mantissas, scripts/rcp_probe.hip), sign-symmetric and exponent
//     independent.  native_rcp() below models it with the table of v_rcp_f32 over the 2^23 mantissas, which the GPU tests read from
//     the device and install with lvko_set_device_rcp_table.  WITHOUT a table (the CPU-only tests, fixtures under tests/golden/)
//     it is the correctly rounded 1.0f / x: at most 1 LSB away in about 2 of 10^5 output bytes.
//   * convert_int2_rtz saturates (NaN -> 0); convert_uchar3 truncates (values are within [0,255]).
//   * min/max are fminf/fmaxf.
#include "lvk_oracle.h"

#include <cmath>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

namespace {

inline float as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// The device reciprocal (see the header comment): g_rcp_tab[m] = v_rcp_f32(1.m) in (0.5, 1] for the 23 mantissa bits m.
std::vector<float> g_rcp_tab;
inline float native_rcp(float x)
{
    if (g_rcp_tab.empty() || x == 0.0f || std::isinf(x) || std::isnan(x)) return 1.0f / x;
    int e;
    const float m = frexpf(x, &e);                                   // |m| in [0.5, 1), x = m * 2^e  (v_frexp_mant / v_frexp_exp)
    const float r = g_rcp_tab[as_uint(m) & 0x7fffffu];               // rcp(1.mant); rcp(|m|) = 2 * r
    return copysignf(ldexpf(r, 1 - e), x);                           // v_ldexp_f32
}

// FSR.cl:60,65
inline float APrxLoRsqF1(float a) { return as_float(0x5f347d74u - (as_uint(a) >> 1)); }
inline float APrxLoRcpF1(float a) { return as_float(0x7ef07ebbu - as_uint(a)); }
// FSR.cl:79
inline float saturate(float x) { return fmaxf(0.0f, fminf(1.0f, x)); }

struct F3 { float x, y, z; };

// FSR.cl:98-126
inline void easu_tap(F3& aC, float& aW, float offx, float offy, float dirx, float diry,
                     float lenx, float leny, float lob, float clp, const F3& c)
{
    float vx = fmaf(offx, dirx, offy * diry);
    float vy = fmaf(offx, -diry, offy * dirx);
    vx *= lenx;
    vy *= leny;
    float d2 = fminf(fmaf(vx, vx, vy * vy), clp);
    float wA = fmaf(lob, d2, -1.0f);
    float wB = fmaf(2.0f / 5.0f, d2, -1.0f);
    wA *= wA;
    wB = fmaf(25.0f / 16.0f, wB * wB, -(25.0f / 16.0f - 1.0f));
    float w = wB * wA;
    aC.x = fmaf(c.x, w, aC.x);
    aC.y = fmaf(c.y, w, aC.y);
    aC.z = fmaf(c.z, w, aC.z);
    aW += w;
}

// FSR.cl:131-176 (w is the bilinear weight selected by the biS/biT/biU/biV predicates)
inline void easu_accumulate(float& dirx, float& diry, float& len, float w,
                            float lA, float lB, float lC, float lD, float lE)
{
    float dc = lD - lC;
    float cb = lC - lB;
    float lenX = APrxLoRcpF1(fmaxf(fabsf(dc), fabsf(cb)));
    float dirX = lD - lB;
    dirx = fmaf(dirX, w, dirx);
    lenX = saturate(fabsf(dirX) * lenX);
    lenX *= lenX;
    len = fmaf(lenX, w, len);

    float ec = lE - lC;
    float ca = lC - lA;
    float lenY = APrxLoRcpF1(fmaxf(fabsf(ec), fabsf(ca)));
    float dirY = lE - lA;
    diry = fmaf(dirY, w, diry);
    lenY = saturate(fabsf(dirY) * lenY);
    lenY *= lenY;
    len = fmaf(lenY, w, len);
}

inline F3 load_px(const uint8_t* p)
{
    const float norm_factor = 0.00392156862f;                         // FSR.cl:205
    return F3{ (float)p[0] * norm_factor, (float)p[1] * norm_factor, (float)p[2] * norm_factor };
}

// FSR.cl:181-318.  (sx, sy) = src_coord, (ppx, ppy) = sub_pixel.
inline void easu(const uint8_t* src, int step, int sx, int sy, float ppx, float ppy, bool yuv, uint8_t out[3])
{
    //      b c
    //    e f g h
    //    i j k l
    //      n o
    const uint8_t* r0 = src + (size_t)(sy - 1) * step + 3 * sx;        // b at (sx, sy-1)
    const uint8_t* r1 = r0 + step - 3;                                // e at (sx-1, sy)
    const uint8_t* r2 = r1 + step;                                    // i at (sx-1, sy+1)
    const uint8_t* r3 = r0 + 3 * (size_t)step;                        // n at (sx, sy+2)

    const F3 b = load_px(r0), c = load_px(r0 + 3);
    const F3 e = load_px(r1), f = load_px(r1 + 3), g = load_px(r1 + 6), h = load_px(r1 + 9);
    const F3 i = load_px(r2), j = load_px(r2 + 3), k = load_px(r2 + 6), l = load_px(r2 + 9);
    const F3 n = load_px(r3), o = load_px(r3 + 3);

    // FSR.cl:229-241 -- note the reference's inverted macro: the YUV program uses the 3-channel luma.
    auto luma = [yuv](const F3& p) -> float {
        return yuv ? fmaf(p.z, 0.5f, fmaf(p.x, 0.5f, p.y)) : p.x;
    };
    const float bL = luma(b), cL = luma(c), eL = luma(e), fL = luma(f), gL = luma(g), hL = luma(h);
    const float iL = luma(i), jL = luma(j), kL = luma(k), lL = luma(l), nL = luma(n), oL = luma(o);

    // FSR.cl:244-249
    float len = 0.0f, dirx = 0.0f, diry = 0.0f;
    const float omx = 1.0f - ppx, omy = 1.0f - ppy;
    easu_accumulate(dirx, diry, len, omx * omy, bL, eL, fL, gL, jL);   // s
    easu_accumulate(dirx, diry, len, ppx * omy, cL, fL, gL, hL, kL);   // t
    easu_accumulate(dirx, diry, len, omx * ppy, fL, iL, jL, kL, nL);   // u
    easu_accumulate(dirx, diry, len, ppx * ppy, gL, jL, kL, lL, oL);   // v

    // FSR.cl:252-258
    // FSR.cl:252-253 are TWO statements (dir2 = dir * dir; dirR = dir2.x + dir2.y): clang contracts within an expression only
    float dirR = dirx * dirx + diry * diry;
    const bool zro = dirR < (1.0f / 32768.0f);
    dirR = APrxLoRsqF1(dirR);
    dirR = zro ? 1.0f : dirR;
    dirx = zro ? 1.0f : dirx;
    dirx *= dirR;
    diry *= dirR;

    // FSR.cl:261-277
    len = len * 0.5f;
    len *= len;
    const float stretch = fmaf(dirx, dirx, diry * diry) * APrxLoRcpF1(fmaxf(fabsf(dirx), fabsf(diry)));
    const float len2x = fmaf(stretch - 1.0f, len, 1.0f);
    const float len2y = fmaf(-0.5f, len, 1.0f);
    const float lob = fmaf((1.0f / 4.0f - 0.04f) - 0.5f, len, 0.5f);
    const float clp = APrxLoRcpF1(lob);

    // FSR.cl:284-296  min/max of the 2x2 centre (f, g, j, k)
    F3 mi4{ fminf(f.x, fminf(g.x, fminf(j.x, k.x))), fminf(f.y, fminf(g.y, fminf(j.y, k.y))), fminf(f.z, fminf(g.z, fminf(j.z, k.z))) };
    F3 ma4{ fmaxf(f.x, fmaxf(g.x, fmaxf(j.x, k.x))), fmaxf(f.y, fmaxf(g.y, fmaxf(j.y, k.y))), fmaxf(f.z, fmaxf(g.z, fmaxf(j.z, k.z))) };

    // FSR.cl:299-313 (tap order preserved)
    F3 aC{0.0f, 0.0f, 0.0f};
    float aW = 0.0f;
    easu_tap(aC, aW,  0.0f - ppx, -1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, b);
    easu_tap(aC, aW,  1.0f - ppx, -1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, c);
    easu_tap(aC, aW, -1.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, i);
    easu_tap(aC, aW,  0.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, j);
    easu_tap(aC, aW,  0.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, f);
    easu_tap(aC, aW, -1.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, e);
    easu_tap(aC, aW,  1.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, k);
    easu_tap(aC, aW,  2.0f - ppx,  1.0f - ppy, dirx, diry, len2x, len2y, lob, clp, l);
    easu_tap(aC, aW,  2.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, h);
    easu_tap(aC, aW,  1.0f - ppx,  0.0f - ppy, dirx, diry, len2x, len2y, lob, clp, g);
    easu_tap(aC, aW,  0.0f - ppx,  2.0f - ppy, dirx, diry, len2x, len2y, lob, clp, n);
    easu_tap(aC, aW,  1.0f - ppx,  2.0f - ppy, dirx, diry, len2x, len2y, lob, clp, o);

    // FSR.cl:316-317
    const float rW = native_rcp(aW);
    const float px = fminf(ma4.x, fmaxf(mi4.x, aC.x * rW));
    const float py = fminf(ma4.y, fmaxf(mi4.y, aC.y * rW));
    const float pz = fminf(ma4.z, fmaxf(mi4.z, aC.z * rW));
    out[0] = (uint8_t)(int)(px * 255.0f);
    out[1] = (uint8_t)(int)(py * 255.0f);
    out[2] = (uint8_t)(int)(pz * 255.0f);
}

inline int cvt_int_rtz_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int)(-2147483647 - 1);
    return (int)v;   // C conversion truncates toward zero
}

// Shared tail of FSR.cl:380-402 / 429-451: given the source coordinate of one output pixel.
inline void remap_pixel(const uint8_t* src, int src_step, int src_rows, int src_cols,
                        uint8_t* dpx, float subx, float suby, const uint8_t bg[3], bool yuv)
{
    const int sx = cvt_int_rtz_sat(subx);
    const int sy = cvt_int_rtz_sat(suby);
    const float ppx = subx - floorf(subx);
    const float ppy = suby - floorf(suby);

    if (sx < 1 || sy < 1 || sx >= src_cols - 4 || sy >= src_rows - 4)
    {
        if (sx >= 0 && sx < src_cols && sy >= 0 && sy < src_rows)
        {
            const uint8_t* s = src + (size_t)sy * src_step + 3 * sx;
            dpx[0] = s[0]; dpx[1] = s[1]; dpx[2] = s[2];
        }
        else { dpx[0] = bg[0]; dpx[1] = bg[1]; dpx[2] = bg[2]; }
        return;
    }
    easu(src, src_step, sx, sy, ppx, ppy, yuv, dpx);
}

// Fused lens pre-warp (oracle/lens.cpp lvko_lens_model, rounded to binary32): (u, v) = position in the lens-corrected frame
// the stabilizing warp asks for -> (subx, suby) = position in the RAW frame.  Returns false when (u, v) itself is outside the
// corrected frame (background, as the second pass of the reference chain would decide).
// L[17] = 1/nfx, 1/nfy, ncx, ncy, fx, fy, cx, cy, k1, k2, p1, p2, k3, kxc, vxc, kyc, vyc.
inline bool lens_forward(const float* L, int rows, int cols, float u, float v, float& subx, float& suby)
{
    const int ux = cvt_int_rtz_sat(u), vy = cvt_int_rtz_sat(v);
    if (ux < 0 || ux >= cols || vy < 0 || vy >= rows) return false;
    const float x = (u - L[2]) * L[0], y = (v - L[3]) * L[1];
    const float r2 = fmaf(x, x, y * y);
    const float kr = fmaf(fmaf(fmaf(L[12], r2, L[9]), r2, L[8]), r2, 1.0f);
    const float xy2 = (x + x) * y;
    const float xd = fmaf(x, kr, fmaf(L[10], xy2, L[11] * fmaf(x + x, x, r2)));
    const float yd = fmaf(y, kr, fmaf(L[10], fmaf(y + y, y, r2), L[11] * xy2));
    subx = fmaf(L[4], xd, L[6]) + fmaf(u, L[13], L[14]);
    suby = fmaf(L[5], yd, L[7]) + fmaf(v, L[15], L[16]);
    return true;
}

inline void remap_pixel_lens(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dpx, float u, float v,
                             const uint8_t bg[3], bool yuv, const float* L)
{
    float sx, sy;
    if (!L) { remap_pixel(src, src_step, rows, cols, dpx, u, v, bg, yuv); return; }
    if (!lens_forward(L, rows, cols, u, v, sx, sy)) { dpx[0] = bg[0]; dpx[1] = bg[1]; dpx[2] = bg[2]; return; }
    remap_pixel(src, src_step, rows, cols, dpx, sx, sy, bg, yuv);
}

template <class Fn>
void parallel_rows(int rows, int nthreads, Fn fn)
{
    if (nthreads <= 1 || rows < 2 * nthreads) { fn(0, rows); return; }
    std::vector<std::thread> pool;
    const int chunk = (rows + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++)
    {
        const int r0 = t * chunk, r1 = std::min(rows, r0 + chunk);
        if (r0 >= r1) break;
        pool.emplace_back([=] { fn(r0, r1); });
    }
    for (auto& th : pool) th.join();
}

// cv::resize(INTER_LINEAR) coordinate table on float data (OpenCV 4.8 imgproc resize.cpp):
// fx = (float)((d + 0.5) * scale - 0.5); s = floor(fx); fx -= s; columns clamp (sx, fx) at the edges and use a
// single tap beyond xmax; rows keep (1-fy, fy) and clip the two row indices instead.
struct LinTab { std::vector<int> s0, s1; std::vector<float> a0, a1; };

LinTab make_lintab(int ssize, int dsize, bool vertical)
{
    LinTab t; t.s0.resize(dsize); t.s1.resize(dsize); t.a0.resize(dsize); t.a1.resize(dsize);
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; d++)
    {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(fx);
        fx -= (float)s;
        if (vertical)
        {
            // resizeGeneric_Invoker: beta = (1-fy, fy) unclamped; the two source rows are clipped
            // individually: clip(sy, 0, h), clip(sy+1, 0, h).
            t.s0[d] = std::min(std::max(s, 0), ssize - 1);
            t.s1[d] = std::min(std::max(s + 1, 0), ssize - 1);
            t.a0[d] = 1.0f - fx;
            t.a1[d] = fx;
            continue;
        }
        if (s < 0) { fx = 0.0f; s = 0; }
        bool single = false;
        if (s + 1 >= ssize)                      // sx + ksize2 >= ssize.width (ksize2 = 1): dx >= xmax
        {
            single = true;                       // HResizeLinear tail: D = S[sx] * 1
            if (s >= ssize - 1) { fx = 0.0f; s = ssize - 1; }
        }
        t.s0[d] = s;
        t.s1[d] = single ? s : s + 1;
        t.a0[d] = single ? 1.0f : 1.0f - fx;
        t.a1[d] = single ? 0.0f : fx;
    }
    return t;
}

} // namespace

extern "C" {

static int remap_homography_impl(const uint8_t* src, int src_step, int src_rows, int src_cols,
                          uint8_t* dst, int dst_step, int dst_rows, int dst_cols,
                          int off_x, int off_y, const float H[9], const uint8_t bg[3],
                          int yuv, int nthreads, const float* L)
{
    if (!src || !dst || src_rows <= 0 || src_cols <= 0) return -1;
    parallel_rows(dst_rows, nthreads, [=](int r0, int r1) {
        for (int y = r0; y < r1; y++)
        {
            uint8_t* drow = dst + (size_t)y * dst_step;
            for (int x = 0; x < dst_cols; x++)
            {
                // FSR.cl:422-427
                const float fx = (float)x, fy = (float)y;
                // `r.x * fx + r.y * fy + r.z` parses as ((r.x * fx) + (r.y * fy)) + r.z: one fused multiply-add, then a plain add
                const float dz = native_rcp(fmaf(H[6], fx, H[7] * fy) + H[8]);
                const float ox = (fmaf(H[0], fx, H[1] * fy) + H[2]) * dz - fx;
                const float oy = (fmaf(H[3], fx, H[4] * fy) + H[5]) * dz - fy;
                // FSR.cl:430
                const float subx = (float)(x + off_x) + ox;
                const float suby = (float)(y + off_y) + oy;
                remap_pixel_lens(src, src_step, src_rows, src_cols, drow + 3 * x, subx, suby, bg, yuv != 0, L);
            }
        }
    });
    return 0;
}

int lvko_remap_homography(const uint8_t* src, int src_step, int src_rows, int src_cols,
                          uint8_t* dst, int dst_step, int dst_rows, int dst_cols,
                          int off_x, int off_y, const float H[9], const uint8_t bg[3],
                          int yuv, int nthreads)
{
    return remap_homography_impl(src, src_step, src_rows, src_cols, dst, dst_step, dst_rows, dst_cols, off_x, off_y, H, bg, yuv, nthreads, nullptr);
}

void lvko_mesh_to_map(const float* mesh, int mesh_rows, int mesh_cols, int rows, int cols, float* map)
{
    // WarpMesh.cpp:190-191: resize(offsets, src.size, INTER_LINEAR_EXACT) then multiply by (cols, rows).
    const LinTab tx = make_lintab(mesh_cols, cols, false), ty = make_lintab(mesh_rows, rows, true);
    const float sw = (float)cols, sh = (float)rows;
    for (int y = 0; y < rows; y++)
    {
        const float* m0 = mesh + (size_t)ty.s0[y] * mesh_cols * 2;
        const float* m1 = mesh + (size_t)ty.s1[y] * mesh_cols * 2;
        const float b0 = ty.a0[y], b1 = ty.a1[y];
        for (int x = 0; x < cols; x++)
        {
            const int x0 = tx.s0[x], x1 = tx.s1[x];
            const float a0 = tx.a0[x], a1 = tx.a1[x];
            for (int ch = 0; ch < 2; ch++)
            {
                // HResizeLinear: D = S[sx]*alpha0 + S[sx+cn]*alpha1 (tail: S[sx]*1); VResizeLinear: S0*b0 + S1*b1
                const float h0 = (x1 == x0) ? m0[2 * x0 + ch] * 1.0f : m0[2 * x0 + ch] * a0 + m0[2 * x1 + ch] * a1;
                const float h1 = (x1 == x0) ? m1[2 * x0 + ch] * 1.0f : m1[2 * x0 + ch] * a0 + m1[2 * x1 + ch] * a1;
                const float v = h0 * b0 + h1 * b1;
                map[((size_t)y * cols + x) * 2 + ch] = v * (ch == 0 ? sw : sh);
            }
        }
    }
}

static int remap_mesh_impl(const uint8_t* src, int src_step, int src_rows, int src_cols,
                    uint8_t* dst, int dst_step,
                    const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3],
                    int yuv, int nthreads, const float* L)
{
    if (!src || !dst || !mesh || mesh_rows < 2 || mesh_cols < 2) return -1;
    const LinTab tx = make_lintab(mesh_cols, src_cols, false), ty = make_lintab(mesh_rows, src_rows, true);
    const float sw = (float)src_cols, sh = (float)src_rows;
    parallel_rows(src_rows, nthreads, [&, sw, sh](int r0, int r1) {
        for (int y = r0; y < r1; y++)
        {
            const float* m0 = mesh + (size_t)ty.s0[y] * mesh_cols * 2;
            const float* m1 = mesh + (size_t)ty.s1[y] * mesh_cols * 2;
            const float b0 = ty.a0[y], b1 = ty.a1[y];
            uint8_t* drow = dst + (size_t)y * dst_step;
            for (int x = 0; x < src_cols; x++)
            {
                const int x0 = tx.s0[x], x1 = tx.s1[x];
                const float a0 = tx.a0[x], a1 = tx.a1[x];
                float off[2];
                for (int ch = 0; ch < 2; ch++)
                {
                    const float h0 = (x1 == x0) ? m0[2 * x0 + ch] * 1.0f : m0[2 * x0 + ch] * a0 + m0[2 * x1 + ch] * a1;
                    const float h1 = (x1 == x0) ? m1[2 * x0 + ch] * 1.0f : m1[2 * x0 + ch] * a0 + m1[2 * x1 + ch] * a1;
                    off[ch] = (h0 * b0 + h1 * b1) * (ch == 0 ? sw : sh);
                }
                // FSR.cl:381 (dst_bounds.xy == 0: the map is never an ROI on this path)
                const float subx = (float)x + off[0];
                const float suby = (float)y + off[1];
                remap_pixel_lens(src, src_step, src_rows, src_cols, drow + 3 * x, subx, suby, bg, yuv != 0, L);
            }
        }
    });
    return 0;
}

int lvko_remap_mesh(const uint8_t* src, int src_step, int src_rows, int src_cols,
                    uint8_t* dst, int dst_step,
                    const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3],
                    int yuv, int nthreads)
{
    return remap_mesh_impl(src, src_step, src_rows, src_cols, dst, dst_step, mesh, mesh_rows, mesh_cols, bg, yuv, nthreads, nullptr);
}

int lvko_remap_map(const uint8_t* src, int src_step, int src_rows, int src_cols, uint8_t* dst, int dst_step,
                   const float* offsets, const uint8_t bg[3], int yuv, int nthreads)
{
    // FSR.cl:362-403 easu_remap with a materialised offset map (dst size == map size == src size, no ROI)
    if (!src || !dst || !offsets) return -1;
    parallel_rows(src_rows, nthreads, [=](int r0, int r1) {
        for (int y = r0; y < r1; y++)
            for (int x = 0; x < src_cols; x++)
            {
                const float* o = offsets + ((size_t)y * src_cols + x) * 2;
                remap_pixel(src, src_step, src_rows, src_cols, dst + (size_t)y * dst_step + 3 * x, (float)x + o[0], (float)y + o[1], bg, yuv != 0);
            }
    });
    return 0;
}

int lvko_get_perspective_transform(const float src[8], const float dst[8], double M[9])
{
    // OpenCV imgproc getPerspectiveTransform: rows i<4 [x y 1 0 0 0 -x*u -y*u | u], rows i+4 [0 0 0 x y 1 -x*v -y*v | v]
    double A[8][8], B[8];
    for (int i = 0; i < 4; i++)
    {
        const double x = src[2 * i], y = src[2 * i + 1], u = dst[2 * i], v = dst[2 * i + 1];
        A[i][0] = A[i + 4][3] = x;
        A[i][1] = A[i + 4][4] = y;
        A[i][2] = A[i + 4][5] = 1.0;
        A[i][3] = A[i][4] = A[i][5] = A[i + 4][0] = A[i + 4][1] = A[i + 4][2] = 0.0;
        A[i][6] = -x * u; A[i][7] = -y * u;
        A[i + 4][6] = -x * v; A[i + 4][7] = -y * v;
        B[i] = u; B[i + 4] = v;
    }
    // LU with partial pivoting (cv::solve DECOMP_LU), double.
    for (int i = 0; i < 8; i++)
    {
        int k = i;
        for (int j = i + 1; j < 8; j++) if (std::fabs(A[j][i]) > std::fabs(A[k][i])) k = j;
        if (std::fabs(A[k][i]) < 2.220446049250313e-16 * 100) { for (int q = 0; q < 9; q++) M[q] = (q % 4 == 0) ? 1.0 : 0.0; return -1; }
        if (k != i) { for (int j = i; j < 8; j++) std::swap(A[i][j], A[k][j]); std::swap(B[i], B[k]); }
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
    return 0;
}

int lvko_mesh2x2_to_homography(const float mesh[8], int rows, int cols, float H[9])
{
    // WarpMesh.cpp:185,197-214: destination corners, source = destination + offset * (cols, rows)
    // (Point2f * Scalar: float * double -> rounded back to float, Extensions.cpp operator*(Point2f, Scalar)).
    const float w = (float)cols, h = (float)rows;
    const float dstp[8] = { 0, 0, w, 0, 0, h, w, h };
    float srcp[8];
    for (int i = 0; i < 4; i++)
    {
        const float mx = (float)((double)mesh[2 * i] * (double)cols);
        const float my = (float)((double)mesh[2 * i + 1] * (double)rows);
        srcp[2 * i] = dstp[2 * i] + mx;
        srcp[2 * i + 1] = dstp[2 * i + 1] + my;
    }
    double M[9];
    const int rc = lvko_get_perspective_transform(dstp, srcp, M);   // maps destination -> source
    for (int q = 0; q < 9; q++) H[q] = (float)M[q];                 // Image.cpp:137-139
    return rc;
}

int lvko_warpmesh_apply(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dst, int dst_step,
                        const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3],
                        int yuv, int nthreads)
{
    if (mesh_rows == 2 && mesh_cols == 2)
    {
        float H[9];
        lvko_mesh2x2_to_homography(mesh, rows, cols, H);
        return lvko_remap_homography(src, src_step, rows, cols, dst, dst_step, rows, cols, 0, 0, H, bg, yuv, nthreads);
    }
    return lvko_remap_mesh(src, src_step, rows, cols, dst, dst_step, mesh, mesh_rows, mesh_cols, bg, yuv, nthreads);
}

// WarpMesh::apply with the lens pre-warp composed into the coordinate (fused mode; model from lvko_lens_model, NULL = plain)
int lvko_warpmesh_apply_lens(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dst, int dst_step,
                             const float* mesh, int mesh_rows, int mesh_cols, const uint8_t bg[3],
                             int yuv, int nthreads, const double* model)
{
    float L[17]; const float* Lp = nullptr;
    if (model)
    {
        L[0] = (float)(1.0 / model[0]); L[1] = (float)(1.0 / model[1]);
        for (int i = 2; i < 17; i++) L[i] = (float)model[i];
        Lp = L;
    }
    if (mesh_rows == 2 && mesh_cols == 2)
    {
        float H[9];
        lvko_mesh2x2_to_homography(mesh, rows, cols, H);
        return remap_homography_impl(src, src_step, rows, cols, dst, dst_step, rows, cols, 0, 0, H, bg, yuv, nthreads, Lp);
    }
    return remap_mesh_impl(src, src_step, rows, cols, dst, dst_step, mesh, mesh_rows, mesh_cols, bg, yuv, nthreads, Lp);
}

// lvk::upscale -> kernel easu_scale (Functions/Image.cpp:155-202, FSR.cl:324-358).  rscale = (float)src / (float)dst per axis;
// a destination pixel maps to dst_coord * rscale; where the 12-tap window leaves the image the NEAREST source pixel is copied
// (there is no background here).  size == src size is a plain copy (Image.cpp:162-166).
int lvko_upscale(const uint8_t* src, int src_step, int src_rows, int src_cols,
                 uint8_t* dst, int dst_step, int dst_rows, int dst_cols, int yuv, int nthreads)
{
    if (!src || !dst || src_rows <= 0 || src_cols <= 0) return -1;
    if (dst_cols < src_cols || dst_rows < src_rows) return -1;              // Image.cpp:157
    if (dst_cols == src_cols && dst_rows == src_rows)
    {
        for (int y = 0; y < src_rows; y++) std::memcpy(dst + (size_t)y * dst_step, src + (size_t)y * src_step, 3 * (size_t)src_cols);
        return 0;
    }
    const float rsx = (float)src_cols / (float)dst_cols, rsy = (float)src_rows / (float)dst_rows;   // Image.cpp:192-195
    const uint8_t bg[3] = {0, 0, 0};                                         // unreachable: the source coordinate is inside
    parallel_rows(dst_rows, nthreads, [=](int r0, int r1) {
        for (int y = r0; y < r1; y++)
        {
            uint8_t* drow = dst + (size_t)y * dst_step;
            for (int x = 0; x < dst_cols; x++)
                remap_pixel(src, src_step, src_rows, src_cols, drow + 3 * x, (float)x * rsx, (float)y * rsy, bg, yuv != 0);   // FSR.cl:334-356
        }
    });
    return 0;
}

// lvk::sharpen -> kernel rcas (Functions/Image.cpp:206-233, FSR.cl:460-535).  `sharpness` is the user value in [0, 1]; the kernel
// receives exp2(-2 (1 - sharpness)) (Image.cpp:227).  Arithmetic definition, as for EASU: `x * y + z` is one fmaf, min/max are
// fminf/fmaxf (a NaN operand loses: 0 * inf appears when a ring is all 0 or all 1).  The two limiter reciprocals differ in the
// compiled reference: LLVM folds the negation of `-hitMin` into the reciprocal, `min * (-1.0f / (4 mx))`, and that division has lost
// its 2.5 ulp licence -> a correctly rounded IEEE divide (v_div_scale / v_div_fmas / v_div_fixup in oracle/_ref); the hitMax
// reciprocal stays the device reciprocal native_rcp().
// convert_uchar3 truncates.  Out of place: the reference's ScalingFilter runs it in place (ScalingFilter.cpp:57), which races
// reads of neighbours against writes; the defined result is the one of distinct src and dst.  Border pixels are copied; the
// reference's `coord <= cols || coord <= rows` guard (FSR.cl:478) lets the padding threads of a work-group write past the row
// end, which is not reproduced.
namespace {
inline float APrxMedRcpF1(float a) { const float b = as_float(0x7ef19fffu - as_uint(a)); return b * fmaf(-b, a, 2.0f); }   // FSR.cl:70
inline float min4f(float a, float b, float c, float d) { return fminf(a, fminf(b, fminf(c, d))); }                     // FSR.cl:85
inline float max4f(float a, float b, float c, float d) { return fmaxf(a, fmaxf(b, fmaxf(c, d))); }                     // FSR.cl:84
}

int lvko_sharpen(const uint8_t* src, int src_step, int rows, int cols, uint8_t* dst, int dst_step, float sharpness, int nthreads)
{
    if (!src || !dst || rows <= 0 || cols <= 0 || !(sharpness >= 0.0f && sharpness <= 1.0f)) return -1;
    const float sharp = exp2f(-2.0f * (1.0f - sharpness));                  // Image.cpp:227
    parallel_rows(rows, nthreads, [=](int r0, int r1) {
        for (int y = r0; y < r1; y++)
        {
            const uint8_t* srow = src + (size_t)y * src_step;
            uint8_t* drow = dst + (size_t)y * dst_step;
            for (int x = 0; x < cols; x++)
            {
                const uint8_t* pe = srow + 3 * x;
                uint8_t* o = drow + 3 * x;
                if (x == 0 || x >= cols - 1 || y == 0 || y >= rows - 1) { o[0] = pe[0]; o[1] = pe[1]; o[2] = pe[2]; continue; }   // FSR.cl:475-481
                const F3 b = load_px(pe - src_step), h = load_px(pe + src_step), d = load_px(pe - 3), e = load_px(pe), f = load_px(pe + 3);
                const float bc[3] = {b.x, b.y, b.z}, hc[3] = {h.x, h.y, h.z}, dc[3] = {d.x, d.y, d.z}, ec[3] = {e.x, e.y, e.z}, fc[3] = {f.x, f.y, f.z};
                float lobe_c[3];
                for (int c = 0; c < 3; c++)                                 // FSR.cl:503-521
                {
                    const float mn4 = min4f(bc[c], dc[c], fc[c], hc[c]), mx4 = max4f(bc[c], dc[c], fc[c], hc[c]);
                    const float hitMin = fminf(mn4, ec[c]) * (1.0f / (4.0f * mx4));
                    const float hitMax = (1.0f - fmaxf(mx4, ec[c])) * native_rcp(fmaf(4.0f, mn4, -4.0f));
                    lobe_c[c] = fmaxf(-hitMin, hitMax);
                }
                // FSR.cl renames .z -> R, .y -> G, .x -> B and takes max(lobeR, max(lobeG, lobeB))
                const float lobe = fminf(fmaxf(fmaxf(lobe_c[2], fmaxf(lobe_c[1], lobe_c[0])), -0.1875f), 0.0f) * sharp;   // FSR.cl:525
                const float rcpL = APrxMedRcpF1(fmaf(4.0f, lobe, 1.0f));    // FSR.cl:528
                for (int c = 0; c < 3; c++)                                 // FSR.cl:529-531
                {
                    const float v = fmaf(((bc[c] + dc[c]) + hc[c]) + fc[c], lobe, ec[c]) * rcpL;
                    o[c] = (uint8_t)(int)(v * 255.0f);
                }
            }
        }
    });
    return 0;
}

// Installs (n == 1 << 23) or removes (tab == nullptr) the device reciprocal table: tab[m] = v_rcp_f32(as_float(0x3f800000 | m)).
int lvko_set_device_rcp_table(const float* tab, int n)
{
    if (tab == nullptr) { g_rcp_tab.clear(); return 0; }
    if (n != (1 << 23)) return -1;
    g_rcp_tab.assign(tab, tab + n);
    return 0;
}

} // extern "C"
