// Local motion estimate of the vector-field preset on the device.
//
// Replaces FrameTracker::estimate_local_motions (reference: LiveVisionKit/Vision/FrameTracker.cpp:200-321) with the constraint system of
// generate_mesh_constraints (:380-457).  The reference hands the sparse least-squares problem to Eigen::LeastSquaresConjugateGradient;
// this library solves the same problem directly through its normal equations N x = g:
//   static rows  -> a constant band matrix, built once per configuration on the host (host_logic.hpp, MeshSolverH::generate);
//   feature rows -> Q32 fixed-point sums accumulated with integer atomics (exact, order independent)          k_mesh_assemble
//   N = L D L^T  -> right-looking root-free band factorisation with reciprocal pivots, forward substitution carried along,
//                   every entry updated in pivot order by fused multiply-subtracts (binary64)                        k_mesh_solve, phase 1
//   L^T          -> column-oriented backward substitution (one wavefront, the rows in flight in registers)       k_mesh_solve, phase 2
//   inlier flags (L1 reprojection error through the feature's quad) and the normalised mesh offsets             k_mesh_solve, phase 3
// The previous solution (the reference's m_OptimizedMesh: warm start there, right-hand side of the temporal rows here) stays on the
// device.  n = 2 * cols * rows unknowns, half bandwidth hb = 2 * (3 * cols + 3) + 1 (512 and 103 for the 16 x 16 mesh of the preset).
//
// Phase 1 is one workgroup: the (hb + 1)-column window of the band that a pivot column touches lives in REGISTERS (4 columns x 13
// band offsets per thread, the slot of a column is its index modulo the window size), the pivot column and its scaled copy go through
// LDS.  An entry (column k, offset t) is touched by pivot p iff (k - p) + t <= hb: the scaled column is stored zero-padded, so the same
// straight-line update serves every thread and every step, no masks.
#include "lvk_hip_internal.hpp"
#include "host_logic.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

#ifndef LVK_MESH_TB
#define LVK_MESH_TB 8
#endif
#ifndef LVK_MESH_WIN_WAVES
#define LVK_MESH_WIN_WAVES 4
#endif
#ifndef LVK_MESH_FWD_WAVE
#define LVK_MESH_FWD_WAVE (LVK_MESH_WIN_WAVES + 1)
#endif
// k_mesh_solve's wavefronts by role: wavefront 0 walks the pivot chain, wavefront MS_FWD_WAVE carries the forward substitution and stores
// the columns of L, the others update the window.  (Wavefronts go to the CU's four SIMDs round robin: with three window wavefronts the two
// light roles share SIMD 0 and every window wavefront has a SIMD of its own.)
constexpr int MS_WIN_WAVES = LVK_MESH_WIN_WAVES, MS_FWD_WAVE = LVK_MESH_FWD_WAVE;
constexpr int MS_NT = 64 * (MS_WIN_WAVES + 2);
constexpr int MS_BULK = 64 * MS_WIN_WAVES;  // window threads
constexpr int MS_CA = 4, MS_TB = LVK_MESH_TB;   // register tile of a window thread: columns x band offsets
constexpr int MS_HB_MAX = 103;              // widest band phase 1 holds in registers (meshes up to 16 columns)
constexpr int MS_WP_MAX = 108;              // window columns: hb + 1 + (MS_CA - 1), rounded up to a multiple of MS_CA
constexpr int MS_PAD = 8;                   // zeros in front of the LDS columns (negative relative indices of the pivot's own group)
constexpr int MS_LCOL = MS_PAD + 2 * MS_WP_MAX + 2 * MS_TB + 8;
// LDS layouts.  All window wavefronts read their operands from the pivot column every step -- 15 values per thread, 23 KB per step
// through the CU's one LDS pipeline -- at addresses 4 m + MS_TB b + j (m: column group of the tile, b: its band).  In a plain array the
// 64 lanes of a read fall on a few banks (strides of 4 and 8 doubles over 32 double-wide banks): measured, every read took four passes
// and the LDS pipeline, not the arithmetic, set the pace of the factorisation.  So the pivot columns are stored with one spare slot
// after every four entries -- logical 4 y + r at 5 y + r: tiles with different (m + MS_TB / 4 b) hit different banks, equal ones the
// same address (a broadcast); MS_TB is a multiple of 4 so that r is a compile-time constant of every read.  The hand-over arrays are
// skewed the same way (one spare slot per band).
static_assert(MS_TB % 4 == 0 && MS_PAD % 4 == 0, "the padded LDS layout needs band boundaries at multiples of 4");
constexpr int ms_px(int x) { return 5 * ((x + 64) / 4 - 16) + (x + 64) % 4; }      // padded position of logical index x (x >= -64)
constexpr int MS_LCOL_P = ms_px(MS_LCOL) + 8;
constexpr int MS_COL_P = 128 + 128 / MS_TB + 1;                                     // col[]: logical t at t + t / MS_TB
constexpr int MS_NEXT_BAND = MS_CA * MS_TB + 1;                                     // next[]: one spare slot per band
constexpr double MS_Q = 4294967296.0;       // Q32
constexpr int MS_N_MAX = 2048;              // unknowns whose right-hand side fits the workgroup's LDS (16 x 64 vertices)
constexpr int MS_CHUNK = 16;                // rows of L staged per round of the backward substitution
constexpr int MS_BANDS = (MS_HB_MAX + MS_TB) / MS_TB;                      // bands of MS_TB band offsets
constexpr int MS_PF = (MS_BANDS * MS_CA * MS_TB + MS_BULK - 1) / MS_BULK;  // prefetched entries per window thread and column group
constexpr int ms_tiles(int hb) { int t = 0; for (int b = 0; b < (hb + MS_TB) / MS_TB; b++) t += (hb - MS_TB * b + 3) / MS_CA + 1; return t; }
static_assert(ms_tiles(MS_HB_MAX) <= MS_BULK, "one register tile per window thread");
static_assert(MS_BANDS * MS_TB <= 128 && MS_HB_MAX < 128, "a column is two registers per lane of the chain");

struct MeshArgs
{
    int cols, rows, n, hb, nbands;          // nbands: bands of MS_TB band offsets covering 0 .. hb
    const double* stat;                     // static band, column layout: entry (i, k), k <= i <= k + hb, at [k * (hb + 1) + (i - k)]
    long long* Nq; long long* gq;           // Q32 sums of the feature rows (same layout as stat / one per unknown)
    double* N; double* g0;                  // the assembled system (k_mesh_prepare)
    float* mesh;                            // previous solution (absolute tracking-frame coordinates), updated on success
    double* Lc;                             // columns of L: L(i, k) at [k * (hb + 1) + (i - k)]
    int* fidx; float* fw;                   // per feature: the 4 unknown indices (x components) and barycentric weights
    const float2* p1; const float2* p2;     // tracked / matched points
    const int* count; int n_pts;            // number of pairs: *count when count != nullptr (decided on the GPU), else n_pts
    int min_samples;
    float region_w, region_h, ts_gen, ts_now, threshold;
    int* flags;                             // device: bit 0 = a feature fell outside the mesh
    float* out_offsets; uint8_t* out_mask; int* out_status;      // device-visible host memory
};

__device__ __forceinline__ int pair_count(const MeshArgs& a) { return a.count ? min(*a.count, a.n_pts) : a.n_pts; }

// FrameTracker.cpp:233-262: the feature's cell, its barycentric weights, and its rows' contribution to N and g
__global__ __launch_bounds__(128)
void k_mesh_assemble(MeshArgs a)
{
    LVK_TRACKER_PRIORITY();
    const int m = pair_count(a);
    if (m < a.min_samples) return;
    const int f = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (f >= m) return;
    const int ld = a.hb + 1, W = a.cols;
    const float kw = (((float)a.cols / (float)(a.cols - 1)) * a.region_w) / (float)a.cols;
    const float kh = (((float)a.rows / (float)(a.rows - 1)) * a.region_h) / (float)a.rows;
    const float px = a.p1[f].x, py = a.p1[f].y;
    int kx = (int)(px / kw), ky = (int)(py / kh);                       // VirtualGrid::key_of, then the clamp of :243-244
    kx = min(max(kx, 0), a.cols - 1); ky = min(max(ky, 0), a.rows - 1);
    const int i00 = 2 * (ky * W + kx), i11 = 2 * ((ky + 1) * W + kx + 1);
    if (i11 + 1 >= a.n) { atomicOr(a.flags, 1); return; }               // last cell row / column: would index past the mesh
    const int id[4] = {i00, i11 - 2, i11, i00 + 2};                     // TL, BL, BR, TR
    const float x1 = (float)kx * kw, y1 = (float)ky * kh;
    const float cw = (float)(kx + 1) * kw - x1, chh = (float)(ky + 1) * kh - y1;
    const float inv = 1.0f / (cw * chh);
    const float rx1 = (x1 + cw) - px, ry1 = (y1 + chh) - py, rx2 = px - x1, ry2 = py - y1;
    const float wgt[4] = {rx1 * ry1 * inv, rx1 * ry2 * inv, rx2 * ry2 * inv, rx2 * ry1 * inv};
#pragma unroll
    for (int q = 0; q < 4; q++) { a.fidx[4 * f + q] = id[q]; a.fw[4 * f + q] = wgt[q]; }
    const float tgt[2] = {a.p2[f].x, a.p2[f].y};
#pragma unroll
    for (int comp = 0; comp < 2; comp++)
#pragma unroll
        for (int p = 0; p < 4; p++)
        {
            const int ia = id[p] + comp;
            atomicAdd((unsigned long long*)&a.gq[ia], (unsigned long long)__double2ll_rn((double)wgt[p] * (double)tgt[comp] * MS_Q));
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                const int ib = id[q] + comp;
                if (ia >= ib)
                    atomicAdd((unsigned long long*)&a.Nq[(size_t)ib * ld + (ia - ib)], (unsigned long long)__double2ll_rn((double)wgt[p] * (double)wgt[q] * MS_Q));
            }
        }
}

// N and g in binary64: static part + Q32 sums, ridge on the diagonal (S4 of the specification); the Q32 accumulators are cleared for the
// next frame on the way.
__global__ __launch_bounds__(256)
void k_mesh_prepare(MeshArgs a)
{
    LVK_TRACKER_PRIORITY();
    const int ld = a.hb + 1;
    const size_t band = (size_t)a.n * ld;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t at = (size_t)blockIdx.x * blockDim.x + threadIdx.x; at < band + a.n; at += stride)
    {
        if (at < band)
        {
            const long long q = a.Nq[at]; a.Nq[at] = 0;
            double v = a.stat[at] + (double)q / MS_Q;
            if (at % ld == 0) v = v + 1e-6;
            a.N[at] = v;
        }
        else
        {
            const size_t i = at - band;
            const long long q = a.gq[i]; a.gq[i] = 0;
            a.g0[i] = (double)a.ts_gen * (double)(a.ts_now * a.mesh[i]) + (double)q / MS_Q;
        }
    }
}

// entry (k + t, k) of N; 0 outside the matrix
__device__ __forceinline__ double load_entry(const MeshArgs& a, int k, int t)
{
    if (k >= a.n || t > a.hb || k + t >= a.n) return 0.0;
    return a.N[(size_t)k * (a.hb + 1) + t];
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is also a release fence for GLOBAL memory: it drains the workgroup's
// outstanding global stores (the rows of L written every step) before every barrier -- measured 1.9 us per elimination step.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#if defined(LVK_MESH_TIMING) && LVK_MESH_TIMING > 1
__device__ long long g_mesh_phase[8];
#endif

struct FactorShared
{
    // pivot column p, by step parity: raw[MS_PAD + x] = N(p + x, p) after all earlier pivots (the entries x = 0, 1 -- the pivot itself and
    // the row of column p + 1, which is the chain's -- are stored as zeros: to the window threads those columns are spent), lcol = the
    // column times the reciprocal pivot (column p of L, entry 0 stored as zero); zeros outside 0 .. hb: the updates rely on the padding
    // instead of masks.  rinv: the reciprocal pivot.
    double raw[2][MS_LCOL_P], lcol[2][MS_LCOL_P];     // padded layout: logical index x at ms_px(x)
    double rinv[2];
    double col[2][MS_COL_P];                   // column p + 2 as the window threads leave it after pivot p (read by the chain one step later)
    double w[MS_N_MAX + 128];               // right-hand side: g, then D^-1 L^-1 g and finally the solution, in place (+ slack: rows beyond n)
    double next[MS_BANDS][MS_NEXT_BAND];    // phase 1: per band, the columns that enter its window when the current group is done
    double lt[2][MS_CHUNK][MS_HB_MAX + 1];  // phase 2: rows of L, staged chunk by chunk
    int fail;
};

// ---- phase 1, look-ahead organisation -----------------------------------------------------------------------------------------------
// The steps of the elimination are latency bound: a single wavefront issues an instruction every 5-7 cycles, whatever it is, so the time
// of a step is the longest instruction stream between two barriers plus the dependent latencies on it (a trip through LDS, a division).
// Three roles run side by side, one step apart, with ONE barrier per step:
//   * the chain (one wavefront) owns the NEXT pivot column.  During step p it takes column p + 1 as the window threads left it after
//     pivot p - 1 (col[]), applies pivot p to it, forms the reciprocal pivot and the column of L and publishes them (parity p + 1).
//     Nothing else: ~45 instructions with one LDS read, one division and one LDS write on the dependent path.
//   * the window (MS_WIN_WAVES wavefronts) applies pivot p (published during step p - 1) to its 4 x MS_TB register tiles; the column
//     p + 2 goes first and is handed to the chain through col[].
//   * the forward substitution (one wavefront) applies pivot p to the right-hand side, scales its row p by the reciprocal pivot and
//     stores column p of L for phase 2.
// Every entry still receives its updates in pivot order with the same operands: bit-identical to the plain loop of the specification.
__device__ __forceinline__ double readlane64(double v, int src);

// (All parts are written without per-entry conditions: the LDS arrays are padded with zeros, entries outside the band are zeros that
//  stay zeros, and every store is unconditional.  The workgroup's wavefronts share ONE scalar unit: exec-mask bookkeeping for
//  per-entry branches made the first version of this loop scalar-issue bound -- 125 scalar instructions per step and wavefront.)

// the chain's part of step p = p0 + CK (p0 a multiple of MS_CA): finishes column p + 1 and turns it into pivot data of parity (p + 1).
// r1 = N(p + 1, p), the unscaled entry of the pivot column p in the row of column p + 1 (wave uniform); bad: a pivot was not positive.
template <int CK>
__device__ __forceinline__ void chain_step(const MeshArgs& a, FactorShared& s, int p, double& r1, bool& bad)
{
    const int lane = (int)threadIdx.x, q = p + 1;
    if (q >= a.n) return;
    const double* lcol = s.lcol[CK & 1];
    const int x1 = ms_px(MS_PAD + 1 + lane), x0 = ms_px(MS_PAD + lane), tc = lane + lane / MS_TB, tc2 = (lane + 64) + (lane + 64) / MS_TB;
    const double c0 = __builtin_fma(-lcol[x1], r1, s.col[CK & 1][tc]);                 // band offsets t = lane and t = lane + 64 of
    const double c1 = __builtin_fma(-lcol[x1 + 80], r1, s.col[CK & 1][tc2]);           // column q; beyond hb everything is zero
    const double d = readlane64(c0, 0);
    r1 = readlane64(c0, 1);
    bad = bad || !(d > 0.0);
    const double r = 1.0 / d;
    double* raw_n = s.raw[(CK + 1) & 1]; double* lcol_n = s.lcol[(CK + 1) & 1];
    raw_n[x0] = lane <= 1 ? 0.0 : c0;
    raw_n[x0 + 80] = c1;                                                                // ms_px(x + 64) = ms_px(x) + 80
    lcol_n[x0] = lane == 0 ? 0.0 : c0 * r;
    lcol_n[x0 + 80] = c1 * r;
    if (lane == 0) s.rinv[(CK + 1) & 1] = r;
}

// the forward substitution's part of step p: z(p + t) -= L(p + t, p) z(p) for the rows of the band, row p becomes z(p) / d(p); column p
// of L goes to global memory for phase 2.  (The wavefront's LDS accesses execute in order: the rows written here are read back by the
// same wavefront one step later without a barrier.)
template <int CK>
__device__ __forceinline__ void forward_step(const MeshArgs& a, FactorShared& s, int p)
{
    const int lane = (int)threadIdx.x & 63, hb = a.hb;
    const double* lcol = s.lcol[CK & 1];
    const int x0 = ms_px(MS_PAD + lane);
    const double l0 = lcol[x0], l1 = lcol[x0 + 80];                                                  // l0 of lane 0 is stored as zero
    const double wq = s.w[p], w0 = s.w[p + lane], w1 = s.w[p + 64 + lane], r = s.rinv[CK & 1];
    const double n0 = __builtin_fma(-l0, wq, w0);
    s.w[p + lane] = lane == 0 ? wq * r : n0;                                                        // rows beyond the matrix keep their zeros (l = 0)
    s.w[p + 64 + lane] = __builtin_fma(-l1, wq, w1);
    double* Lp = a.Lc + (size_t)p * (hb + 1);
    if (lane <= hb) Lp[lane] = l0;
    if (lane + 64 <= hb) Lp[lane + 64] = l1;
}

// The window.  An entry (column k, band offset t) is touched by pivot p iff (k - p) + t <= hb: a column at distance s from the pivot
// needs only its offsets t <= hb - s.  The offsets are therefore split into bands of MS_TB, and every band keeps its OWN window of
// columns: band b (offsets from T = MS_TB b) holds nb(b) = (hb - T + 3) / 4 + 1 column groups -- 27 for the first band, 2 for the last,
// 177 tiles of 4 x 9 in all instead of the 324 of one common window (of which half would hold entries no pivot reaches yet).  A tile's
// thread follows the rotation of its band: m = distance (in groups) of its column group from the pivot's, counting down; when its
// group has been pivoted it takes over the group that enters the band's window, nb(b) groups on.
__host__ __device__ __forceinline__ int band_groups(int hb, int b) { return (hb - MS_TB * b + 3) / MS_CA + 1; }

// entries of the column groups that enter the bands' windows when the group at p0 has been pivoted: fetch (N is read once, in band
// order) and hand-over to the tiles through LDS
__device__ __forceinline__ void window_fetch(const MeshArgs& a, double (&pf)[MS_PF], int p0, int btid)
{
    const int hb = a.hb, ld = hb + 1;
#pragma unroll
    for (int q = 0; q < MS_PF; q++)
    {
        const int idx = btid + MS_BULK * q, pb = idx / (MS_CA * MS_TB), rem = idx - pb * (MS_CA * MS_TB), c = rem / MS_TB, ti = rem - c * MS_TB;
        const int k = p0 + MS_CA * band_groups(hb, pb) + c, t = MS_TB * pb + ti;
        const bool in = pb < a.nbands && t <= hb && k + t < a.n;
        pf[q] = a.N[in ? (size_t)k * ld + t : 0];
    }
}
__device__ __forceinline__ void window_park(const MeshArgs& a, FactorShared& s, const double (&pf)[MS_PF], int p0, int btid)
{
    const int hb = a.hb;
#pragma unroll
    for (int q = 0; q < MS_PF; q++)
    {
        const int idx = btid + MS_BULK * q, pb = idx / (MS_CA * MS_TB), rem = idx - pb * (MS_CA * MS_TB), c = rem / MS_TB, ti = rem - c * MS_TB;
        const int k = p0 + MS_CA * band_groups(hb, pb) + c, t = MS_TB * pb + ti;
        const bool in = pb < a.nbands && t <= hb && k + t < a.n;
        if (idx < MS_BANDS * MS_CA * MS_TB) s.next[pb][rem] = in ? pf[q] : 0.0;
    }
}

// the window's part of step p, pivot at position CK of its column group
template <int CK>
__device__ __forceinline__ void window_step(const MeshArgs& a, FactorShared& s, double (&A)[MS_CA][MS_TB], double (&pf)[MS_PF],
                                            int p, int& m, int nb, int band, bool valid, int btid)
{
    const int t0 = band * MS_TB;
    const double* raw = s.raw[CK & 1]; const double* lcol = s.lcol[CK & 1];
    // the columns that take over the slots of a pivoted group in each band's window travel one group ahead: what was fetched a group ago
    // is parked in LDS now (read at the end of this group), and the fetch for the next group is issued -- four steps of distance, no
    // step ever waits for global memory (fetch and use in the same group cost a memory latency every fourth step: 800 cycles per step
    // on average, more than the arithmetic)
#ifndef LVK_MESH_DBG_NOFETCH
    if (CK == 0) { window_park(a, s, pf, p, btid); window_fetch(a, pf, p + MS_CA, btid); }
#endif
    if (valid)
    {
        const int s0 = MS_CA * m - CK;                                  // column distance of the tile's first column from the pivot
        const int yr = 5 * m, yl = 5 * (m + (MS_TB / 4) * band);        // padded positions of MS_PAD + 4 m and MS_PAD + 4 m + t0, less ms_px(MS_PAD)
        double rc[MS_CA], lw[MS_CA + MS_TB - 1];
#pragma unroll
        for (int ck = 0; ck < MS_CA; ck++) rc[ck] = raw[yr + ms_px(MS_PAD + ck - CK)];      // zero for spent columns, the pivot and column p + 1 (the chain's)
#pragma unroll
        for (int j = 0; j < MS_CA + MS_TB - 1; j++) lw[j] = lcol[yl + ms_px(MS_PAD + j - CK)];
#if defined(LVK_MESH_DBG_READS)
        for (int rr = 1; rr < LVK_MESH_DBG_READS; rr++)
        {
            int off = 1024 * (rr & 1);                                                              // other addresses, same bank pattern
            asm volatile("" : "+v"(off));
            const double* vl = lcol + off; const double* vr = raw + off;
            double t0v[MS_CA], t1v[MS_CA + MS_TB - 1];
#pragma unroll
            for (int ck = 0; ck < MS_CA; ck++) t0v[ck] = vr[yr + ms_px(MS_PAD + ck - CK)];
#pragma unroll
            for (int j = 0; j < MS_CA + MS_TB - 1; j++) t1v[j] = vl[yl + ms_px(MS_PAD + j - CK)];
#pragma unroll
            for (int ck = 0; ck < MS_CA; ck++) asm volatile("" :: "v"(t0v[ck]));
#pragma unroll
            for (int j = 0; j < MS_CA + MS_TB - 1; j++) asm volatile("" :: "v"(t1v[j]));
        }
#endif
#if defined(LVK_MESH_DBG_FMAS)
        for (int rr = 1; rr < LVK_MESH_DBG_FMAS; rr++)
#pragma unroll
            for (int ck = 0; ck < MS_CA; ck++)
#pragma unroll
                for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = __builtin_fma(-lw[ck + ti], rc[ck], A[ck][ti]);
#endif
        // column p + 2 first: it is the chain's input of the next step
        constexpr int NK = (CK + 2) % MS_CA;
#pragma unroll
        for (int ti = 0; ti < MS_TB; ti++) A[NK][ti] = __builtin_fma(-lw[NK + ti], rc[NK], A[NK][ti]);
#ifndef LVK_MESH_DBG_NOHANDOFF
        if (s0 + NK == 2)
#else
        if (s0 + NK == 2 && p < 0)
#endif
#pragma unroll
            for (int ti = 0; ti < MS_TB; ti++) s.col[(CK + 1) & 1][t0 + band + ti] = A[NK][ti];
#pragma unroll
        for (int ck = 0; ck < MS_CA; ck++)
            if (ck != NK)
#pragma unroll
                for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = __builtin_fma(-lw[ck + ti], rc[ck], A[ck][ti]);
        if (CK == MS_CA - 1)
        {
#ifndef LVK_MESH_DBG_NOTAKEOVER
            if (m == 0)
#else
            if (m == 0 && p < 0)
#endif
#pragma unroll
                for (int ck = 0; ck < MS_CA; ck++)
#pragma unroll
                    for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = s.next[band][ck * MS_TB + ti];
            m = m == 0 ? nb - 1 : m - 1;
        }
    }
}

// wave-uniform value of lane `src` of a binary64 register
__device__ __forceinline__ double readlane64(double v, int src)
{
    const long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(u >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__global__ __launch_bounds__(MS_NT)
void k_mesh_solve(MeshArgs a)
{
    LVK_TRACKER_PRIORITY();
    __shared__ FactorShared s;
    const int tid = (int)threadIdx.x;
    const int n = a.n, hb = a.hb, ld = hb + 1;
    const int m = pair_count(a);
    const int flags = *a.flags;
    __syncthreads();
    if (tid == 0) *a.flags = 0;                                         // cleared for the next frame
    if (m < a.min_samples) { if (tid == 0) *a.out_status = 1; return; }
    if (flags & 1) { if (tid == 0) *a.out_status = 2; return; }
#ifdef LVK_MESH_TIMING
    const long long tm0 = wall_clock64();
#endif

    // ---- phase 1: N = L D L^T and w = D^-1 L^-1 g
    const int wave = tid >> 6;
    const bool chain = wave == 0, forward = wave == MS_FWD_WAVE;
    const int btid = (wave - 1 - (wave > MS_FWD_WAVE ? 1 : 0)) * 64 + (tid & 63);      // index among the window threads
    // window thread -> (band, slot), slot-major: neighbouring lanes hold neighbouring BANDS of one column group.  Their operand reads are
    // then MS_TB doubles apart (2-way bank conflicts at worst); neighbouring groups of one band are 4 doubles apart -- every fourth lane on
    // the same LDS banks, a 16-way conflict that made the loop twice as slow.
    int band = a.nbands, slot = 0, nb = 1;
    if (!chain && !forward)
    {
        int rem = btid;
        for (int j = 0; j < band_groups(hb, 0); j++)
        {
            const int cnt = min(a.nbands, (hb + 3 - MS_CA * j) / MS_TB + 1);     // bands that have a slot j: band_groups(hb, b) > j
            if (rem < cnt) { band = rem; slot = j; break; }
            rem -= cnt;
        }
        nb = band_groups(hb, min(band, a.nbands - 1));
    }
    const bool valid = !chain && !forward && band < a.nbands;
    int gdist = slot;                                                   // distance (in column groups) of this tile's group from the pivot's
    if (chain) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);      // the chain is the critical path
    double A[MS_CA][MS_TB], pf[MS_PF];
#pragma unroll
    for (int ck = 0; ck < MS_CA; ck++)
#pragma unroll
        for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = valid ? load_entry(a, MS_CA * slot + ck, band * MS_TB + ti) : 0.0;
#pragma unroll
    for (int q = 0; q < MS_PF; q++) pf[q] = 0.0;
    if (!chain && !forward) window_fetch(a, pf, 0, btid);              // parked at step 0, taken over at the end of the first group
    for (int i = tid; i < 2 * MS_LCOL_P; i += MS_NT) { (&s.raw[0][0])[i] = 0.0; (&s.lcol[0][0])[i] = 0.0; }
    for (int i = tid; i < 2 * MS_COL_P; i += MS_NT) (&s.col[0][0])[i] = 0.0;
    for (int i = tid; i < MS_BANDS * MS_NEXT_BAND; i += MS_NT) (&s.next[0][0])[i] = 0.0;
    for (int i = tid; i < n + 128; i += MS_NT) s.w[i] = i < n ? a.g0[i] : 0.0;
    if (tid == 0) s.fail = 0;
    __syncthreads();
    // start-up: pivot 0 straight from N (parity 0), column 1 as the chain's first input
    double r1 = 0.0; bool bad = false;
    if (chain)
    {
        const double d0 = a.N[0];
        bad = !(d0 > 0.0);
        const double r = 1.0 / d0;
        r1 = n > 1 ? load_entry(a, 0, 1) : 0.0;
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int t = tid + 64 * h;
            const double c = load_entry(a, 0, t);
            if (t <= hb) { s.raw[0][ms_px(MS_PAD + t)] = t <= 1 ? 0.0 : c; s.lcol[0][ms_px(MS_PAD + t)] = t == 0 ? 0.0 : c * r; s.col[0][t + t / MS_TB] = load_entry(a, 1, t); }
        }
        if (tid == 0) s.rinv[0] = r;
    }
    __syncthreads();
#ifdef LVK_MESH_TIMING
    const long long tm1 = wall_clock64();
#endif
#if defined(LVK_MESH_TIMING) && LVK_MESH_TIMING > 1
    long long pt0 = 0, pt1 = 0;
    const int pslot = tid == 0 ? 0 : (tid == 64 * MS_FWD_WAVE ? 2 : (tid == 64 * (MS_FWD_WAVE == 1 ? 2 : 1) ? 4 : -1));
#define MESH_PROBE_BEGIN() pt0 = (long long)__builtin_readcyclecounter()
#define MESH_PROBE_MID() do { pt1 = (long long)__builtin_readcyclecounter(); if (pslot >= 0) g_mesh_phase[pslot] += pt1 - pt0; } while (0)
#define MESH_PROBE_END() do { pt0 = (long long)__builtin_readcyclecounter(); if (pslot >= 0) g_mesh_phase[pslot + 1] += pt0 - pt1; } while (0)
#else
#define MESH_PROBE_BEGIN() do { } while (0)
#define MESH_PROBE_MID() do { } while (0)
#define MESH_PROBE_END() do { } while (0)
#endif
    for (int p0 = 0; p0 < n; p0 += MS_CA)
    {
#define LVK_MESH_STEP(CKV)                                                                      \
        if (p0 + CKV < n)                                                                        \
        {                                                                                        \
            MESH_PROBE_BEGIN();                                                                  \
            if (chain) chain_step<CKV>(a, s, p0 + CKV, r1, bad);                                 \
            else if (forward) forward_step<CKV>(a, s, p0 + CKV);                                 \
            else window_step<CKV>(a, s, A, pf, p0 + CKV, gdist, nb, band, valid, btid);          \
            MESH_PROBE_MID();                                                                    \
            lds_barrier();                                                                       \
            MESH_PROBE_END();                                                                    \
        }
        LVK_MESH_STEP(0) LVK_MESH_STEP(1) LVK_MESH_STEP(2) LVK_MESH_STEP(3)
#undef LVK_MESH_STEP
    }
    if (tid == 0 && bad) s.fail = 1;
    __syncthreads();
#ifndef LVK_MESH_TIMING
    if (s.fail != 0) { if (tid == 0) *a.out_status = 3; return; }
#endif
#ifdef LVK_MESH_TIMING
    const long long tm2 = wall_clock64();
#endif

    // ---- phase 2: L^T x = w column by column: x(j) = w(j); w(k) -= L(j, k) x(j) for the rows k above j in the band.
    // One wavefront walks the chain with the rows in flight in registers; all threads stage the rows of L it needs, one chunk ahead
    // (the loads of chunk c + 1 are in flight while the chain runs over chunk c).  L is stored by columns: a chunk of rows is a set of
    // short contiguous column segments.
    const int nchunks = (n + MS_CHUNK - 1) / MS_CHUNK;
    constexpr int PER = ((MS_HB_MAX + MS_CHUNK) * MS_CHUNK + MS_NT - 1) / MS_NT;
    double stage_a[PER], stage_b[PER];
    auto fetch = [&](int c, double (&stage)[PER]) {
        const int ilo = n - (c + 1) * MS_CHUNK;                          // rows ilo .. ilo + MS_CHUNK - 1 (the top chunk may start below 0)
#pragma unroll
        for (int q = 0; q < PER; q++)
        {
            const int idx = tid + MS_NT * q, kk = idx / MS_CHUNK, rr = idx - kk * MS_CHUNK, i = ilo + rr, k = ilo - hb + kk, t = i - k;
            const bool in = kk < hb + MS_CHUNK && i >= 0 && k >= 0 && t >= 1 && t <= hb;
            stage[q] = a.Lc[in ? (size_t)k * ld + t : 0];
            if (!in) stage[q] = 0.0;
        }
    };
    auto commit = [&](int c, const double (&stage)[PER]) {
        const int ilo = n - (c + 1) * MS_CHUNK;
#pragma unroll
        for (int q = 0; q < PER; q++)
        {
            const int idx = tid + MS_NT * q, kk = idx / MS_CHUNK, rr = idx - kk * MS_CHUNK, t = (ilo + rr) - (ilo - hb + kk);
            if (kk < hb + MS_CHUNK && t >= 1 && t <= hb) s.lt[c & 1][MS_CHUNK - 1 - rr][t] = stage[q];      // row index counted from the chunk's top row
        }
    };
    // chunk c is consumed from LDS while chunk c + 1 waits in registers and chunk c + 2 is in flight (an L2 round trip is several chain chunks long)
    for (int i = tid; i < 2 * MS_CHUNK; i += MS_NT) (&s.lt[0][0][0])[(size_t)i * (MS_HB_MAX + 1)] = 0.0;     // entry 0 of every staged row: the zero the chain reads for offsets outside the band
    fetch(0, stage_a); commit(0, stage_a);
    if (nchunks > 1) fetch(1, stage_a);
    __syncthreads();
    // rows of the 64-row blocks B, B - 1, B - 2 (B = the block of row n - 1), one per lane of wavefront 0
    const int nblocks = (n + 63) / 64;
    auto block_rows = [&](int b) -> double { const int i = 64 * b + tid; return (b >= 0 && i < n) ? s.w[i] : 0.0; };
    double cur = 0.0, p1 = 0.0, p2 = 0.0;
    if (tid < 64) { cur = block_rows(nblocks - 1); p1 = block_rows(nblocks - 2); p2 = block_rows(nblocks - 3); }
    int B = nblocks - 1;
    auto chain_chunk = [&](int c) {
        if (tid < 64)
        {
            const int top = n - 1 - c * MS_CHUNK;
            const int rows_here = min(MS_CHUNK, top + 1);
            // The three entries of L this lane needs for a row do not depend on the chain: they are read one row ahead.  An offset
            // outside 1 .. hb reads the row's entry 0, which is zero: the updates below need no masks (the single wavefront of the
            // chain issues one instruction every ~5 cycles: the row time is its instruction count).
            auto entries = [&](int r, double& l0, double& l1, double& l2) {
                const int tc = ((top - r) & 63) - tid;
                const double* row = s.lt[c & 1][r];
                l0 = row[(unsigned)(tc - 1) < (unsigned)hb ? tc : 0];
                l1 = row[tc + 64 <= hb ? tc + 64 : 0];
                l2 = row[tc + 128 <= hb ? tc + 128 : 0];
            };
            double n0, n1, n2;
            entries(0, n0, n1, n2);
            for (int r = 0; r < rows_here; r++)
            {
                const int lj = (top - r) & 63;
                const double l0 = n0, l1 = n1, l2 = n2;
                if (r + 1 < rows_here) entries(r + 1, n0, n1, n2);
                const double xj = readlane64(cur, lj);
                cur = __builtin_fma(-l0, xj, cur);
                p1 = __builtin_fma(-l1, xj, p1);
                p2 = __builtin_fma(-l2, xj, p2);
                if (lj == 0)
                {
                    // block B is final: x of its rows; the registers move up one block
                    if (64 * B + tid < n) s.w[64 * B + tid] = cur;
                    cur = p1; p1 = p2; p2 = block_rows(B - 3);
                    B--;
                }
            }
        }
    };
    for (int c = 0; c < nchunks; c += 2)
    {
        // even chunk: chunk c + 1 sits in stage_a, chunk c + 2 goes to stage_b
        if (c + 2 < nchunks) fetch(c + 2, stage_b);
        chain_chunk(c);
        if (c + 1 < nchunks) commit(c + 1, stage_a);
        __syncthreads();
        if (c + 1 >= nchunks) break;
        if (c + 3 < nchunks) fetch(c + 3, stage_a);
        chain_chunk(c + 1);
        if (c + 2 < nchunks) commit(c + 2, stage_b);
        __syncthreads();
    }
#ifdef LVK_MESH_TIMING
    const long long tm3 = wall_clock64();
    if (tid == 0)
    {
        printf("mesh solve: init %lld, factor %lld, backsolve %lld (100 MHz ticks), n %d hb %d\n", tm1 - tm0, tm2 - tm1, tm3 - tm2, n, hb);
#if LVK_MESH_TIMING > 1
        printf("  per step (shader cycles): chain work %lld wait %lld | forward work %lld wait %lld | window work %lld wait %lld\n",
               g_mesh_phase[0] / n, g_mesh_phase[1] / n, g_mesh_phase[2] / n, g_mesh_phase[3] / n, g_mesh_phase[4] / n, g_mesh_phase[5] / n);
        for (int k = 0; k < 8; k++) g_mesh_phase[k] = 0;
#endif
    }
#endif

    // ---- phase 3: the solution as float, inlier flags, offsets (FrameTracker.cpp:276-320)
    for (int i = tid; i < n; i += MS_NT) a.mesh[i] = (float)s.w[i];
    __syncthreads();
    for (int f = tid; f < m; f += MS_NT)
    {
        const int* id = a.fidx + 4 * f; const float* wq = a.fw + 4 * f;
        const float x = wq[0] * a.mesh[id[0]] + wq[1] * a.mesh[id[1]] + wq[2] * a.mesh[id[2]] + wq[3] * a.mesh[id[3]];
        const float y = wq[0] * a.mesh[id[0] + 1] + wq[1] * a.mesh[id[1] + 1] + wq[2] * a.mesh[id[2] + 1] + wq[3] * a.mesh[id[3] + 1];
        a.out_mask[f] = (fabsf(x - a.p2[f].x) + fabsf(y - a.p2[f].y)) < a.threshold ? 1 : 0;
    }
    const float kw = (((float)a.cols / (float)(a.cols - 1)) * a.region_w) / (float)a.cols;
    const float kh = (((float)a.rows / (float)(a.rows - 1)) * a.region_h) / (float)a.rows;
    for (int v = tid; v < a.cols * a.rows; v += MS_NT)
    {
        const int r = v / a.cols, c = v - r * a.cols;
        a.out_offsets[2 * v] = ((float)c * kw - a.mesh[2 * v]) / a.region_w;
        a.out_offsets[2 * v + 1] = ((float)r * kh - a.mesh[2 * v + 1]) / a.region_h;
    }
    __syncthreads();
    if (tid == 0) { __threadfence_system(); *a.out_status = 0; }
}

} // namespace

// ---- host side ------------------------------------------------------------------------------------------------------------------
struct lvk_mesh_solver_dev
{
    lvk_hip_ctx* ctx = nullptr;
    int cols = 0, rows = 0, n = 0, hb = 0;
    float ts_gen = 0.0f;
    double* d_stat = nullptr; long long* d_acc = nullptr;   // d_acc: Nq (n * ld) then gq (n), one allocation, one memset per solve
    float* d_mesh = nullptr; double* d_Lc = nullptr; double* d_N = nullptr;      // d_N: band then right-hand side
    int* d_flags = nullptr;
};

void lvk_mesh_solver_free(lvk_mesh_solver_dev* s)
{
    if (!s) return;
    void* dev[] = {s->d_stat, s->d_acc, s->d_mesh, s->d_Lc, s->d_N, s->d_flags};
    for (void* p : dev) if (p) (void)hipFree(p);
    delete s;
}

// generate_mesh_constraints for a cols x rows mesh (FrameTracker.cpp:380-457): the static band is built on the host and uploaded once
int lvk_mesh_solver_create(lvk_hip_ctx* ctx, int cols, int rows, float gen_w, float gen_h, float temporal, float local, lvk_mesh_solver_dev** out)
{
    LVK_HIP_REQUIRE(ctx, out != nullptr && cols >= 2 && rows >= 2);
    *out = nullptr;
    lvkh::MeshSolverH host;
    host.generate(cols, rows, gen_w, gen_h, temporal, local);
    LVK_HIP_REQUIRE(ctx, host.hb() <= MS_HB_MAX && host.n() <= MS_N_MAX);      // meshes wider than 16 columns / beyond 16 x 64: not supported by the device solver
    auto* s = new lvk_mesh_solver_dev();
    s->ctx = ctx; s->cols = cols; s->rows = rows; s->n = host.n(); s->hb = host.hb(); s->ts_gen = temporal;
    const size_t band = (size_t)s->n * (s->hb + 1);
    auto fail = [&](hipError_t e) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMalloc((void**)&s->d_stat, band * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_acc, (band + s->n) * sizeof(long long))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_mesh, s->n * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_Lc, (band + MS_NT) * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_N, (band + s->n) * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_flags, sizeof(int))) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(s->d_stat, host.static_band().data(), band * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->d_mesh, 0, s->n * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->d_Lc, 0, (band + MS_NT) * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->d_acc, 0, (band + s->n) * sizeof(long long))) != hipSuccess) return fail(e);       // kept clear by k_mesh_prepare
    if ((e = hipMemset(s->d_flags, 0, sizeof(int))) != hipSuccess) return fail(e);                            // kept clear by k_mesh_solve
    *out = s;
    return LVK_HIP_OK;
}

int lvk_mesh_solver_reset(lvk_mesh_solver_dev* s, hipStream_t stream)      // FrameTracker::restart: m_OptimizedMesh = 0 (:103)
{
    LVK_HIP_CHECK(s->ctx, hipMemsetAsync(s->d_mesh, 0, s->n * sizeof(float), stream));
    return LVK_HIP_OK;
}

int lvk_mesh_solver_cols(const lvk_mesh_solver_dev* s) { return s->cols; }
int lvk_mesh_solver_rows(const lvk_mesh_solver_dev* s) { return s->rows; }

// d_scratch: 8 x 4 bytes per pair (the pair's unknown indices and weights, kept between the kernels).  d_p1 / d_p2: tracked / matched points; d_count: pair count decided on the GPU (or nullptr: n_pts pairs).  Results are written to
// device-visible host memory: offsets (cols * rows * 2 floats), inlier flags, status (0 ok, 1 fewer than min_samples pairs, 2 a feature
// outside the mesh, 3 factorisation broke down); for a status != 0 the previous solution is left untouched.
int lvk_launch_mesh_solve(lvk_mesh_solver_dev* s, hipStream_t stream, void* d_scratch, const float2* d_p1, const float2* d_p2, const int* d_count, int n_pts,
                          int min_samples, float region_w, float region_h, float temporal_now, float threshold,
                          float* h_offsets, uint8_t* h_mask, int* h_status)
{
    lvk_hip_ctx* ctx = s->ctx;
    LVK_HIP_REQUIRE(ctx, n_pts >= 0 && d_scratch != nullptr);
    const size_t band = (size_t)s->n * (s->hb + 1);
    MeshArgs a;
    a.cols = s->cols; a.rows = s->rows; a.n = s->n; a.hb = s->hb; a.nbands = (s->hb + MS_TB) / MS_TB;
    a.stat = s->d_stat; a.Nq = s->d_acc; a.gq = s->d_acc + band; a.N = s->d_N; a.g0 = s->d_N + band; a.mesh = s->d_mesh; a.Lc = s->d_Lc;
    a.fidx = (int*)d_scratch; a.fw = (float*)d_scratch + 4 * (size_t)std::max(n_pts, 1); a.p1 = d_p1; a.p2 = d_p2; a.count = d_count; a.n_pts = n_pts; a.min_samples = min_samples;
    a.region_w = region_w; a.region_h = region_h; a.ts_gen = s->ts_gen; a.ts_now = temporal_now; a.threshold = threshold;
    a.flags = s->d_flags; a.out_offsets = h_offsets; a.out_mask = h_mask; a.out_status = h_status;
    if (n_pts > 0) hipLaunchKernelGGL(k_mesh_assemble, dim3((unsigned)((n_pts + 127) / 128)), dim3(128), 0, stream, a);
    hipLaunchKernelGGL(k_mesh_prepare, dim3(64), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(k_mesh_solve, dim3(1), dim3(MS_NT), 0, stream, a);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

// ---- C-ABI: the solver on its own (per-stage entry point, lvk_hip.h) ----------------------------------------------------------------
struct lvk_hip_mesh_solver
{
    lvk_hip_ctx* ctx = nullptr;
    lvk_mesh_solver_dev* dev = nullptr;
    float2* d_pts = nullptr;                 // tracked | matched
    void* d_scratch = nullptr; int cap = 0;
    float* h_offsets = nullptr; uint8_t* h_mask = nullptr; int* h_status = nullptr;      // pinned
};

extern "C" {

void lvk_hip_mesh_solver_destroy(lvk_hip_mesh_solver* s)
{
    if (!s) return;
    lvk_mesh_solver_free(s->dev);
    if (s->d_pts) (void)hipFree(s->d_pts);
    if (s->d_scratch) (void)hipFree(s->d_scratch);
    if (s->h_offsets) (void)hipHostFree(s->h_offsets);
    if (s->h_mask) (void)hipHostFree(s->h_mask);
    if (s->h_status) (void)hipHostFree(s->h_status);
    delete s;
}

int lvk_hip_mesh_solver_create(lvk_hip_ctx* ctx, int cols, int rows, float gen_region_w, float gen_region_h,
                               float temporal_smoothing, float local_smoothing, int max_points, lvk_hip_mesh_solver** out)
{
    if (!ctx) return LVK_HIP_ERR_ARG;
    LVK_HIP_REQUIRE(ctx, out != nullptr && max_points > 0);
    *out = nullptr;
    auto* s = new lvk_hip_mesh_solver();
    s->ctx = ctx; s->cap = max_points;
    int rc = lvk_mesh_solver_create(ctx, cols, rows, gen_region_w, gen_region_h, temporal_smoothing, local_smoothing, &s->dev);
    if (rc != LVK_HIP_OK) { delete s; return rc; }
    if (hipMalloc((void**)&s->d_pts, 2 * (size_t)max_points * sizeof(float2)) != hipSuccess ||
        hipMalloc(&s->d_scratch, 32 * (size_t)max_points) != hipSuccess ||
        hipHostMalloc((void**)&s->h_offsets, (size_t)cols * rows * 2 * sizeof(float), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&s->h_mask, (size_t)max_points, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&s->h_status, sizeof(int), hipHostMallocDefault) != hipSuccess)
    { lvk_hip_mesh_solver_destroy(s); return ctx->fail(LVK_HIP_ERR_RUNTIME, "mesh solver: allocation failed"); }
    *out = s;
    return LVK_HIP_OK;
}

int lvk_hip_mesh_solver_reset(lvk_hip_mesh_solver* s)
{
    if (!s) return LVK_HIP_ERR_ARG;
    return lvk_mesh_solver_reset(s->dev, s->ctx->stream);
}

// Returns 0 when an estimate was produced, 2 when a point fell into the last cell row / column of the mesh, 3 when the factorisation
// broke down (both: "no estimate", the previous solution is kept -- FrameTracker.cpp:243-247), negative on errors.
int lvk_hip_mesh_solver_solve(lvk_hip_mesh_solver* s, const float* tracked, const float* matched, int n, float region_w, float region_h,
                              float temporal_now, float threshold, uint8_t* inliers, float* offsets)
{
    if (!s) return LVK_HIP_ERR_ARG;
    lvk_hip_ctx* ctx = s->ctx;
    const int cap = s->cap;
    LVK_HIP_REQUIRE(ctx, tracked && matched && inliers && offsets && n >= 0 && n <= cap);
    hipStream_t st = ctx->stream;
    if (n > 0)
    {
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(s->d_pts, tracked, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(s->d_pts + cap, matched, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));
    }
    const int rc = lvk_launch_mesh_solve(s->dev, st, s->d_scratch, s->d_pts, s->d_pts + cap, nullptr, n, 0, region_w, region_h, temporal_now, threshold,
                                         s->h_offsets, s->h_mask, s->h_status);
    if (rc != LVK_HIP_OK) return rc;
    LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (*s->h_status != 0) return *s->h_status;
    std::memcpy(inliers, s->h_mask, (size_t)n);
    std::memcpy(offsets, s->h_offsets, (size_t)lvk_mesh_solver_cols(s->dev) * lvk_mesh_solver_rows(s->dev) * 2 * sizeof(float));
    return 0;
}

} // extern "C"
