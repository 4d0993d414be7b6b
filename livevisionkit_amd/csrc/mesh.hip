// Local motion estimate of the vector-field preset on the device.
//
// Replaces FrameTracker::estimate_local_motions (reference: LiveVisionKit/Vision/FrameTracker.cpp:200-321) with the constraint system of
// generate_mesh_constraints (:380-457).  The reference hands the sparse least-squares problem to Eigen::LeastSquaresConjugateGradient;
// this library solves the same problem directly through its normal equations N x = g:
//   static rows  -> a constant band matrix, built once per configuration on the host (host_logic.hpp, MeshSolverH::generate);
//   feature rows -> Q32 fixed-point sums accumulated with integer atomics (exact, order independent)          k_mesh_assemble
//   N = L D L^T  -> right-looking root-free band factorisation with reciprocal pivots, forward substitution carried along,
//                   every entry updated in pivot order by fused multiply-subtracts (binary64)                        k_mesh_solve
//   L^T          -> column-oriented backward substitution (one wavefront, the rows in flight in registers)       k_mesh_backsolve
//   inlier flags (L1 reprojection error through the feature's quad) and the normalised mesh offsets             k_mesh_backsolve
// The previous solution (the reference's m_OptimizedMesh: warm start there, right-hand side of the temporal rows here) stays on the
// device.  n = 2 * cols * rows unknowns, half bandwidth hb = 2 * (3 * cols + 3) + 1 (512 and 103 for the 16 x 16 mesh of the preset).
//
// The factorisation is one workgroup: the part of the band that the current pivots reach lives in REGISTERS (a 4-column x 8-offset tile
// per thread, per-band windows), the pivot columns and their scaled copies go through LDS, zero-padded so that the same straight-line
// update serves every thread and every step.  It is a chain of 512 dependent pivots: what it costs is instruction issue and latency
// per barrier interval, not arithmetic -- see "phase 1" below for the organisation and what was measured on the way (740 us for the
// first correct version, 155 us now; 38 us for the backward substitution).
#include "mesh_internal.hpp"
#include "host_logic.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

using namespace lvkmesh;

namespace {

// wave-uniform copies (the fields of a block descriptor are read with vector loads: without this every pointer derived from them occupies
// two VGPRs per lane instead of two SGPRs)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> __device__ __forceinline__ T* uniform(T* p)
{
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

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
    // nested dissection: every entry of the natural system goes to its place in ONE block's band (a.ndst / a.gdst; -1: structurally zero);
    // status bits of the frame for the kernels that follow (they only read *a.flags)
    if (a.nd && blockIdx.x == 0 && threadIdx.x == 0 && pair_count(a) < a.min_samples) atomicOr(a.flags, 8);
    for (size_t at = (size_t)blockIdx.x * blockDim.x + threadIdx.x; at < band + a.n; at += stride)
    {
        if (at < band)
        {
            const long long q = a.Nq[at]; a.Nq[at] = 0;
            double v = a.stat[at] + (double)q / MS_Q;
            if (at % ld == 0) v = v + 1e-6;
            if (!a.ndst) a.N[at] = v;
            else { const int d = a.ndst[at]; if (d >= 0) a.N[d] = v; }
        }
        else
        {
            const size_t i = at - band;
            const long long q = a.gq[i]; a.gq[i] = 0;
            a.g0[a.gdst ? a.gdst[i] : (int)i] = (double)a.ts_gen * (double)(a.ts_now * a.mesh[i]) + (double)q / MS_Q;
        }
    }
}

// entry (k + t, k) of N; 0 outside the matrix
__device__ __forceinline__ double load_entry(const MeshArgs& a, int k, int t)
{
    const bool in = k < a.n && t <= a.hb && k + t < a.n;
    const double v = a.N[in ? (size_t)k * (a.hb + 1) + t : 0];          // unconditional load, then a select: the loads of a tile stay in flight together
    return in ? v : 0.0;
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
    // The two pivot columns of an interval, by interval parity: raw[j][x] = N(p + x, p) after all earlier pivots, lcol[j] = the column times
    // the reciprocal pivot (column p of L, entry 0 stored as zero), both in the padded layout (logical index MS_PAD + x at ms_px(MS_PAD + x)),
    // zeros outside 0 .. hb: the updates rely on the padding instead of masks.  To the window threads the columns up to the chain's
    // two are spent: raw[0] is stored with zeros at x <= 3, raw[1] at x <= 2.  rinv: the reciprocal pivots.
    double raw[2][2][MS_LCOL_P], lcol[2][2][MS_LCOL_P];
    double rinv[2][2];
    double rawhead[2][2][4];                // the first entries of the two raw pivot columns, which raw[] stores as zeros (nested dissection: the forward wavefront writes the raw columns out)
    double col[2][2][MS_COL_P];             // the two columns behind the interval's pivots as the window threads leave them (read by the chain one interval later)
    double w[MS_N_MAX + 128];               // right-hand side: g, then D^-1 L^-1 g, in place (+ slack: rows beyond n)
    double next[MS_BANDS + 1][MS_NEXT_BAND];    // phase 1: per band, the columns that enter its window when the current group is done (+ a spare slot)
    int fail;
};

// ---- phase 1, look-ahead organisation -----------------------------------------------------------------------------------------------
// The elimination is latency bound: a single wavefront issues an instruction every 4-7 cycles, whatever it is, so the time between two
// barriers is the longest instruction stream plus the dependent latencies on it (a trip through LDS ~100 cycles under load, a division
// ~110, the barrier ~70) -- the arithmetic of the rank-1 update itself (6.2 K fused multiply-subtracts per pivot) is a quarter of it.
// Hence: few barriers, little per-barrier overhead.  The pivots go in PAIRS (one interval = two pivots, one barrier), three roles side
// by side, one interval apart:
//   * the chain (one wavefront) owns the NEXT pair.  During the interval of the pivots a, a + 1 it takes the columns a + 2, a + 3 as the
//     window threads left them after pivot a - 1 (col[]), applies a and a + 1 to them, then a + 2 to the second (the column of L it has
//     just formed, shifted by one lane with a whole-wavefront DPP move), and publishes both pivots (the other parity).
//   * the window (MS_WIN_WAVES wavefronts) applies a, then a + 1 (published during the previous interval) to its 4 x MS_TB register
//     tiles and hands the two columns behind the chain's to the chain through col[].
//   * the forward substitution (one wavefront) applies a, a + 1 to the right-hand side, scales their rows by the reciprocal pivots and
//     stores the columns a, a + 1 of L for phase 2.
// Every entry still receives its updates in pivot order with the same operands: bit-identical to the plain loop of the specification.
__device__ __forceinline__ double readlane64(double v, int src);

// lane j <- lane j + 1 of a binary64 register pair (v_mov_b32_dpp wave_shl:1 moves across the whole wavefront); lane 63 <- `last`
__device__ __forceinline__ double shift_down1(double v, double last)
{
    const long long u = __double_as_longlong(v), e = __double_as_longlong(last);
    const int lo = __builtin_amdgcn_update_dpp((int)(e & 0xffffffffll), (int)(u & 0xffffffffll), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(u >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// (All parts are written without per-entry conditions: the LDS arrays are padded with zeros, entries outside the band are zeros that
//  stay zeros, and every store is unconditional.  The workgroup's wavefronts share ONE scalar unit: exec-mask bookkeeping for
//  per-entry branches made the first version of this loop scalar-issue bound -- 125 scalar instructions per step and wavefront.)

// what the chain carries from one interval to the next (wave uniform): the unscaled entries of the two pivot columns it has formed in
// the rows of the two columns it forms next -- N(a + 2, a), N(a + 3, a), N(a + 2, a + 1), N(a + 3, a + 1) -- and the failure flag
struct ChainCarry { double a2, a3, b1, b2; bool bad; };

// the chain's part of the interval whose pivot data has parity PAR: finishes the columns c = a + 2 and c + 1 and turns them into the
// pivot data of the other parity
template <int PAR>
__device__ __forceinline__ void chain_interval(const MeshArgs& a, FactorShared& s, int c, ChainCarry& cc)
{
    if (c >= a.n) return;
    const int lane = (int)threadIdx.x;
    const int x0 = ms_px(MS_PAD + lane), x1 = ms_px(MS_PAD + 1 + lane), x2 = ms_px(MS_PAD + 2 + lane), x3 = ms_px(MS_PAD + 3 + lane);
    const int tc = lane + lane / MS_TB, tc2 = (lane + 64) + (lane + 64) / MS_TB;
    const double* la = s.lcol[PAR][0]; const double* lb = s.lcol[PAR][1];
    // band offsets t = lane (0) and t = lane + 64 (1) of the columns c (C) and c + 1 (D); beyond hb everything is zero
    double C0 = s.col[PAR][0][tc], C1 = s.col[PAR][0][tc2], D0 = s.col[PAR][1][tc], D1 = s.col[PAR][1][tc2];
    C0 = __builtin_fma(-la[x2], cc.a2, C0); C1 = __builtin_fma(-la[x2 + 80], cc.a2, C1);           // ms_px(x + 64) = ms_px(x) + 80
    D0 = __builtin_fma(-la[x3], cc.a3, D0); D1 = __builtin_fma(-la[x3 + 80], cc.a3, D1);
    C0 = __builtin_fma(-lb[x1], cc.b1, C0); C1 = __builtin_fma(-lb[x1 + 80], cc.b1, C1);
    D0 = __builtin_fma(-lb[x2], cc.b2, D0); D1 = __builtin_fma(-lb[x2 + 80], cc.b2, D1);
    const double dc = readlane64(C0, 0), c1 = readlane64(C0, 1);
    cc.a2 = readlane64(C0, 2); cc.a3 = readlane64(C0, 3);
    const double rc = 1.0 / dc;
    const double lc0 = lane == 0 ? 0.0 : C0 * rc, lc1 = C1 * rc;
    D0 = __builtin_fma(-shift_down1(lc0, readlane64(lc1, 0)), c1, D0);
    D1 = __builtin_fma(-shift_down1(lc1, 0.0), c1, D1);
    const double dd = readlane64(D0, 0);
    cc.b1 = readlane64(D0, 1); cc.b2 = readlane64(D0, 2);
    cc.bad = cc.bad || (c < a.n_elim && !(dc > 0.0)) || (c + 1 < a.n_elim && !(dd > 0.0));      // (a block's held-back separator pivots are formed one interval ahead and never used)
    const double rd = 1.0 / dd;
    s.raw[PAR ^ 1][0][x0] = lane <= 3 ? 0.0 : C0; s.raw[PAR ^ 1][0][x0 + 80] = C1;
    s.lcol[PAR ^ 1][0][x0] = lc0; s.lcol[PAR ^ 1][0][x0 + 80] = lc1;
    s.raw[PAR ^ 1][1][x0] = lane <= 2 ? 0.0 : D0; s.raw[PAR ^ 1][1][x0 + 80] = D1;
    s.lcol[PAR ^ 1][1][x0] = lane == 0 ? 0.0 : D0 * rd; s.lcol[PAR ^ 1][1][x0 + 80] = D1 * rd;
    if (lane == 0) { s.rinv[PAR ^ 1][0] = rc; s.rinv[PAR ^ 1][1] = rd; }
    if (lane <= 3) { s.rawhead[PAR ^ 1][0][lane] = C0; s.rawhead[PAR ^ 1][1][lane] = D0; }      // (the entries raw[] hides from the window threads: see forward_interval)
}

// The window.  An entry (column k, band offset t) is touched by pivot p iff (k - p) + t <= hb: a column at distance s from the pivot
// needs only its offsets t <= hb - s.  The offsets are therefore split into bands of MS_TB, and every band keeps its OWN window of
// column groups: band b (offsets from T = MS_TB b) holds the groups at distance m = 1 .. (hb - T + 3) / 4 from the pivots' group -- 26
// for the first band, 1 for the last, 182 tiles of 4 x 8 in all instead of the 351 of one common window (of which half would hold
// entries no pivot reaches yet).  A tile's thread follows the rotation of its band: m counts down; the tile at m = 1 hands its columns
// to the chain pair by pair and then takes over the group that enters the band's window.

// Entries of the column groups that enter the bands' windows when the group at p0 has been pivoted: fetch (N is read once, in band
// order) and hand-over to the tiles through LDS -- the forward-substitution wavefront's job: it has time to spare, the window
// wavefronts do not.  What does not depend on p0 is worked out once per lane: the entry's offset in N relative to column p0, the
// last row it touches relative to p0 (the matrix ends at n), and its slot in next[].
struct FetchPlan { int off[MS_PF], reach[MS_PF], dst[MS_PF]; };
__device__ __forceinline__ void fetch_plan(const MeshArgs& a, FetchPlan& fp, int lane)
{
    const int hb = a.hb, ld = hb + 1;
#pragma unroll
    for (int q = 0; q < MS_PF; q++)
    {
        const int idx = lane + 64 * q, pb = idx / (MS_CA * MS_TB), rem = idx - pb * (MS_CA * MS_TB), c = rem / MS_TB, ti = rem - c * MS_TB;
        const int k = MS_CA * (band_groups(hb, pb) + 1) + c, t = MS_TB * pb + ti;
        const bool in = idx < MS_BANDS * MS_CA * MS_TB && pb < a.nbands && t <= hb;
        fp.off[q] = k * ld + t; fp.reach[q] = in ? k + t : (1 << 29);
        fp.dst[q] = idx < MS_BANDS * MS_CA * MS_TB ? pb * MS_NEXT_BAND + rem : MS_BANDS * MS_NEXT_BAND;      // beyond the plan: the spare slot
    }
}
__device__ __forceinline__ void window_fetch(const MeshArgs& a, const FetchPlan& fp, double (&pf)[MS_PF], int p0)
{
    const double* col0 = a.N + (size_t)p0 * (a.hb + 1);
#pragma unroll
    for (int q = 0; q < MS_PF; q++) pf[q] = p0 + fp.reach[q] < a.n ? col0[fp.off[q]] : a.N[0];
}
__device__ __forceinline__ void window_park(const MeshArgs& a, FactorShared& s, const FetchPlan& fp, const double (&pf)[MS_PF], int p0)
{
#pragma unroll
    for (int q = 0; q < MS_PF; q++) (&s.next[0][0])[fp.dst[q]] = p0 + fp.reach[q] < a.n ? pf[q] : 0.0;
}

// the forward substitution's part of an interval, pivots p and p + 1: z(p + t) -= L(p + t, p) z(p) for the rows of the band, row p
// becomes z(p) / d(p); the columns of L go to global memory for phase 2.  (The wavefront's LDS accesses execute in order: rows written
// here are read back by the same wavefront without a barrier.)
template <int PAR>
__device__ __forceinline__ void forward_interval(const MeshArgs& a, FactorShared& s, int p, const FetchPlan& fp, double (&pf)[MS_PF])     // pf: the register set of this group's parity
{
    const int lane = (int)threadIdx.x & 63, hb = a.hb;
    // The columns that take over the slots of a pivoted group in each band's window travel two groups ahead (N was written by another
    // kernel on other XCDs: the first touch of a line comes from memory): what was fetched two groups ago is parked in LDS now (read
    // by the window threads at the end of this group), and the fetch for the group after the next is issued into the same registers.
    if (PAR == 0) { window_park(a, s, fp, pf, p); window_fetch(a, fp, pf, p + 2 * MS_CA); }
    const int x0 = ms_px(MS_PAD + lane);
    // both pivots in registers: the rows p .. p + 127 after pivot p are shifted down one lane (rows p + 1 .. p + 128) for pivot p + 1
    const double la0 = s.lcol[PAR][0][x0], la1 = s.lcol[PAR][0][x0 + 80], lb0 = s.lcol[PAR][1][x0], lb1 = s.lcol[PAR][1][x0 + 80];   // entry 0 is stored as zero
    const double ra = s.rinv[PAR][0], rb = s.rinv[PAR][1];
    const double w0 = s.w[p + lane], w1 = s.w[p + 64 + lane], w2 = s.w[p + 128];
    const double za = readlane64(w0, 0);
    const double n0 = __builtin_fma(-la0, za, w0), n1 = __builtin_fma(-la1, za, w1);                // rows beyond the matrix keep their zeros (l = 0)
    const double zb = readlane64(n0, 1);
    const double m0 = __builtin_fma(-lb0, zb, shift_down1(n0, readlane64(n1, 0)));
    const double m1 = __builtin_fma(-lb1, zb, shift_down1(n1, w2));
    if (lane == 0) s.w[p] = za * ra;
    s.w[p + 1 + lane] = lane == 0 ? zb * rb : m0;
    s.w[p + 65 + lane] = m1;
    double* Lp = a.Lc + (size_t)p * (hb + 1);
    if (lane <= hb) { Lp[lane] = la0; Lp[hb + 1 + lane] = lb0; }
    if (lane + 64 <= hb) { Lp[lane + 64] = la1; Lp[hb + 1 + lane + 64] = lb1; }
    if (a.Rc)
    {
        // a block of the nested dissection: the UNSCALED pivot columns too -- what its trailing separator window is rebuilt from after the
        // early stop (k_nd_sep_assemble).  raw[] holds them with the chain's own entries zeroed; those come from rawhead[].
        double r0 = s.raw[PAR][0][x0], q0 = s.raw[PAR][1][x0];
        const double r1 = s.raw[PAR][0][x0 + 80], q1 = s.raw[PAR][1][x0 + 80];
        if (lane <= 3) r0 = s.rawhead[PAR][0][lane];
        if (lane <= 2) q0 = s.rawhead[PAR][1][lane];
        // (addressed relative to the columns of L, behind an opaque scalar: the compiler otherwise keeps a second set of per-lane addresses
        //  alive across the whole elimination loop, and this kernel has no register to spare -- they went to scratch)
        long long rc_off = a.Rc - a.Lc;
        asm volatile("" : "+s"(rc_off));
        double* Rp = Lp + rc_off;
        if (lane <= hb) { Rp[lane] = r0; Rp[hb + 1 + lane] = q0; }
        if (lane + 64 <= hb) { Rp[lane + 64] = r1; Rp[hb + 1 + lane + 64] = q1; }
    }
}

// the window's part of an interval: the pivots p = p0 + 2 H and p + 1 of the group at p0, pivot data of parity H
template <int H>
__device__ __forceinline__ void window_interval(const MeshArgs& a, FactorShared& s, double (&A)[MS_CA][MS_TB],
                                                int p0, int& m, int nb, int band, bool valid)
{
    constexpr int PAR = H;
    if (valid)
    {
        const int t0 = band * MS_TB;
        const int yr = 5 * m, yl = 5 * (m + (MS_TB / 4) * band);        // padded positions of MS_PAD + 4 m and MS_PAD + 4 m + t0, less ms_px(MS_PAD)
        double rc[2][MS_CA], lw[2][MS_CA + MS_TB - 1];                  // operands of the pivots p0 + 2 H + j: the tile's first column is at distance 4 m - 2 H - j
#pragma unroll
        for (int j = 0; j < 2; j++)
        {
#pragma unroll
            for (int ck = 0; ck < MS_CA; ck++) rc[j][ck] = s.raw[PAR][j][yr + ms_px(MS_PAD + ck - 2 * H - j)];       // zero for the chain's columns
#pragma unroll
            for (int q = 0; q < MS_CA + MS_TB - 1; q++) lw[j][q] = s.lcol[PAR][j][yl + ms_px(MS_PAD + q - 2 * H - j)];
        }
        // the columns p + 4, p + 5 first: they are the chain's input of the next interval, and the rest of the update covers the time
        // their stores take
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
            {
                const int ck = half == 0 ? 2 * H + cc : 2 * (1 - H) + cc;
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = __builtin_fma(-lw[j][ck + ti], rc[j][ck], A[ck][ti]);
            }
            if (half == 0)
            {
                if (m == 1)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int ti = 0; ti < MS_TB; ti++) s.col[PAR ^ 1][j][t0 + band + ti] = A[2 * H + j][ti];
            }
        }
        if (H == 1)
        {
            if (m == 1)
#pragma unroll
                for (int ck = 0; ck < MS_CA; ck++)
#pragma unroll
                    for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = s.next[band][ck * MS_TB + ti];
            m = m == 1 ? nb : m - 1;
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
    if (a.blocks)
    {
        const MeshBlockDev b = a.blocks[blockIdx.x];
        a.n = uniform(b.n); a.hb = uniform(b.hb); a.nbands = uniform(b.nbands); a.n_elim = uniform(b.n_elim);
        a.N = uniform(b.N); a.g0 = uniform(b.g0); a.wz = uniform(b.wz); a.Lc = uniform(b.Lc); a.Rc = uniform(b.Rc);
    }
    const int n = a.n, hb = a.hb;
    const int m = pair_count(a);
    const int flags = *a.flags;
    __syncthreads();
    if (a.nd)
    {
        // nested dissection: the kernels of a solve only READ the status bits of earlier kernels (1 a feature outside the mesh, 4 a
        // factorisation broke down, 8 too few pairs) and OR their own in; the last kernel reports and clears them
        if (flags & 0xf) return;
    }
    else
    {
    // *a.flags on exit: 0 = factorised, k_mesh_backsolve goes ahead; 2 = no solution this frame (the status is already with the host)
    if (m < a.min_samples) { if (tid == 0) { *a.out_status = 1; *a.flags = 2; } return; }
    if (flags & 1) { if (tid == 0) { *a.out_status = 2; *a.flags = 2; } return; }
    }
#ifdef LVK_MESH_TIMING
    const long long tm0 = wall_clock64();
#endif

    // ---- phase 1: N = L D L^T and w = D^-1 L^-1 g
    const int wave = tid >> 6;
    const bool chain = wave == 0, forward = wave == MS_FWD_WAVE;
    const int btid = (wave - 1 - (wave > MS_FWD_WAVE ? 1 : 0)) * 64 + (tid & 63);      // index among the window threads
    // window thread -> (band, slot), slot-major: neighbouring lanes hold neighbouring BANDS of one column group (see the LDS layouts)
    int band = a.nbands, slot = 0, nb = 1;
    if (!chain && !forward)
    {
        int rem = btid;
        for (int j = 0; j < band_groups(hb, 0); j++)
        {
            const int cnt = min(a.nbands, (hb + 3 - MS_CA * (j + 1)) / MS_TB + 1);     // bands that have a slot j: band_groups(hb, b) > j
            if (rem < cnt) { band = rem; slot = j; break; }
            rem -= cnt;
        }
        nb = band_groups(hb, min(band, a.nbands - 1));
    }
    const bool valid = !chain && !forward && band < a.nbands;
    int gdist = slot + 1;                                               // distance (in column groups) of this tile's group from the pivots'
    if (chain) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);      // the chain is the critical path
    double A[MS_CA][MS_TB], pf[2][MS_PF];
#pragma unroll
    for (int ck = 0; ck < MS_CA; ck++)
#pragma unroll
        for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = load_entry(a, valid ? MS_CA * gdist + ck : n, band * MS_TB + ti);     // invalid threads: zeros
#pragma unroll
    for (int q = 0; q < MS_PF; q++) pf[0][q] = pf[1][q] = 0.0;
    FetchPlan fp;
    fetch_plan(a, fp, tid & 63);
    if (forward) { window_fetch(a, fp, pf[0], 0); window_fetch(a, fp, pf[1], MS_CA); }      // the groups that enter at the end of the first two groups
    for (int i = tid; i < 4 * MS_LCOL_P; i += MS_NT) { (&s.raw[0][0][0])[i] = 0.0; (&s.lcol[0][0][0])[i] = 0.0; }
    for (int i = tid; i < 4 * MS_COL_P; i += MS_NT) (&s.col[0][0][0])[i] = 0.0;
    for (int i = tid; i < MS_BANDS * MS_NEXT_BAND; i += MS_NT) (&s.next[0][0])[i] = 0.0;
    for (int i = tid; i < n + 128; i += MS_NT) s.w[i] = i < n ? a.g0[i] : 0.0;
    if (tid == 0) s.fail = 0;
    __syncthreads();
    // start-up: the interval before the first -- the chain forms the pivots 0 and 1 from the columns 0, 1 of N with nothing to apply
    // (pivot data of parity 1 is all zeros); the columns 2, 3 are its input of the first interval
    for (int t = tid; t <= hb; t += MS_NT)
    {
        s.col[1][0][t + t / MS_TB] = load_entry(a, 0, t); s.col[1][1][t + t / MS_TB] = load_entry(a, 1, t);
        s.col[0][0][t + t / MS_TB] = load_entry(a, 2, t); s.col[0][1][t + t / MS_TB] = load_entry(a, 3, t);
    }
    __syncthreads();
    ChainCarry cc = {0.0, 0.0, 0.0, 0.0, false};
    if (chain) chain_interval<1>(a, s, 0, cc);
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
    // the intervals: the pivots (p0, p0 + 1) with the pivot data of parity 0 (written by the start-up interval), (p0 + 2, p0 + 3) with parity 1
    const int n_elim = a.n_elim;
    for (int p0 = 0; p0 < n_elim; p0 += 2 * MS_CA)
    {
#define LVK_MESH_INTERVAL(GV, HV)                                                                               \
        if (p0 + MS_CA * GV + 2 * HV < n_elim)                                                                   \
        {                                                                                                        \
            MESH_PROBE_BEGIN();                                                                                  \
            if (chain) chain_interval<HV>(a, s, p0 + MS_CA * GV + 2 * HV + 2, cc);                               \
            else if (forward) forward_interval<HV>(a, s, p0 + MS_CA * GV + 2 * HV, fp, pf[GV]);                  \
            else window_interval<HV>(a, s, A, p0 + MS_CA * GV, gdist, nb, band, valid);                          \
            MESH_PROBE_MID();                                                                                    \
            lds_barrier();                                                                                       \
            MESH_PROBE_END();                                                                                    \
        }
        LVK_MESH_INTERVAL(0, 0) LVK_MESH_INTERVAL(0, 1) LVK_MESH_INTERVAL(1, 0) LVK_MESH_INTERVAL(1, 1)
#undef LVK_MESH_INTERVAL
    }
    if (tid == 0 && cc.bad) s.fail = 1;
    __syncthreads();
#ifdef LVK_MESH_TIMING
    if (tid == 0) printf("mesh factor: init %lld, factor %lld (100 MHz ticks), n %d hb %d\n", tm1 - tm0, wall_clock64() - tm1, n, hb);
#if LVK_MESH_TIMING > 1
    if (tid == 0)
    {
        printf("  per pivot (shader cycles): chain work %lld wait %lld | forward work %lld wait %lld | window work %lld wait %lld\n",
               g_mesh_phase[0] / n, g_mesh_phase[1] / n, g_mesh_phase[2] / n, g_mesh_phase[3] / n, g_mesh_phase[4] / n, g_mesh_phase[5] / n);
        for (int k = 0; k < 8; k++) g_mesh_phase[k] = 0;
    }
#endif
#endif
    if (s.fail != 0) { if (tid == 0) { if (a.nd) atomicOr(a.flags, 4); else { *a.out_status = 3; *a.flags = 2; } } return; }
    for (int i = tid; i < n; i += MS_NT) a.wz[i] = s.w[i];
    if (!a.nd) { if (tid == 0) *a.flags = 0; return; }
}

// The separator system of the nested dissection: S = sum of what the blocks' pivots left of their separator x separator windows, g = sum of
// their reduced right-hand sides, in ascending block order starting from +0 (at most two blocks meet in an entry); band layout of
// k_mesh_solve.  A block's window lived in the registers of its factorisation and is gone: every entry is rebuilt here from its original
// value and the columns of L and of the unscaled pivots in pivot order -- the same fused multiply-subtracts with the same operands the
// right-looking update applied to it.  One thread per entry, many workgroups (a source: block << 16 | row << 8 | column of the window).
__device__ __forceinline__ double nd_trailing_entry(const MeshBlockDev& b, int li, int lk)
{
    const int ld = b.hb + 1, i = b.n_elim + li, k = b.n_elim + lk;
    double acc = b.N[(size_t)k * ld + (i - k)];
    for (int j = max(0, i - b.hb); j < b.n_elim; j++) acc = __builtin_fma(-b.Lc[(size_t)j * ld + (i - j)], b.Rc[(size_t)j * ld + (k - j)], acc);
    return acc;
}

__global__ __launch_bounds__(64)
void k_nd_sep_assemble(MeshArgs a)
{
    LVK_TRACKER_PRIORITY();
    if (*a.flags & 0xf) return;
    const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e < a.s_entries)
    {
        double v = 0.0;
        const int s0 = a.ssrc[2 * e], s1 = a.ssrc[2 * e + 1];
        if (s0 >= 0) v = v + nd_trailing_entry(a.blocks[s0 >> 16], (s0 >> 8) & 0xff, s0 & 0xff);
        if (s1 >= 0) v = v + nd_trailing_entry(a.blocks[s1 >> 16], (s1 >> 8) & 0xff, s1 & 0xff);
        a.N[e] = v;
    }
    else if (e < a.s_entries + a.ns)
    {
        const int q = e - a.s_entries;
        double v = 0.0;
        const int s0 = a.gsrc[2 * q], s1 = a.gsrc[2 * q + 1];
        if (s0 >= 0) v = v + a.wzall[s0];
        if (s1 >= 0) v = v + a.wzall[s1];
        a.g0[q] = v;
    }
}

// The same for meshes whose vertex rows are a multiple of 16 unknowns wide (8 and 16 columns: the OBS preset): one workgroup per 16 x 16 tile
// of S.  All entries of a tile take their (at most two) contributions from the same two blocks, at consecutive rows / columns of those
// blocks' windows -- so the 16 rows of L and the 16 raw rows the tile needs are staged in LDS once per source and every thread walks its
// chain out of LDS (the one-thread-per-entry kernel above follows the two strided global loads of every step: 47 us for the preset's
// 6 K entries; this one ~5 us).  Terms, operands and order are the same.
__global__ __launch_bounds__(256)
void k_nd_sep_assemble_tiled(MeshArgs a, int tiles_per_col)
{
    LVK_TRACKER_PRIORITY();
    __shared__ double sL[NT_TILE][NT_JMAX + 2], sR[NT_TILE][NT_JMAX + 2];      // pitch 121 doubles: the 16 rows a wavefront reads at one step fall on 16 different bank pairs
    if (*a.flags & 0xf) return;
    const int tid = (int)threadIdx.x, lds = a.hb + 1;
    const int ntiles = (a.ns / NT_TILE) * tiles_per_col;
    if ((int)blockIdx.x >= ntiles)
    {
        for (int q = tid; q < a.ns; q += 256)
        {
            double v = 0.0;
            const int s0 = a.gsrc[2 * q], s1 = a.gsrc[2 * q + 1];
            if (s0 >= 0) v = v + a.wzall[s0];
            if (s1 >= 0) v = v + a.wzall[s1];
            a.g0[q] = v;
        }
        return;
    }
    // tile (column block tc, band block tt): separator entries (si, sk) = (16 (tc + tt) + r, 16 tc + c), r, c < 16
    const int tc = (int)blockIdx.x / tiles_per_col, tt = (int)blockIdx.x - tc * tiles_per_col;
    const int r = tid >> 4, c = tid & 15;
    const int sk = NT_TILE * tc + c, si = NT_TILE * (tc + tt) + r;
    const bool inside = si < a.ns && si - sk <= a.hb && si >= sk;
    const int e0 = (NT_TILE * tc) * lds + NT_TILE * tt;                                  // the tile's first entry (r = c = 0) names its sources
    double v = 0.0;
    const bool tile_exists = NT_TILE * (tc + tt) < a.ns && NT_TILE * tt <= a.hb;
    for (int src = 0; src < 2; src++)
    {
        const int code = tile_exists ? a.ssrc[2 * e0 + src] : -1;
        if (code < 0) continue;                                                        // (uniform over the workgroup)
        const MeshBlockDev b = a.blocks[code >> 16];
        const int li0 = (code >> 8) & 0xff, lk0 = code & 0xff, ld = b.hb + 1;
        const int jmin = max(0, b.n_elim + li0 - b.hb), J = b.n_elim - jmin;              // the pivots that reach the tile's first row
        __syncthreads();
        // (all loads of the sweep in flight together: the columns were written by other compute units a moment ago and come from memory)
        constexpr int SWEEPS = (NT_TILE * NT_JMAX + 255) / 256;
        double vl[SWEEPS], vr[SWEEPS];
#pragma unroll
        for (int it = 0; it < SWEEPS; it++)
        {
            const int x = tid + 256 * it, jj = x >> 4, row = x & (NT_TILE - 1), j = jmin + jj;      // 16 lanes read 16 consecutive entries of one column: a 128-byte segment
            const int oi = b.n_elim + li0 + row - j, ok = b.n_elim + lk0 + row - j;
            const bool live = x < NT_TILE * J;
            vl[it] = b.Lc[live && oi <= b.hb ? (size_t)j * ld + oi : 0];
            vr[it] = b.Rc[live && ok <= b.hb ? (size_t)j * ld + ok : 0];
            if (!(live && oi <= b.hb)) vl[it] = 0.0;
            if (!(live && ok <= b.hb)) vr[it] = 0.0;
        }
#pragma unroll
        for (int it = 0; it < SWEEPS; it++)
        {
            const int x = tid + 256 * it, jj = x >> 4, row = x & (NT_TILE - 1);
            if (x < NT_TILE * J) { sL[row][jj] = vl[it]; sR[row][jj] = vr[it]; }
        }
        __syncthreads();
        if (inside)
        {
            const int i = b.n_elim + li0 + r, k = b.n_elim + lk0 + c;
            double acc = b.N[(size_t)k * ld + (i - k)];
            // (operands of the next steps are fetched while the dependent chain of fused multiply-subtracts advances)
#pragma unroll 8
            for (int jj = max(0, i - b.hb) - jmin; jj < J; jj++) acc = __builtin_fma(-sL[r][jj], sR[c][jj], acc);
            v = v + acc;
        }
    }
    if (inside) a.N[(size_t)sk * lds + (si - sk)] = v;
}

// phase 3 of both solvers: the solution as float (Eigen::VectorXf m_OptimizedMesh), inlier flags, offsets (FrameTracker.cpp:276-320);
// sol(i): the solution in binary64 (a functor, so that an LDS array stays an LDS access: handing k_mesh_backsolve's shared array over as a
// generic pointer cost it 32 VGPRs and put 112 bytes of its walking wavefront's state into scratch -- 43 -> 89 us), visible to every thread
template <class Sol>
__device__ __forceinline__ void finish_solution(const MeshArgs& a, Sol sol, int tid, int nthreads, int m)
{
    const int n = a.n;
    for (int i = tid; i < n; i += nthreads) a.mesh[i] = (float)sol(i);
    __syncthreads();
    for (int f = tid; f < m; f += nthreads)
    {
        const int* id = a.fidx + 4 * f; const float* wq = a.fw + 4 * f;
        const float x = wq[0] * a.mesh[id[0]] + wq[1] * a.mesh[id[1]] + wq[2] * a.mesh[id[2]] + wq[3] * a.mesh[id[3]];
        const float y = wq[0] * a.mesh[id[0] + 1] + wq[1] * a.mesh[id[1] + 1] + wq[2] * a.mesh[id[2] + 1] + wq[3] * a.mesh[id[3] + 1];
        a.out_mask[f] = (fabsf(x - a.p2[f].x) + fabsf(y - a.p2[f].y)) < a.threshold ? 1 : 0;
    }
    const float kw = (((float)a.cols / (float)(a.cols - 1)) * a.region_w) / (float)a.cols;
    const float kh = (((float)a.rows / (float)(a.rows - 1)) * a.region_h) / (float)a.rows;
    for (int v = tid; v < a.cols * a.rows; v += nthreads)
    {
        const int r = v / a.cols, c = v - r * a.cols;
        a.out_offsets[2 * v] = ((float)c * kw - a.mesh[2 * v]) / a.region_w;
        a.out_offsets[2 * v + 1] = ((float)r * kh - a.mesh[2 * v + 1]) / a.region_h;
    }
    __syncthreads();
    if (tid == 0) { __threadfence_system(); *a.out_status = 0; }
}

// ---- phase 2 and 3: a kernel of its own (one walking wavefront, eight that stage; the two phases in one kernel spilled registers) ------
struct BackShared
{
    double w[MS_N_MAX + 128];               // D^-1 L^-1 g, then the solution, in place
    double lt[2][MS_CHUNK][MS_RPITCH];      // rows of L as rings, staged round by round
    double xs[MS_N_MAX];                    // nested dissection: the separator system's solution (every block substitutes it backwards itself)
};

__global__ __launch_bounds__(MB_NT)
void k_mesh_backsolve(MeshArgs a)
{
    LVK_TRACKER_PRIORITY();
    __shared__ BackShared s;
    __shared__ unsigned s_last;
    const int tid = (int)threadIdx.x;
    // nested dissection: either the separator system (a whole system whose solution goes to a.xs) or one block per workgroup (its
    // separators' values given; the solutions are gathered in a.X and the last workgroup to finish runs phase 3 for the whole mesh)
    const bool nd_block = a.blocks != nullptr, nd_sep = a.nd && !nd_block;
    const int* xs_of = nullptr; const int* nat_of = nullptr;
    if (nd_block)
    {
        const MeshBlockDev b = a.blocks[blockIdx.x];
        a.n = uniform(b.n); a.hb = uniform(b.hb); a.n_elim = uniform(b.n_elim); a.wz = uniform(b.wz); a.Lc = uniform(b.Lc); xs_of = uniform(b.xs_of); nat_of = uniform(b.nat_of);
    }
    const int n = a.n, hb = a.hb;
    const int m = pair_count(a);
    const int flags = *a.flags;
    __syncthreads();
    if (!a.nd && (flags & 2)) { if (tid == 0) *a.flags = 0; return; }   // cleared for the next frame
    if (nd_sep && (flags & 0xf)) return;
    const bool dead = nd_block && (flags & 0xf);                        // no solution this frame: only the last workgroup's report is left to do
#ifdef LVK_MESH_TIMING
    const long long tm2 = wall_clock64();
#endif
    if (!dead)
    {
    if (nd_block && a.fuse_sep) { }                                  // (filled below, behind the separator system)
    else if (nd_block) for (int i = tid; i < n + 128; i += MB_NT) s.w[i] = i < a.n_elim ? a.wz[i] : (i < n ? a.xs[xs_of[i - a.n_elim]] : 0.0);
    else for (int i = tid; i < n + 128; i += MB_NT) s.w[i] = i < n ? a.wz[i] : 0.0;
    __syncthreads();

    // ---- phase 2: L^T x = w column by column: x(j) = w(j); w(k) = fma(-L(j, k), x(j), w(k)) for the rows k above j in the band.
    // One wavefront walks the rows j = n - 1 .. 0 with the rows in flight in three registers per lane: register s holds the rows of the
    // 64-row block b with b mod 3 = s (three blocks cover the band).  The other wavefronts stage the rows of L it needs in LDS, 16 rows per
    // round, each row as a ring of 192 slots -- L(j, k) at slot k mod 192, zeros outside the band -- so that the walking wavefront reads
    // its three operands of a row at FIXED addresses (lane, lane + 64, lane + 128) and a row costs it 9 instructions: two readlanes for
    // x(j), three LDS reads, three fused multiply-subtracts (an earlier version looked the band offsets up per lane: 25 instructions
    // per row, 77 us per solve; the wavefront's instruction count IS the time of this phase).  The rows run to the next multiple of 16
    // above n: the extra rows are zeros and change nothing, and every round and every block boundary is aligned.
    auto phase2 = [&](const double* Lcp, const int n, const int hb) {
        const int ld = hb + 1;
        const int n16 = (n + MS_CHUNK - 1) / MS_CHUNK * MS_CHUNK, nchunks = n16 / MS_CHUNK;
        const int lane = tid & 63;
        constexpr int STG = MB_NT - 64, PER = (MS_CHUNK * MS_RING + STG - 1) / STG;     // staging threads, ring slots per thread and round
        static_assert(STG % MS_CHUNK == 0, "a staging thread keeps its row of the round");
        const int st = tid - 64, srow = st & (MS_CHUNK - 1), sslot = st / MS_CHUNK;      // element q of a staging thread: row srow, slot sslot + (STG / 16) q
        double stage[2][PER];
        // rows top - srow of round c (top = n16 - 1 - 16 c): fetch into registers, two rounds ahead of their use
        auto fetch = [&](int c, double (&reg)[PER]) {
            const int j = n16 - 1 - MS_CHUNK * c - srow;
#pragma unroll
            for (int q = 0; q < PER; q++)
            {
                const int slot = sslot + (STG / MS_CHUNK) * q;
                const int t = (int)((unsigned)(j - slot + 2 * MS_RING) % (unsigned)MS_RING), k = j - t;      // the column whose slot this is: k = j - t, k mod 192 = slot
                const bool in = slot < MS_RING && j < n && t >= 1 && t <= hb && k >= 0;
                reg[q] = Lcp[in ? (size_t)k * ld + t : 0];
            }
        };
        auto commit = [&](int c, const double (&reg)[PER]) {
            const int j = n16 - 1 - MS_CHUNK * c - srow;
#pragma unroll
            for (int q = 0; q < PER; q++)
            {
                const int slot = sslot + (STG / MS_CHUNK) * q;
                const int t = (int)((unsigned)(j - slot + 2 * MS_RING) % (unsigned)MS_RING), k = j - t;
                const bool in = j < n && t >= 1 && t <= hb && k >= 0;
                if (slot < MS_RING) s.lt[c & 1][srow][slot] = in ? reg[q] : 0.0;
            }
        };
        // the walking wavefront: blocks B, B - 1, B - 2 (B = the block of row n16 - 1) in the registers of their residues
        double R[3] = {0.0, 0.0, 0.0};
        auto block_rows = [&](int b) -> double { const int i = 64 * b + lane; return (b >= 0 && i < n) ? s.w[i] : 0.0; };
        const int Btop = (n16 - 1) >> 6;
        if (tid < 64)
#pragma unroll
            for (int d = 0; d < 3; d++)
            {
                const int b = Btop - d, sgm = b >= 0 ? b % 3 : 0;
                const double v = block_rows(b);
                if (b >= 0) { if (sgm == 0) R[0] = v; else if (sgm == 1) R[1] = v; else R[2] = v; }
            }
        auto walk = [&](int c, auto sc_tag) {
            constexpr int SC = decltype(sc_tag)::value;
            const int top = n16 - 1 - MS_CHUNK * c, B = top >> 6;
            const double* ring = &s.lt[c & 1][0][lane];
#pragma unroll
            for (int r = 0; r < MS_CHUNK; r++)
            {
                const double xj = readlane64(R[SC], (top - r) & 63);
                R[0] = __builtin_fma(-ring[r * MS_RPITCH], xj, R[0]);
                R[1] = __builtin_fma(-ring[r * MS_RPITCH + 64], xj, R[1]);
                R[2] = __builtin_fma(-ring[r * MS_RPITCH + 128], xj, R[2]);
            }
            if (((top - (MS_CHUNK - 1)) & 63) == 0)
            {
                // block B is final: x of its rows; its register takes the rows of block B - 3
                if (64 * B + lane < n) s.w[64 * B + lane] = R[SC];
                R[SC] = block_rows(B - 3);
            }
        };
        auto walk_round = [&](int c) {
            const int sc = ((n16 - 1 - MS_CHUNK * c) >> 6) % 3;
            if (sc == 0) walk(c, std::integral_constant<int, 0>{});
            else if (sc == 1) walk(c, std::integral_constant<int, 1>{});
            else walk(c, std::integral_constant<int, 2>{});
        };
        if (tid >= 64)
        {
            fetch(0, stage[0]); commit(0, stage[0]);
            if (nchunks > 1) fetch(1, stage[1]);
            if (nchunks > 2) fetch(2, stage[0]);
        }
        __syncthreads();
        for (int c = 0; c < nchunks; c += 2)
        {
            // round c is walked while round c + 1 is written to the other buffer and round c + 3 is fetched
            if (tid < 64) walk_round(c);
            else { if (c + 1 < nchunks) commit(c + 1, stage[1]); if (c + 3 < nchunks) fetch(c + 3, stage[1]); }
            lds_barrier();
            if (c + 1 >= nchunks) break;
            if (tid < 64) walk_round(c + 1);
            else { if (c + 2 < nchunks) commit(c + 2, stage[0]); if (c + 4 < nchunks) fetch(c + 4, stage[0]); }
            lds_barrier();
        }
        __syncthreads();
    };
    // nested dissection with the separator system's backward substitution inside (fuse_sep): pass 0 substitutes the separator system
    // backwards -- every block for itself (96 rows; a kernel of its own would cost a kernel boundary on the critical chain and finish no
    // sooner) --, pass 1 the block with the separators' values given.  One call site: phase 2 is compiled once.
    const bool fused = nd_block && a.fuse_sep;
    for (int pass = fused ? 0 : 1; pass < 2; pass++)
    {
        if (fused && pass == 0)
        {
            for (int i = tid; i < a.ns + 128; i += MB_NT) s.w[i] = i < a.ns ? a.sep_wz[i] : 0.0;
            __syncthreads();
        }
        else if (fused)
        {
            for (int i = tid; i < a.ns; i += MB_NT) { s.xs[i] = s.w[i]; if (blockIdx.x == 0) a.xs[i] = s.w[i]; }
            __syncthreads();
            for (int i = tid; i < n + 128; i += MB_NT) s.w[i] = i < a.n_elim ? a.wz[i] : (i < n ? s.xs[xs_of[i - a.n_elim]] : 0.0);
            __syncthreads();
        }
        const bool sep_pass = fused && pass == 0;
        phase2(sep_pass ? a.sep_Lc : a.Lc, sep_pass ? a.ns : n, sep_pass ? a.sep_hb : hb);
    }
#ifdef LVK_MESH_TIMING
    if (tid == 0) printf("mesh backsolve: %lld (100 MHz ticks)\n", wall_clock64() - tm2);
#endif
    }   // !dead

    if (nd_sep) { for (int i = tid; i < n; i += MB_NT) a.xs[i] = s.w[i]; return; }
    if (nd_block)
    {
        if (!dead) for (int i = tid; i < a.n_elim; i += MB_NT) a.X[nat_of[i]] = s.w[i];
        __threadfence();                                                // this workgroup's part of X before its ticket
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(a.ticket, 1u) == (unsigned)a.nblocks - 1u ? 1u : 0u;
        __syncthreads();
        if (!s_last) return;
        __threadfence();                                                // the other workgroups' parts after theirs
        if (tid == 0) *a.ticket = 0;
        if (dead)
        {
            if (tid == 0) { *a.out_status = (flags & 8) ? 1 : ((flags & 1) ? 2 : 3); *a.flags = 0; }
            return;
        }
        for (int q = tid; q < a.ns; q += MB_NT) a.X[a.sep_nat[q]] = a.fuse_sep ? s.xs[q] : a.xs[q];
        __syncthreads();
        a.n = a.n_nat;
        const double* X = a.X;
        finish_solution(a, [&](int i) { return __builtin_nontemporal_load(X + i); }, tid, MB_NT, m);
        return;
    }

    // ---- phase 3: the solution as float, inlier flags, offsets (FrameTracker.cpp:276-320)
    finish_solution(a, [&](int i) { return s.w[i]; }, tid, MB_NT, m);
}

// ---- the same factorisation for ANY mesh (WarpMesh.cpp:34-41,79-90 allows every N x M; FrameTracker.cpp:57-92 takes every motion_resolution) --
// k_mesh_solve keeps the window of the band in registers, which fits half bandwidths up to 103 (16 columns) and 2048 unknowns.  Wider
// meshes (motion_resolution 17 x 17, 32 x 32, ...) and the tiny ones below its tile shapes run the SAME specification -- pivot order, one
// fused multiply-subtract per entry and pivot, reciprocal pivots: every entry sees the same operands in the same order, hence the same
// bits -- straight out of global memory (the band of a 32 x 32 mesh is 3.3 MB: L2 / MALL resident), one workgroup of 1024 threads, two
// barriers per pivot: ~2-3 us per pivot, a few milliseconds per solve.  Not a fast path: the presets never get here (2 x 2 meshes take
// the homography route, 16 x 16 the register-window solver).

__global__ __launch_bounds__(MG_NT)
void k_mesh_solve_generic(MeshArgs a)
{
    LVK_TRACKER_PRIORITY();
    __shared__ double raw[MG_HB_MAX + 1], lc[MG_HB_MAX + 1];
    __shared__ int bad;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = a.n, hb = a.hb, ld = hb + 1;
    const int m = pair_count(a);
    const int flags = *a.flags;
    if (tid == 0) bad = 0;
    __syncthreads();
    if (a.nd) { if (flags & 0xf) return; }                          // (the separator system of a very tall mesh: see k_mesh_solve)
    else
    {
        if (m < a.min_samples) { if (tid == 0) { *a.out_status = 1; *a.flags = 2; } return; }
        if (flags & 1) { if (tid == 0) { *a.out_status = 2; *a.flags = 2; } return; }
    }
    double* N = a.N; double* g = a.g0;
    for (int j = 0; j < n; j++)
    {
        const int len = min(n - 1 - j, hb);                             // rows j + 1 .. j + len are reached by pivot j
        const double d = N[(size_t)j * ld];
        const double r = 1.0 / d;
        if (tid == 0 && !(d > 0.0)) bad = 1;
        for (int t = 1 + tid; t <= len; t += MG_NT)
        {
            const double v = N[(size_t)j * ld + t];
            raw[t] = v; lc[t] = v * r;
            a.Lc[(size_t)j * ld + t] = v * r;
        }
        __syncthreads();
        const double gj = g[j];
        // right-hand side, then row j of D^-1 L^-1 g
        for (int t = 1 + tid; t <= len; t += MG_NT) g[j + t] = __builtin_fma(-lc[t], gj, g[j + t]);
        if (tid == 0) a.wz[j] = gj * r;
        // N(i, k) = fma(-L(i, j), N(k, j), N(i, k)) for j < k <= i <= j + len: a wavefront per column k, lanes over the rows
        for (int kk = 1 + wave; kk <= len; kk += MG_NT / 64)
        {
            const double c = raw[kk];
            double* col = N + (size_t)(j + kk) * ld;
            for (int t = lane; t <= len - kk; t += 64) col[t] = __builtin_fma(-lc[kk + t], c, col[t]);
        }
        __syncthreads();
    }
    if (bad) { if (tid == 0) { if (a.nd) atomicOr(a.flags, 4); else { *a.out_status = 3; *a.flags = 2; } } return; }
    if (tid == 0 && !a.nd) *a.flags = 0;
}

__global__ __launch_bounds__(MG_NT)
void k_mesh_backsolve_generic(MeshArgs a)
{
    LVK_TRACKER_PRIORITY();
    const int tid = (int)threadIdx.x;
    const int n = a.n, hb = a.hb, ld = hb + 1;
    const int m = pair_count(a);
    const int flags = *a.flags;
    __syncthreads();
    if (!a.nd && (flags & 2)) { if (tid == 0) *a.flags = 0; return; }
    if (a.nd && (flags & 0xf)) return;
    double* w = a.wz;
    // L^T x = w column by column: x(j) = w(j); w(k) = fma(-L(j, k), x(j), w(k)) for the rows k above j in the band
    for (int j = n - 1; j >= 1; j--)
    {
        const double xj = w[j];
        const int first = max(0, j - hb);
        for (int k = first + tid; k < j; k += MG_NT) w[k] = __builtin_fma(-a.Lc[(size_t)k * ld + (j - k)], xj, w[k]);
        __syncthreads();
    }
    if (a.nd) { for (int i = tid; i < n; i += MG_NT) a.xs[i] = w[i]; return; }      // the separator system's solution
    finish_solution(a, [&](int i) { return w[i]; }, tid, MG_NT, m);
}

} // namespace

namespace lvkmesh {

void launch_assemble(unsigned blocks, hipStream_t stream, const MeshArgs& a) { hipLaunchKernelGGL(k_mesh_assemble, dim3(blocks), dim3(128), 0, stream, a); }
void launch_prepare(hipStream_t stream, const MeshArgs& a) { hipLaunchKernelGGL(k_mesh_prepare, dim3(216), dim3(256), 0, stream, a); }
void launch_solve(unsigned blocks, hipStream_t stream, const MeshArgs& a) { hipLaunchKernelGGL(k_mesh_solve, dim3(blocks), dim3(MS_NT), 0, stream, a); }
void launch_backsolve(unsigned blocks, hipStream_t stream, const MeshArgs& a) { hipLaunchKernelGGL(k_mesh_backsolve, dim3(blocks), dim3(MB_NT), 0, stream, a); }
void launch_sep_assemble(unsigned blocks, hipStream_t stream, const MeshArgs& a) { hipLaunchKernelGGL(k_nd_sep_assemble, dim3(blocks), dim3(64), 0, stream, a); }
void launch_sep_assemble_tiled(unsigned blocks, hipStream_t stream, const MeshArgs& a, int tiles_per_col) { hipLaunchKernelGGL(k_nd_sep_assemble_tiled, dim3(blocks), dim3(256), 0, stream, a, tiles_per_col); }
void launch_solve_generic(hipStream_t stream, const MeshArgs& a) { hipLaunchKernelGGL(k_mesh_solve_generic, dim3(1), dim3(MG_NT), 0, stream, a); }
void launch_backsolve_generic(hipStream_t stream, const MeshArgs& a) { hipLaunchKernelGGL(k_mesh_backsolve_generic, dim3(1), dim3(MG_NT), 0, stream, a); }

} // namespace lvkmesh
