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
#include "lvk_hip_internal.hpp"
#include "host_logic.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

#ifndef LVK_MESH_TB
#define LVK_MESH_TB 8
#endif
#ifndef LVK_MESH_WIN_WAVES
#define LVK_MESH_WIN_WAVES 3
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
constexpr int MS_RING = 192, MS_RPITCH = 196;   // a staged row: L(j, k) at slot k mod 192 (three 64-row blocks cover the band); pitch: rows 4 slots apart in the banks
static_assert(MS_HB_MAX + 1 <= 128, "three 64-row blocks cover the rows a pivot row reaches");
constexpr int MS_BANDS = (MS_HB_MAX + MS_TB) / MS_TB;                      // bands of MS_TB band offsets
constexpr int MS_PF = (MS_BANDS * MS_CA * MS_TB + 63) / 64;                // prefetched entries per lane of the forward wavefront and column group
constexpr int ms_tiles(int hb) { int t = 0; for (int b = 0; b < (hb + MS_TB) / MS_TB; b++) t += (hb - MS_TB * b + 3) / MS_CA; return t; }
static_assert(ms_tiles(MS_HB_MAX) <= MS_BULK, "one register tile per window thread");
static_assert(MS_BANDS * MS_TB <= 128 && MS_HB_MAX < 128, "a column is two registers per lane of the chain");

// One block of the nested dissection (oracle S5'): a band system of its own -- the block's vertex rows, then its separators --
// of which only the first n_elim pivots are eliminated.
struct MeshBlockDev
{
    int n, hb, n_elim, nbands;
    double* N; double* g0; double* wz; double* Lc; double* Rc; double* T;      // Rc: unscaled pivot columns (layout of Lc); T: trailing (n - n_elim)^2 window, row major
    const int* xs_of;                       // separator position n_elim + q -> index into the separator system's solution
    const int* nat_of;                      // own position -> natural unknown index
};

struct MeshArgs
{
    // ---- nested dissection (nd != 0): kernels launched with one workgroup per block take n, hb, ... and the arrays from blocks[blockIdx.x]
    int nd, nblocks, n_elim, n_nat;         // n_elim: pivots to eliminate (n for a whole system); n_nat: unknowns of the whole mesh
    const MeshBlockDev* blocks;
    double* Rc; double* T;                  // (per block, see MeshBlockDev)
    const int* ndst; const int* gdst;       // k_mesh_prepare: natural band entry / unknown -> position in the blocks' arrays (or -1)
    const int* ssrc; const int* gsrc; int s_entries, ns;      // k_nd_sep_assemble: separator band entry / unknown -> its (<= 2) sources in T / wz of the blocks
    const double* Tall; const double* wzall;
    double* xs;                             // separator solution (binary64)
    const double* sep_wz; const double* sep_Lc; int sep_hb, fuse_sep;      // k_mesh_backsolve over the blocks: the separator system's backward substitution runs inside (fuse_sep)
    double* X;                              // solution in natural order (binary64), gathered from the blocks
    const int* sep_nat;                     // separator unknown -> natural index
    unsigned* ticket;                       // last-block-done counter of the block-parallel kernels
    int cols, rows, n, hb, nbands;          // nbands: bands of MS_TB band offsets covering 0 .. hb
    const double* stat;                     // static band, column layout: entry (i, k), k <= i <= k + hb, at [k * (hb + 1) + (i - k)]
    long long* Nq; long long* gq;           // Q32 sums of the feature rows (same layout as stat / one per unknown)
    double* N; double* g0;                  // the assembled system (k_mesh_prepare)
    double* wz;                             // D^-1 L^-1 g, from k_mesh_solve to k_mesh_backsolve
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
__host__ __device__ __forceinline__ int band_groups(int hb, int b) { return (hb - MS_TB * b + 3) / MS_CA; }

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
constexpr int NT_TILE = 16, NT_JMAX = MS_HB_MAX + NT_TILE;
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
constexpr int MB_NT = 64 + 512;
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
constexpr int MG_NT = 1024;
constexpr int MG_HB_MAX = 1023;             // LDS copies of the pivot column (2 x 8 KB); motion_resolution up to 167 columns

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

// ---- host side ------------------------------------------------------------------------------------------------------------------
struct lvk_mesh_solver_dev
{
    lvk_hip_ctx* ctx = nullptr;
    int cols = 0, rows = 0, n = 0, hb = 0;
    float ts_gen = 0.0f;
    double* d_stat = nullptr; long long* d_acc = nullptr;   // d_acc: Nq (n * ld) then gq (n), one allocation, one memset per solve
    float* d_mesh = nullptr; double* d_Lc = nullptr; double* d_N = nullptr;      // d_N: band then right-hand side
    int* d_flags = nullptr;
    bool generic = false;                   // k_mesh_solve_generic / k_mesh_backsolve_generic (meshes outside the register-window solver's shapes)
    // nested dissection (8 .. 16 columns, >= 9 rows): the blocks' band systems and the separator system, all built once per configuration
    bool nd = false;
    int nblocks = 0, ns = 0, hbs = 0, s_entries = 0;
    bool sep_generic = false;
    MeshBlockDev* d_blocks = nullptr;
    double* d_nd = nullptr;                 // one allocation: per block N | g | wz | Lc | Rc | T, then the separator system S | gs | wzs | Lcs | xs, then X
    int* d_ndi = nullptr;                   // one allocation: ndst | gdst | ssrc | gsrc | sep_nat | per block xs_of, nat_of
    size_t off_Nall = 0, off_wzall = 0, off_Tall = 0, off_S = 0, off_gs = 0, off_wzs = 0, off_Lcs = 0, off_xs = 0, off_X = 0;
    size_t ioff_ndst = 0, ioff_gdst = 0, ioff_ssrc = 0, ioff_gsrc = 0, ioff_sepnat = 0;
    unsigned* d_ticket = nullptr;
};

static bool nd_applies(int cols, int rows) { return cols >= 8 && cols <= 16 && rows >= 9; }     // oracle S5' (the register-window kernels' column range)

void lvk_mesh_solver_free(lvk_mesh_solver_dev* s)
{
    if (!s) return;
    void* dev[] = {s->d_stat, s->d_acc, s->d_mesh, s->d_Lc, s->d_N, s->d_flags, s->d_blocks, s->d_nd, s->d_ndi, s->d_ticket};
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
    // the register-window solver: meshes up to 16 columns (half bandwidth 103) and 16 x 64 vertices, every band with a tile that hands its
    // columns to the chain; everything else (17 x 17, 32 x 32, ...) takes the generic kernels -- same specification, same bits
    const bool force_generic = std::getenv("LVK_HIP_MESH_GENERIC") != nullptr;      // tests: the generic kernels on the preset's mesh
    const bool fast = host.hb() <= MS_HB_MAX && host.n() <= MS_N_MAX && host.n() >= 4 && band_groups(host.hb(), (host.hb() + MS_TB) / MS_TB - 1) >= 1;
    LVK_HIP_REQUIRE(ctx, host.hb() <= MG_HB_MAX);                           // motion_resolution beyond 167 columns
    auto* s = new lvk_mesh_solver_dev();
    s->ctx = ctx; s->cols = cols; s->rows = rows; s->n = host.n(); s->hb = host.hb(); s->ts_gen = temporal;
    s->generic = !fast || force_generic;
    const size_t band = (size_t)s->n * (s->hb + 1);
    auto fail = [&](hipError_t e) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMalloc((void**)&s->d_stat, band * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_acc, (band + s->n) * sizeof(long long))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_mesh, s->n * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_Lc, (band + MS_NT) * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_N, (band + 2 * (size_t)s->n) * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_flags, sizeof(int))) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(s->d_stat, host.static_band().data(), band * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->d_mesh, 0, s->n * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->d_Lc, 0, (band + MS_NT) * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->d_acc, 0, (band + s->n) * sizeof(long long))) != hipSuccess) return fail(e);       // kept clear by k_mesh_prepare
    if ((e = hipMemset(s->d_flags, 0, sizeof(int))) != hipSuccess) return fail(e);                            // kept clear by k_mesh_solve
    if (nd_applies(cols, rows) && !force_generic)
    {
        // ---- nested dissection (oracle S5'): separator rows 4, 8, ...; blocks = the rows between them, each followed by its separators
        const int W = 2 * cols, n = s->n, hb = s->hb, ld = hb + 1;
        std::vector<int> seps;
        for (int r = 4; r <= rows - 1; r += 4) seps.push_back(r);
        const int K = (int)seps.size();
        struct Blk { std::vector<int> rows; int own; int n, hb, n_elim, ns; size_t N, g, wz, Lc, Rc, T; size_t ixs, inat; };
        std::vector<Blk> blks;
        for (int k = 0; k <= K; k++)
        {
            const int first = k == 0 ? 0 : seps[k - 1] + 1, last = k < K ? seps[k] - 1 : rows - 1;
            if (first > last) continue;
            Blk b{};
            for (int r = first; r <= last; r++) b.rows.push_back(r);
            b.own = (int)b.rows.size();
            if (k > 0) b.rows.push_back(seps[k - 1]);
            if (k < K) b.rows.push_back(seps[k]);
            b.n = (int)b.rows.size() * W; b.n_elim = b.own * W; b.ns = b.n - b.n_elim; b.hb = std::min(b.n - 1, hb);
            blks.push_back(b);
        }
        s->nblocks = (int)blks.size(); s->ns = K * W; s->hbs = std::min(s->ns - 1, 2 * W - 1);
        bool blocks_fit = true;
        for (const Blk& b : blks)
            blocks_fit = blocks_fit && b.ns <= 255 && b.hb <= MS_HB_MAX && b.n <= MS_N_MAX && b.n >= 4 && band_groups(b.hb, (b.hb + MS_TB) / MS_TB - 1) >= 1;
        if (!blocks_fit) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_ARG, "mesh solver: a block of the nested dissection does not fit the register-window kernels"); }
        s->sep_generic = !(s->hbs <= MS_HB_MAX && s->ns <= MS_N_MAX && s->ns >= 4 && band_groups(s->hbs, (s->hbs + MS_TB) / MS_TB - 1) >= 1);
        if (s->sep_generic && s->hbs > MG_HB_MAX) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_ARG, "mesh solver: separator system too wide"); }
        // layout of the binary64 arena
        size_t at = 0;
        auto take = [&](size_t count) { const size_t o = at; at += (count + 1) & ~(size_t)1; return o; };
        s->off_Nall = at;
        for (Blk& b : blks) b.N = take((size_t)b.n * (b.hb + 1));
        for (Blk& b : blks) b.g = take(b.n);
        s->off_wzall = at;
        for (Blk& b : blks) b.wz = take(b.n);
        for (Blk& b : blks) b.Lc = take((size_t)b.n * (b.hb + 1) + MS_NT);
        for (Blk& b : blks) b.Rc = take((size_t)b.n * (b.hb + 1) + MS_NT);
        s->off_Tall = at;
        for (Blk& b : blks) b.T = take((size_t)b.ns * b.ns);
        const size_t lds = s->hbs + 1;
        s->off_S = take((size_t)s->ns * lds); s->off_gs = take(s->ns); s->off_wzs = take(s->ns); s->off_Lcs = take((size_t)s->ns * lds + MS_NT);
        s->off_xs = take(s->ns); s->off_X = take(n);
        const size_t doubles = at;
        // index tables
        std::vector<int> tab;
        auto itake = [&](size_t count) { const size_t o = tab.size(); tab.resize(o + count, -1); return o; };
        s->ioff_ndst = itake((size_t)n * ld); s->ioff_gdst = itake(n);
        s->s_entries = s->ns * (int)lds;
        s->ioff_ssrc = itake(2 * (size_t)s->s_entries); s->ioff_gsrc = itake(2 * (size_t)s->ns); s->ioff_sepnat = itake(s->ns);
        for (Blk& b : blks) { b.ixs = itake(std::max(b.ns, 1)); b.inat = itake(b.n_elim); }
        // where every vertex row sits: (block, slot) of its owning block (separators: the block ABOVE), and the slots it has as a separator
        auto sep_index = [&](int row) { for (int k = 0; k < K; k++) if (seps[k] == row) return k; return -1; };
        auto slot_of = [&](const Blk& b, int row) { for (size_t q = 0; q < b.rows.size(); q++) if (b.rows[q] == row) return (int)q; return -1; };
        auto owner_of = [&](int row) {                                    // block whose band holds the row's own entries and right-hand side
            for (int bi = 0; bi < (int)blks.size(); bi++)
            {
                const int q = slot_of(blks[bi], row);
                if (q < 0) continue;
                if (q < blks[bi].own || row > blks[bi].rows[0]) return bi;    // an own row, or the separator BELOW the block
            }
            return -1;
        };
        const std::vector<double>& stat = host.static_band();
        for (int k = 0; k < n; k++)
            for (int t = 0; t <= hb && k + t < n; t++)
            {
                const int i = k + t, ri = i / W, rk = k / W;
                // the block that holds both: an own row decides; two rows of the same separator go to its owner
                int bi = -1;
                for (int cand = 0; cand < (int)blks.size() && bi < 0; cand++)
                {
                    const int qi = slot_of(blks[cand], ri), qk = slot_of(blks[cand], rk);
                    if (qi < 0 || qk < 0) continue;
                    const bool own_i = qi < blks[cand].own, own_k = qk < blks[cand].own;
                    if (own_i || own_k) bi = cand;
                    else if (ri == rk && owner_of(ri) == cand) bi = cand;
                }
                const size_t src = (size_t)k * ld + t;
                if (bi < 0)
                {
                    if (stat[src] != 0.0) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_RUNTIME, "mesh solver: a constraint couples two blocks of the nested dissection"); }
                    continue;
                }
                const Blk& b = blks[bi];
                const int pi = slot_of(b, ri) * W + i % W, pk = slot_of(b, rk) * W + k % W;
                const int hi = std::max(pi, pk), lo = std::min(pi, pk);
                if (hi - lo > b.hb)
                {
                    if (stat[src] != 0.0) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_RUNTIME, "mesh solver: a constraint leaves a block's band"); }
                    continue;
                }
                tab[s->ioff_ndst + src] = (int)(b.N - s->off_Nall + (size_t)lo * (b.hb + 1) + (hi - lo));
            }
        for (int i = 0; i < n; i++)
        {
            const int bi = owner_of(i / W);
            tab[s->ioff_gdst + i] = (int)(blks[bi].g - s->off_Nall + (size_t)slot_of(blks[bi], i / W) * W + i % W);      // (g follows N in the arena)
        }
        for (int bi = 0; bi < (int)blks.size(); bi++)
        {
            const Blk& b = blks[bi];
            for (int p = 0; p < b.n_elim; p++) tab[b.inat + p] = b.rows[p / W] * W + p % W;
            for (int q = 0; q < b.ns; q++) tab[b.ixs + q] = sep_index(b.rows[(b.n_elim + q) / W]) * W + q % W;
            // this block's trailing window into the separator system (blocks in ascending order fill source 0, then source 1)
            for (int li = 0; li < b.ns; li++)
            {
                const int si = tab[b.ixs + li];
                int* gs = &tab[s->ioff_gsrc + 2 * (size_t)si];
                gs[gs[0] < 0 ? 0 : 1] = (int)(b.wz - s->off_wzall + b.n_elim + li);
                for (int lk = 0; lk <= li; lk++)
                {
                    const int sk = tab[b.ixs + lk];                         // (the separator above comes first: sk <= si)
                    int* ss = &tab[s->ioff_ssrc + 2 * ((size_t)sk * lds + (si - sk))];
                    ss[ss[0] < 0 ? 0 : 1] = (bi << 16) | (li << 8) | lk;
                }
            }
        }
        for (int k = 0; k < K; k++) for (int c = 0; c < W; c++) tab[s->ioff_sepnat + (size_t)k * W + c] = seps[k] * W + c;
        if ((e = hipMalloc((void**)&s->d_nd, doubles * sizeof(double))) != hipSuccess) return fail(e);
        if ((e = hipMemset(s->d_nd, 0, doubles * sizeof(double))) != hipSuccess) return fail(e);       // structural zeros of the blocks' bands, columns of L beyond n_elim
        if ((e = hipMalloc((void**)&s->d_ndi, tab.size() * sizeof(int))) != hipSuccess) return fail(e);
        if ((e = hipMemcpy(s->d_ndi, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
        std::vector<MeshBlockDev> hb_(blks.size());
        for (size_t bi = 0; bi < blks.size(); bi++)
        {
            const Blk& b = blks[bi];
            hb_[bi] = MeshBlockDev{b.n, b.hb, b.n_elim, (b.hb + MS_TB) / MS_TB, s->d_nd + b.N, s->d_nd + b.g, s->d_nd + b.wz, s->d_nd + b.Lc, s->d_nd + b.Rc,
                                   b.ns > 0 ? s->d_nd + b.T : nullptr, s->d_ndi + b.ixs, s->d_ndi + b.inat};
        }
        if ((e = hipMalloc((void**)&s->d_blocks, hb_.size() * sizeof(MeshBlockDev))) != hipSuccess) return fail(e);
        if ((e = hipMemcpy(s->d_blocks, hb_.data(), hb_.size() * sizeof(MeshBlockDev), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
        if ((e = hipMalloc((void**)&s->d_ticket, sizeof(unsigned))) != hipSuccess) return fail(e);
        if ((e = hipMemset(s->d_ticket, 0, sizeof(unsigned))) != hipSuccess) return fail(e);
        s->nd = true; s->generic = false;
    }
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
    a.stat = s->d_stat; a.Nq = s->d_acc; a.gq = s->d_acc + band; a.N = s->d_N; a.g0 = s->d_N + band; a.wz = s->d_N + band + s->n; a.mesh = s->d_mesh; a.Lc = s->d_Lc;
    a.fidx = (int*)d_scratch; a.fw = (float*)d_scratch + 4 * (size_t)std::max(n_pts, 1); a.p1 = d_p1; a.p2 = d_p2; a.count = d_count; a.n_pts = n_pts; a.min_samples = min_samples;
    a.region_w = region_w; a.region_h = region_h; a.ts_gen = s->ts_gen; a.ts_now = temporal_now; a.threshold = threshold;
    a.flags = s->d_flags; a.out_offsets = h_offsets; a.out_mask = h_mask; a.out_status = h_status;
    a.nd = 0; a.nblocks = 0; a.n_elim = s->n; a.n_nat = s->n; a.blocks = nullptr; a.Rc = nullptr; a.T = nullptr; a.ndst = nullptr; a.gdst = nullptr;
    a.sep_wz = nullptr; a.sep_Lc = nullptr; a.sep_hb = 0; a.fuse_sep = 0;
    a.ssrc = nullptr; a.gsrc = nullptr; a.s_entries = 0; a.ns = 0; a.Tall = nullptr; a.wzall = nullptr; a.xs = nullptr; a.X = nullptr; a.sep_nat = nullptr; a.ticket = nullptr;
    if (n_pts > 0) hipLaunchKernelGGL(k_mesh_assemble, dim3((unsigned)((n_pts + 127) / 128)), dim3(128), 0, stream, a);
    if (s->nd)
    {
        // nested dissection: scatter into the blocks' bands | the blocks side by side | separator system | the blocks' backward substitutions
        // side by side, the last one to finish runs phase 3
        MeshArgs p = a;
        p.nd = 1; p.ndst = s->d_ndi + s->ioff_ndst; p.gdst = s->d_ndi + s->ioff_gdst; p.N = s->d_nd + s->off_Nall; p.g0 = s->d_nd + s->off_Nall;
        hipLaunchKernelGGL(k_mesh_prepare, dim3(216), dim3(256), 0, stream, p);
        MeshArgs f = a;
        f.nd = 1; f.blocks = s->d_blocks; f.nblocks = s->nblocks; f.xs = s->d_nd + s->off_xs; f.X = s->d_nd + s->off_X; f.sep_nat = s->d_ndi + s->ioff_sepnat;
        f.ns = s->ns; f.ticket = s->d_ticket;
        hipLaunchKernelGGL(k_mesh_solve, dim3((unsigned)s->nblocks), dim3(MS_NT), 0, stream, f);
        MeshArgs q = a;
        q.nd = 1; q.n = s->ns; q.hb = s->hbs; q.nbands = (s->hbs + MS_TB) / MS_TB; q.n_elim = s->ns;
        q.N = s->d_nd + s->off_S; q.g0 = s->d_nd + s->off_gs; q.wz = s->d_nd + s->off_wzs; q.Lc = s->d_nd + s->off_Lcs; q.xs = s->d_nd + s->off_xs;
        q.ssrc = s->d_ndi + s->ioff_ssrc; q.gsrc = s->d_ndi + s->ioff_gsrc; q.s_entries = s->s_entries; q.ns = s->ns;
        q.Tall = s->d_nd + s->off_Tall; q.wzall = s->d_nd + s->off_wzall;
        q.blocks = s->d_blocks;
        if ((2 * s->cols) % NT_TILE == 0)
        {
            const int tiles_per_col = s->hbs / NT_TILE + 1;
            hipLaunchKernelGGL(k_nd_sep_assemble_tiled, dim3((unsigned)((s->ns / NT_TILE) * tiles_per_col + 1)), dim3(256), 0, stream, q, tiles_per_col);
        }
        else hipLaunchKernelGGL(k_nd_sep_assemble, dim3((unsigned)((s->s_entries + s->ns + 63) / 64)), dim3(64), 0, stream, q);
        q.blocks = nullptr;
        if (s->sep_generic)
        {
            hipLaunchKernelGGL(k_mesh_solve_generic, dim3(1), dim3(MG_NT), 0, stream, q);
            hipLaunchKernelGGL(k_mesh_backsolve_generic, dim3(1), dim3(MG_NT), 0, stream, q);
        }
        else
        {
            hipLaunchKernelGGL(k_mesh_solve, dim3(1), dim3(MS_NT), 0, stream, q);
            f.fuse_sep = 1; f.sep_wz = q.wz; f.sep_Lc = q.Lc; f.sep_hb = s->hbs;      // its backward substitution: inside the blocks' kernel
        }
        hipLaunchKernelGGL(k_mesh_backsolve, dim3((unsigned)s->nblocks), dim3(MB_NT), 0, stream, f);
        LVK_HIP_CHECK(ctx, hipGetLastError());
        return LVK_HIP_OK;
    }
    hipLaunchKernelGGL(k_mesh_prepare, dim3(216), dim3(256), 0, stream, a);
    if (s->generic)
    {
        hipLaunchKernelGGL(k_mesh_solve_generic, dim3(1), dim3(MG_NT), 0, stream, a);
        hipLaunchKernelGGL(k_mesh_backsolve_generic, dim3(1), dim3(MG_NT), 0, stream, a);
    }
    else
    {
        hipLaunchKernelGGL(k_mesh_solve, dim3(1), dim3(MS_NT), 0, stream, a);
        hipLaunchKernelGGL(k_mesh_backsolve, dim3(1), dim3(MB_NT), 0, stream, a);
    }
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
