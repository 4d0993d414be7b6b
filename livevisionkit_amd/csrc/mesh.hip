// Local motion estimate of the vector-field preset on the device.
//
// Replaces FrameTracker::estimate_local_motions (reference: LiveVisionKit/Vision/FrameTracker.cpp:200-321) with the constraint system of
// generate_mesh_constraints (:380-457).  The reference hands the sparse least-squares problem to Eigen::LeastSquaresConjugateGradient;
// this library solves the same problem directly through its normal equations N x = g:
//   static rows  -> a constant band matrix, built once per configuration on the host (host_logic.hpp, MeshSolverH::generate);
//   feature rows -> Q32 fixed-point sums accumulated with integer atomics (exact, order independent)          k_mesh_assemble
//   N = L D L^T  -> right-looking root-free band factorisation with reciprocal pivots, forward substitution carried along,
//                   every entry updated in pivot order (binary64, products and differences rounded separately)   k_mesh_solve, phase 1
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

constexpr int MS_NT = 256;                  // threads of k_mesh_solve
constexpr int MS_CA = 4, MS_TB = 13;        // register tile: columns x band offsets
constexpr int MS_HB_MAX = 103;              // widest band phase 1 holds in registers (meshes up to 16 columns)
constexpr int MS_WP_MAX = 108;              // window columns: hb + 1 + (MS_CA - 1), rounded up to a multiple of MS_CA
constexpr int MS_PAD = 8;                   // zeros in front of the LDS columns (negative relative indices of the pivot's own group)
constexpr int MS_LCOL = MS_PAD + 2 * MS_WP_MAX + 2 * MS_TB + 8;
constexpr double MS_Q = 4294967296.0;       // Q32
constexpr int MS_N_MAX = 3072;              // unknowns whose right-hand side fits the workgroup's LDS (16 x 96 vertices)
constexpr int MS_CHUNK = 16;                // rows of L staged per round of the backward substitution
constexpr int MS_PF = (MS_CA * (MS_HB_MAX + 1) + MS_NT - 1) / MS_NT;      // prefetched entries per thread and column group

struct MeshArgs
{
    int cols, rows, n, hb, wp;              // wp: window size in columns (multiple of MS_CA, >= hb + MS_CA)
    const double* stat;                     // static band, column layout: entry (i, k), k <= i <= k + hb, at [k * (hb + 1) + (i - k)]
    long long* Nq; long long* gq;           // Q32 sums of the feature rows (same layout as stat / one per unknown)
    double* N; double* g0;                  // the assembled system (k_mesh_prepare)
    float* mesh;                            // previous solution (absolute tracking-frame coordinates), updated on success
    double* Lc;                             // columns of L: L(i, k) at [k * (hb + 1) + (i - k)], plus MS_NT entries of dump area
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

struct FactorShared
{
    double raw[2][MS_LCOL];                 // pivot column, unscaled: raw[.][MS_PAD + x] = N(p + x, p); zeros elsewhere (two buffers, by step parity)
    double w[MS_N_MAX];                     // right-hand side: g, then z = L^-1 g, D^-1 z and finally the solution, in place
    double next[MS_CA][MS_HB_MAX + 1];      // phase 1: the columns that enter the window when the current group is done
    double lt[2][MS_CHUNK][MS_HB_MAX + 1];  // phase 2: rows of L, staged chunk by chunk
};

#if defined(LVK_MESH_TIMING) && LVK_MESH_TIMING > 1
__device__ long long g_mesh_phase[8];
#define MESH_T(k) do { const long long t_now = (long long)__builtin_readcyclecounter(); if (threadIdx.x == 0) g_mesh_phase[k] += t_now - t_last; t_last = t_now; } while (0)
#else
#define MESH_T(k) do { } while (0)
#endif

// One elimination step with the pivot at position CK of its column group.  ONE workgroup barrier per step: the pivot column goes through
// LDS unscaled, and every thread forms the reciprocal pivot and the scaled entries it needs itself (a division and 16 products per
// thread cost less than a second barrier plus another LDS round trip on the critical path: the steps are latency bound).  The global
// traffic of a step is one coalesced store per thread (column p of L) and, once per group, two coalesced loads per thread (the columns
// that enter the window next), both unconditional: the compiler's vmcnt bookkeeping stays exact and nothing ever waits for a store.
template <int CK>
__device__ __forceinline__ bool factor_step(const MeshArgs& a, FactorShared& s, double (&A)[MS_CA][MS_TB], double (&pf)[MS_PF],
                                            double& r_prev, int p, int cg, int tg, bool valid)
{
#if defined(LVK_MESH_TIMING) && LVK_MESH_TIMING > 1
    long long t_last = (long long)__builtin_readcyclecounter();
#endif
    const int wp = a.wp, hb = a.hb, ld = hb + 1;
    const int kp = p % wp, pg = kp / MS_CA;
    const int t0 = tg * MS_TB;
    const int tid = (int)threadIdx.x;
    const double* raw = s.raw[p & 1];
    // the columns p0 + wp .. p0 + wp + 3 take this group's slots when its last pivot is done: fetched now, parked in LDS two steps on
    if (CK == 0)
    {
#pragma unroll
        for (int q = 0; q < MS_PF; q++)
        {
            const int idx = tid + MS_NT * q, c = idx / (MS_HB_MAX + 1), t = idx - c * (MS_HB_MAX + 1), k = p + wp + c;
            const bool in = c < MS_CA && k < a.n && t <= hb && k + t < a.n;
            pf[q] = a.N[in ? (size_t)k * ld + t : 0];
            if (!in) pf[q] = 0.0;
        }
    }
    if (CK == 2)
    {
#pragma unroll
        for (int q = 0; q < MS_PF; q++)
        {
            const int idx = tid + MS_NT * q, c = idx / (MS_HB_MAX + 1), t = idx - c * (MS_HB_MAX + 1);
            if (c < MS_CA) s.next[c][t] = pf[q];
        }
    }
    MESH_T(0);
    lds_barrier();                                                      // the pivot column (published at the end of the previous step) is visible
    MESH_T(1);
    // everything this step reads from LDS, issued together
    const int ngroups = wp / MS_CA;
    const int s0 = cg == pg ? -CK : (MS_CA * ((cg - pg + ngroups) % ngroups) - CK);
    const int t = min(tid, hb + 1);                                     // threads beyond the band see the zero at offset hb + 1
    const double d = raw[MS_PAD];
    const double own = raw[MS_PAD + t];
    const double wp_row = s.w[p], wt_row = s.w[min(p + t, a.n - 1)];
    double rc[MS_CA], lw[MS_CA + MS_TB - 1];
#pragma unroll
    for (int ck = 0; ck < MS_CA; ck++) rc[ck] = raw[MS_PAD + s0 + ck];
#pragma unroll
    for (int j = 0; j < MS_CA + MS_TB - 1; j++) lw[j] = raw[MS_PAD + s0 + t0 + j];
    if (!(d > 0.0)) return false;                                       // uniform: every thread reads the same pivot
    const double r = 1.0 / d;
    // column p of L, forward substitution, D^-1 applied to the previous row's z on the way
    {
        const double l = (t >= 1 && t <= hb) ? own * r : 0.0;
        a.Lc[tid <= hb ? (size_t)p * ld + t : (size_t)a.n * ld + tid] = l;   // offset 0 unused; the other threads hit a dump area
        if (tid == 0) { if (p > 0) s.w[p - 1] = s.w[p - 1] * r_prev; r_prev = r; }
        else if (tid <= hb && p + t < a.n) s.w[p + t] = wt_row - l * wp_row;
    }
    MESH_T(2);
    // every entry the pivot touches: N(k + t, k) -= L(k + t, p) * N(k, p), k = p + sc.  The column of the next pivot goes first and
    // is published at once (the other buffer), so that its way through LDS overlaps the rest of the update.
    if (valid)
    {
#pragma unroll
        for (int ck = 0; ck < MS_CA; ck++) if (cg == pg && ck <= CK) rc[ck] = 0.0;
#pragma unroll
        for (int j = 0; j < MS_CA + MS_TB - 1; j++) lw[j] = lw[j] * r;
        double* raw_next = s.raw[(p + 1) & 1];
        constexpr int NK = (CK + 1) % MS_CA;                            // position of the next pivot in ITS group
        const int npg = ((p + 1) % wp) / MS_CA;
#pragma unroll
        for (int ti = 0; ti < MS_TB; ti++) A[NK][ti] = A[NK][ti] - lw[NK + ti] * rc[NK];
        if (CK == MS_CA - 1 && cg == pg)
        {
            // this group's slots are taken over by the columns p0 + wp ..: none of them is the next pivot (that one is in group npg)
        }
        if (cg == npg)
#pragma unroll
            for (int ti = 0; ti < MS_TB; ti++)
                if (t0 + ti <= hb) raw_next[MS_PAD + t0 + ti] = A[NK][ti];
#pragma unroll
        for (int ck = 0; ck < MS_CA; ck++)
            if (ck != NK)
#pragma unroll
                for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = A[ck][ti] - lw[ck + ti] * rc[ck];
        if (CK == MS_CA - 1 && cg == pg)
#pragma unroll
            for (int ck = 0; ck < MS_CA; ck++)
#pragma unroll
                for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = (t0 + ti <= hb) ? s.next[ck][t0 + ti] : 0.0;
    }
    MESH_T(3);
    return true;
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
    const int n = a.n, hb = a.hb, ld = hb + 1, wp = a.wp;
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
    const int TG = (hb + MS_TB) / MS_TB;                                // band offsets 0 .. hb in groups of MS_TB
    const int CG = wp / MS_CA;
    const int cg = tid / TG, tg = tid % TG;
    const bool valid = cg < CG;
    double A[MS_CA][MS_TB], pf[MS_PF];
#pragma unroll
    for (int ck = 0; ck < MS_CA; ck++)
#pragma unroll
        for (int ti = 0; ti < MS_TB; ti++) A[ck][ti] = valid ? load_entry(a, MS_CA * cg + ck, tg * MS_TB + ti) : 0.0;
#pragma unroll
    for (int q = 0; q < MS_PF; q++) pf[q] = 0.0;
    for (int i = tid; i < 2 * MS_LCOL; i += MS_NT) (&s.raw[0][0])[i] = 0.0;
    for (int i = tid; i < n; i += MS_NT) s.w[i] = a.g0[i];
    __syncthreads();
    // the first pivot column
    if (valid && cg == 0)
#pragma unroll
        for (int ti = 0; ti < MS_TB; ti++)
            if (tg * MS_TB + ti <= hb) s.raw[0][MS_PAD + tg * MS_TB + ti] = A[0][ti];
    __syncthreads();
#ifdef LVK_MESH_TIMING
    const long long tm1 = wall_clock64();
#endif
    bool ok = true;
    double r_prev = 0.0;
    for (int p0 = 0; p0 < n && ok; p0 += MS_CA)
    {
        ok = factor_step<0>(a, s, A, pf, r_prev, p0, cg, tg, valid);
        if (ok && p0 + 1 < n) ok = factor_step<1>(a, s, A, pf, r_prev, p0 + 1, cg, tg, valid);
        if (ok && p0 + 2 < n) ok = factor_step<2>(a, s, A, pf, r_prev, p0 + 2, cg, tg, valid);
        if (ok && p0 + 3 < n) ok = factor_step<3>(a, s, A, pf, r_prev, p0 + 3, cg, tg, valid);
    }
    if (!ok) { if (tid == 0) *a.out_status = 3; return; }
    if (tid == 0) s.w[n - 1] = s.w[n - 1] * r_prev;
    __syncthreads();
#ifdef LVK_MESH_TIMING
    const long long tm2 = wall_clock64();
#endif

    // ---- phase 2: L^T x = w column by column: x(j) = w(j); w(k) -= L(j, k) x(j) for the rows k above j in the band.
    // One wavefront walks the chain with the rows in flight in registers; all threads stage the rows of L it needs, one chunk ahead
    // (the loads of chunk c + 1 are in flight while the chain runs over chunk c).  L is stored by columns: a chunk of rows is a set of
    // short contiguous column segments.
    const int nchunks = (n + MS_CHUNK - 1) / MS_CHUNK;
    constexpr int PER = ((MS_HB_MAX + MS_CHUNK) * MS_CHUNK + MS_NT - 1) / MS_NT;
    double stage[PER];
    auto fetch = [&](int c) {
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
    auto commit = [&](int c) {
        const int ilo = n - (c + 1) * MS_CHUNK;
#pragma unroll
        for (int q = 0; q < PER; q++)
        {
            const int idx = tid + MS_NT * q, kk = idx / MS_CHUNK, rr = idx - kk * MS_CHUNK, t = (ilo + rr) - (ilo - hb + kk);
            if (kk < hb + MS_CHUNK && t >= 1 && t <= hb) s.lt[c & 1][MS_CHUNK - 1 - rr][t] = stage[q];      // row index counted from the chunk's top row
        }
    };
    fetch(0); commit(0);
    __syncthreads();
    // rows of the 64-row blocks B, B - 1, B - 2 (B = the block of row n - 1), one per lane of wavefront 0
    const int nblocks = (n + 63) / 64;
    auto block_rows = [&](int b) -> double { const int i = 64 * b + tid; return (b >= 0 && i < n) ? s.w[i] : 0.0; };
    double cur = 0.0, p1 = 0.0, p2 = 0.0;
    if (tid < 64) { cur = block_rows(nblocks - 1); p1 = block_rows(nblocks - 2); p2 = block_rows(nblocks - 3); }
    int B = nblocks - 1;
    for (int c = 0; c < nchunks; c++)
    {
        if (c + 1 < nchunks) fetch(c + 1);
        if (tid < 64)
        {
            const int top = n - 1 - c * MS_CHUNK;
            for (int r = 0; r < MS_CHUNK && top - r >= 0; r++)
            {
                const int j = top - r, lj = j & 63;
                const double xj = readlane64(cur, lj);
                const int tc = lj - tid, t1 = tc + 64, t2 = tc + 128;   // band offsets of this lane's rows in cur / p1 / p2
                const double l0 = s.lt[c & 1][r][min(max(tc, 0), hb)], l1 = s.lt[c & 1][r][min(t1, hb)], l2 = s.lt[c & 1][r][min(t2, hb)];
                if (tc >= 1 && tc <= hb) cur = cur - l0 * xj;
                if (t1 <= hb) p1 = p1 - l1 * xj;
                if (t2 <= hb) p2 = p2 - l2 * xj;
                if (lj == 0)
                {
                    // block B is final: x of its rows; the registers move up one block
                    if (64 * B + tid < n) s.w[64 * B + tid] = cur;
                    cur = p1; p1 = p2; p2 = block_rows(B - 3);
                    B--;
                }
            }
        }
        if (c + 1 < nchunks) commit(c + 1);
        __syncthreads();
    }
#ifdef LVK_MESH_TIMING
    const long long tm3 = wall_clock64();
    if (tid == 0)
    {
        printf("mesh solve: init %lld, factor %lld, backsolve %lld (100 MHz ticks), n %d hb %d\n", tm1 - tm0, tm2 - tm1, tm3 - tm2, n, hb);
#if LVK_MESH_TIMING > 1
        printf("  per step (shader cycles, thread 0): prefetch %lld, barrier %lld, reads + pivot + column %lld, update %lld\n",
               g_mesh_phase[0] / n, g_mesh_phase[1] / n, g_mesh_phase[2] / n, g_mesh_phase[3] / n);
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
    int cols = 0, rows = 0, n = 0, hb = 0, wp = 0;
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
    LVK_HIP_REQUIRE(ctx, host.hb() <= MS_HB_MAX && host.n() <= MS_N_MAX);      // meshes wider than 16 columns / beyond 16 x 96: not supported by the device solver
    auto* s = new lvk_mesh_solver_dev();
    s->ctx = ctx; s->cols = cols; s->rows = rows; s->n = host.n(); s->hb = host.hb(); s->ts_gen = temporal;
    s->wp = ((s->hb + MS_CA + MS_CA - 1) / MS_CA) * MS_CA;
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
    a.cols = s->cols; a.rows = s->rows; a.n = s->n; a.hb = s->hb; a.wp = s->wp;
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
