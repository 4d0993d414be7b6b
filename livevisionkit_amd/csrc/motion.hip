// Robust global motion estimation for gfx950: deterministic RANSAC over a fixed hypothesis schedule, one
// wavefront per hypothesis with exact integer inlier voting, then a single-wave local optimisation whose
// least-squares sums run in a fixed lane order (so the result is reproducible bit for bit).
//
// Replaces cv::findHomography(tracked, matched, mask, UsacParams) and cv::estimateAffinePartial2D as called by
// FrameTracker::estimate_global_motion (reference: LiveVisionKit/Vision/FrameTracker.cpp:325-375).  The
// algorithm is the specification of SURVEY.md Appendix A.8 (OpenCV's USAC is not restated):
//   128 hypotheses, minimal sample from a SplitMix64 stream keyed by the hypothesis index, 8x8 solve /
//   closed-form similarity, score = sum floor(1024 * max(0, 1 - e^2/t^2)), best score (lowest index on ties),
//   up to 3 least-squares refits accepted while the score strictly improves.  All math binary64, no contraction.
#include "lvk_hip_internal.hpp"

#include <utility>

namespace {

constexpr int K_HYPOTHESES = 128;
constexpr int LO_ROUNDS = 3;
constexpr int NT = 256;              // threads per block of k_ransac_hypotheses; also the number of strided partials of the least-squares sums (the specification)
// k_ransac_finalize runs FT threads: 8 waves share the 44 (14) sums of a refit -- wave w owns the sums w, w + 8, ... and still computes all NT
// partials of each (four slots per lane), so every total keeps the specification's pairing and order bit for bit; the sums phase of a
// round is ~5.5 instead of 11 sums deep per wave, the scoring passes run two pairs per SIMD lane-slot instead of one.
#ifndef LVK_FINALIZE_THREADS
#define LVK_FINALIZE_THREADS 512
#endif
constexpr int FT = LVK_FINALIZE_THREADS;
static_assert(FT == 256 || FT == 512, "4 or 8 waves");
constexpr int LDS_POINTS = 2048;     // point pairs staged in LDS (16 B each); larger sets are read from global memory

// threadIdx.x through an opaque move.  The local-optimisation rounds of k_ransac_finalize are one loop around ~7 000 inlined instructions; what
// the compiler can derive from the thread index alone (lane predicates, LDS addresses per slot, permute addresses, per-lane selects of the
// normalisation) it computes once in front of that loop and keeps in registers across all of it -- 109-113 VGPRs, where two waves per SIMD
// next to the remap may take 96.  Re-deriving those few values where they are used costs a handful of cheap instructions per round.
__device__ __forceinline__ int tid_now()
{
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& s)
{
    s += 0x9E3779B97F4A7C15ull;
    unsigned long long z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ bool draw_sample(int h, int n, int m, int* idx)
{
    unsigned long long s = 0x4C564B31ull ^ ((unsigned long long)(h + 1) * 0xD1B54A32D192ED03ull);
    int got = 0;
    for (int draw = 0; draw < 32 && got < m; draw++)
    {
        const int c = (int)((splitmix64(s) >> 32) % (unsigned long long)n);
        bool dup = false;
        for (int j = 0; j < got; j++) dup = dup || (idx[j] == c);
        if (!dup) idx[got++] = c;
    }
    return got == m;
}

__device__ __forceinline__ double lane_value(double v, int src_lane)          // wave-uniform copy of one lane's binary64 value
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffu), src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Gaussian elimination with partial pivoting of the N x N system (A, b) in LDS, N <= 8, executed by wave 0 of the block entirely
// in registers: lane q holds column q of A (lane N: the right-hand side), pivots and multipliers travel by v_readlane, so a
// pivot step needs neither LDS nor a barrier.  Every element sees exactly the operations of the sequential algorithm in the same
// order (f = A[j][i] * (1 / A[i][i]); A[j][q] -= f * A[i][q]; back substitution s -= A[i][q] * x[q], q ascending, x = s * (1 / A[i][i])), so the result
// is bit-identical to it.  Must be called by every thread of the block (one block barrier at the end); the solution replaces b,
// all threads return the same verdict.
template <int N>
__device__ bool solve_n(double* A, double* b, int* s_verdict)
{
    if (tid_now() < 64)
    {
        const int q = tid_now();
        double a[N];
#pragma unroll
        for (int r = 0; r < N; r++) a[r] = q < N ? A[r * N + q] : (q == N ? b[r] : 0.0);
        bool ok = true;
        double rcp[N];
#pragma unroll
        for (int i = 0; i < N; i++) rcp[i] = 0.0;
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            if (ok)
            {
                // pivot search in column i: every lane scans its own column (no cross-lane traffic), lane i's verdict is broadcast
                int piv = i;
                double best = fabs(a[i]);
#pragma unroll
                for (int j = i + 1; j < N; j++)
                {
                    const double v = fabs(a[j]);
                    if (v > best) { best = v; piv = j; }
                }
                piv = __builtin_amdgcn_readlane(piv, i);
                best = lane_value(best, i);
                if (best < 1e-10) ok = false;
                else
                {
#pragma unroll
                    for (int r = i + 1; r < N; r++)
                        if (piv == r) { const double t = a[i]; a[i] = a[r]; a[r] = t; }          // wave-uniform row swap
                    const double inv = 1.0 / lane_value(a[i], i);
                    rcp[i] = inv;
#pragma unroll
                    for (int j = i + 1; j < N; j++)
                    {
                        const double f = lane_value(a[j], i) * inv;
                        a[j] = a[j] - f * a[i];                 // (the columns q <= i of the rows below the pivot are never read again: no mask)
                    }
                }
            }
        }
        if (ok)
        {
            double x[N];
#pragma unroll
            for (int i = N - 1; i >= 0; i--)
            {
                double sum = lane_value(a[i], N);
#pragma unroll
                for (int c = i + 1; c < N; c++) sum = sum - lane_value(a[i], c) * x[c];
                x[i] = sum * rcp[i];                            // the reciprocal pivot of the elimination
            }
            if (q == 0)
            {
#pragma unroll
                for (int i = 0; i < N; i++) b[i] = x[i];
            }
        }
        if (q == 0) *s_verdict = ok ? 1 : 0;
    }
    __syncthreads();
    return *s_verdict != 0;
}

__device__ __forceinline__ double reproj_err2(const double* H, double x, double y, double u, double v)
{
    const double w = H[6] * x + H[7] * y + H[8];
    if (fabs(w) < 1e-12) return 1e300;
    const double px = (H[0] * x + H[1] * y + H[2]) / w, py = (H[3] * x + H[4] * y + H[5]) / w;
    const double ex = px - u, ey = py - v;
    return ex * ex + ey * ey;
}

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Exact integer sum over the whole block of BT threads (order free).  scratch: BT / 64 int64 in LDS.
template <int BT>
__device__ __forceinline__ long long block_sum_ll(long long v, long long* scratch)
{
    v = wave_sum_ll(v);
    __syncthreads();
    if ((tid_now() & 63) == 0) scratch[tid_now() >> 6] = v;
    __syncthreads();
    long long t = 0;
#pragma unroll
    for (int w = 0; w < BT / 64; w++) t += scratch[w];
    return t;
}

__device__ __forceinline__ double wave_sum_f64(double v)          // xor butterfly 32,16,...,1: own + partner
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = v + __shfl_xor(v, o);
    return v;
}

// Scores model H over all pairs with the whole wave; optionally writes the inlier mask. Returns (score, #inliers) on every lane.
template <int BT>
__device__ long long score_model(const double* H, const float2* __restrict__ p1, const float2* __restrict__ p2, int n, double t2,
                                 uint8_t* mask, int* ninl, long long* scratch)
{
    long long score = 0; long long cnt = 0;
    for (int i = tid_now(); i < n; i += BT)
    {
        const float2 a = p1[i], b = p2[i];
        const double e2 = reproj_err2(H, (double)a.x, (double)a.y, (double)b.x, (double)b.y);
        const bool in = e2 <= t2;
        if (in) { score += (long long)((1.0 - e2 / t2) * 1024.0); cnt++; }
        if (mask) mask[i] = in ? 1 : 0;
    }
    // one block reduction for both: the score of a pair is below 2^10, so the sum over <= 2^22 pairs stays below bit 40
    const long long both = block_sum_ll<BT>(score + (cnt << 40), scratch);
    if (ninl) *ninl = (int)(both >> 40);
    return both & ((1ll << 40) - 1);
}

// Whole-wave: lane 0 assembles the system / closed form, all lanes take part in the solve.
__device__ bool model_from_sample(bool full, const float2* __restrict__ p1, const float2* __restrict__ p2, const int* idx,
                                  double* A, double* b, double* H)
{
    __shared__ int s_solved;
    const int lane = threadIdx.x;
    if (full)
    {
        if (lane == 0)
        for (int i = 0; i < 4; i++)
        {
            const double x = p1[idx[i]].x, y = p1[idx[i]].y, u = p2[idx[i]].x, v = p2[idx[i]].y;
            double* r0 = A + i * 8; double* r1 = A + (i + 4) * 8;
            r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * u; r0[7] = -y * u; b[i] = u;
            r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; b[i + 4] = v;
        }
        __syncthreads();
        if (!solve_n<8>(A, b, &s_solved)) return false;
        if (lane < 8) H[lane] = b[lane];
        if (lane == 8) H[8] = 1.0;
        __syncthreads();
        return true;
    }
    const double x0 = p1[idx[0]].x, y0 = p1[idx[0]].y, x1 = p1[idx[1]].x, y1 = p1[idx[1]].y;
    const double u0 = p2[idx[0]].x, v0 = p2[idx[0]].y, u1 = p2[idx[1]].x, v1 = p2[idx[1]].y;
    const double dx = x1 - x0, dy = y1 - y0, ex = u1 - u0, ey = v1 - v0;
    const double d2 = dx * dx + dy * dy;
    if (d2 < 1e-10) return false;
    const double a = (dx * ex + dy * ey) / d2, bb = (dx * ey - dy * ex) / d2;
    if (lane == 0)
    {
        H[0] = a; H[1] = -bb; H[2] = u0 - (a * x0 - bb * y0);
        H[3] = bb; H[4] = a;  H[5] = v0 - (bb * x0 + a * y0);
        H[6] = 0; H[7] = 0;   H[8] = 1;
    }
    __syncthreads();
    return true;
}

// ---- fast_filter on the GPU (see lvk_launch_match_compact) ---------------------------------------------------------------------------
// The host algorithm (Functions/Container.tpp:97-121) walks k = n-1 .. 0 and, for every dropped k, swaps element k with the last element
// of the still-kept prefix.  Closed form of the result: with r dropped elements, m = n - r, the kept elements below m never move; the hole
// with descending rank j (1 = highest dropped index) receives what position n - j holds at that moment, which is that position's own
// element if it was kept, or else whatever was moved into it when IT was a hole (rank i < j, i.e. the content of position n - i) -- a
// chain that ends at a kept tail element.
constexpr int CMP_CAP = 4096;
// One block of CMP_NT threads over n <= EPT CMP_NT pairs: fills s_keep (the effective status flags) and s_above (per element, the number
// of dropped elements with a higher index) and returns m.  mirror: also copy the raw flow result to device-visible host memory.
template <int CMP_NT, int EPT = 4>
__device__ __forceinline__ int compact_plan(const float2* __restrict__ matched, const uint8_t* __restrict__ status, int n,
                                            const float2* __restrict__ und, float region_w, float region_h,
                                            unsigned short* s_above, uint8_t* s_keep, int* s_wave, bool mirror,
                                            float2* __restrict__ host_matched, uint8_t* __restrict__ host_status)
{
    static_assert(CMP_NT == 1024 || CMP_NT == 256, "EPT elements per thread: 4096 / 1024 pairs at 4, 2048 at 8 x 256");
    const int t = threadIdx.x;
    for (int i = t; i < n; i += CMP_NT)
    {
        uint8_t k = status[i];
        if (und)
        {
            // fused lens mode: a match whose lens-corrected positions leave the tracking region is dropped (not visible in the corrected frame)
            const float2 a = und[i], b = und[n + i];
            const bool inside = a.x >= 0.0f && a.x < region_w && a.y >= 0.0f && a.y < region_h && b.x >= 0.0f && b.x < region_w && b.y >= 0.0f && b.y < region_h;
            if (!inside) k = 0;
        }
        s_keep[i] = k;
        if (mirror) { host_status[i] = k; host_matched[i] = matched[i]; }
    }
    __syncthreads();
    // exclusive prefix count of dropped elements over the reversed index j = n - 1 - i; thread t owns j = EPT t .. EPT t + EPT - 1
    int loc[EPT], sum = 0;
#pragma unroll
    for (int q = 0; q < EPT; q++)
    {
        const int j = EPT * t + q;
        loc[q] = sum;
        sum += (j < n && !s_keep[n - 1 - j]) ? 1 : 0;
    }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if ((t & 63) >= o) inc += v; }
    if ((t & 63) == 63) s_wave[t >> 6] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < CMP_NT / 64; w++) { const int v = s_wave[w]; total += v; if (w < (t >> 6)) base += v; }
    const int excl = base + inc - sum;
#pragma unroll
    for (int q = 0; q < EPT; q++)
    {
        const int j = EPT * t + q;
        if (j < n) s_above[n - 1 - j] = (unsigned short)(excl + loc[q]);
    }
    __syncthreads();
    return n - total;
}
// the element that ends up at position i < m
__device__ __forceinline__ int compact_source(int i, int n, const unsigned short* s_above, const uint8_t* s_keep)
{
    if (s_keep[i]) return i;
    int p = n - ((int)s_above[i] + 1);
    while (!s_keep[p]) p = n - ((int)s_above[p] + 1);
    return p;
}

// One wavefront per hypothesis.  STAGED: the point pairs are first copied into LDS with one coalesced sweep, so the
// voting loop is not a chain of dependent global-memory round trips.
// What the fused variant needs besides: the raw flow result (prev | matched | status) and where the compacted pairs, their count and the
// host mirrors go (k_match_compact's arguments).
struct CompactArgs
{
    const float2* prev; const float2* matched; const uint8_t* status; const float2* und; float region_w, region_h;
    float2* p1; float2* p2; int* count; int* host_count; float2* host_matched; uint8_t* host_status;
    // the number of tracked points / the model choice when they were decided on the device (fast.hip k_fast_insert): n and `full` of the
    // launch are then only upper bound / placeholder
    const int* n_raw_dev; const int* full_dev;
};

// FUSED: fast_filter runs inside this kernel -- every block derives the compacted pairs straight into its LDS copy (the staging sweep
// it does anyway), block 0 also writes them, the count and the host mirrors out for k_ransac_finalize and the host.  One kernel and
// one kernel boundary less on the critical chain of a frame (k_match_compact: 3.5 us + 4-8 us of gaps around it).
template <bool STAGED, bool FUSED = false>
__global__ __launch_bounds__(NT)
void k_ransac_hypotheses(const float2* __restrict__ g1, const float2* __restrict__ g2, int n, const int* __restrict__ n_dev, double t2, int full,
                         double* __restrict__ hyp_H, long long* __restrict__ hyp_score, CompactArgs ca)
{
    LVK_TL(0);
    LVK_TRACKER_PRIORITY();
    static_assert(!FUSED || STAGED, "the fused variant compacts into the LDS copy");
    __shared__ double sA[64], sb[8], sH[9];
    __shared__ int s_ok;
    __shared__ long long s_scratch[NT / 64];
    __shared__ float2 s_p1[STAGED ? LDS_POINTS : 1], s_p2[STAGED ? LDS_POINTS : 1];
    if (ca.full_dev) full = *ca.full_dev;                   // the model choice was made on the device (fast.hip k_fast_insert)
    if (FUSED)
    {
        constexpr int EPT = LVK_COMPACT_RANSAC_MAX / NT;
        __shared__ unsigned short s_above[FUSED ? EPT * NT : 1];
        __shared__ uint8_t s_keep[FUSED ? EPT * NT : 1];
        __shared__ int s_wave[NT / 64];
        const bool first = blockIdx.x == 0;
        if (ca.n_raw_dev) n = min(max(*ca.n_raw_dev, 0), n);
        const int m = compact_plan<NT, EPT>(ca.matched, ca.status, n, ca.und, ca.region_w, ca.region_h, s_above, s_keep, s_wave, first, ca.host_matched, ca.host_status);
        const float2* pair_prev = ca.und ? ca.und : ca.prev;
        const float2* pair_next = ca.und ? ca.und + n : ca.matched;
        for (int i = threadIdx.x; i < m; i += NT)
        {
            const int src = compact_source(i, n, s_above, s_keep);
            const float2 a = pair_prev[src], b = pair_next[src];
            s_p1[i] = a; s_p2[i] = b;
            if (first) { ca.p1[i] = a; ca.p2[i] = b; }
        }
        if (first && threadIdx.x == 0) { *ca.count = m; *ca.host_count = m; }
        __syncthreads();
        n = m;
        if (n < (full ? 4 : 2)) { if (threadIdx.x == 0) hyp_score[blockIdx.x] = -1; return; }
    }
    else
    {
        if (n_dev) n = min(*n_dev, n);                      // pair count decided on the GPU (k_match_compact); n = capacity
        if (n < (full ? 4 : 2)) { if (threadIdx.x == 0) hyp_score[blockIdx.x] = -1; return; }
        if (STAGED)
        {
            for (int i = threadIdx.x; i < n; i += NT) { s_p1[i] = g1[i]; s_p2[i] = g2[i]; }
            __syncthreads();
        }
    }
    const float2* p1 = STAGED ? s_p1 : g1;
    const float2* p2 = STAGED ? s_p2 : g2;
    const int h = blockIdx.x;
    int idx[4];
    bool ok = draw_sample(h, n, full ? 4 : 2, idx);                 // uniform: every lane draws the same sample
    if (ok) ok = model_from_sample(full != 0, p1, p2, idx, sA, sb, sH);
    (void)s_ok;
    if (!ok) { if (threadIdx.x == 0) hyp_score[h] = -1; return; }
    const long long s = score_model<NT>(sH, p1, p2, n, t2, nullptr, nullptr, s_scratch);
    if (threadIdx.x == 0)
    {
        hyp_score[h] = s;
        for (int q = 0; q < 9; q++) hyp_H[h * 9 + q] = sH[q];
    }
}

// v[j] + v[j + o] for the lanes j < o of one wavefront (other lanes: unspecified), binary64: o = 32, 16 cross the 16-lane DPP
// rows (ds_bpermute), o = 8 ... 1 are DPP row shifts.
template <int O>
__device__ __forceinline__ double lane_plus(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    int lo = (int)(unsigned)(u & 0xffffffffu), hi = (int)(unsigned)(u >> 32);
    if (O >= 16)
    {
        const int src = ((tid_now() & 63) + O) << 2;
        lo = __builtin_amdgcn_ds_bpermute(src, lo); hi = __builtin_amdgcn_ds_bpermute(src, hi);
    }
    else
    {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + O, 0xf, 0xf, false);      // row_shl:O -- lane j reads lane j + O of its row
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + O, 0xf, 0xf, false);
    }
    return v + __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// Least-squares refit on the masked pairs.  Block-order sums (the specification): partial t, t = 0 .. NT-1, is the sum over the
// pairs i = t, t + NT, ... in increasing i; the NT partials are combined by the tree  v[j] += v[j + o]  for j < o, o = NT/2 ... 1.
// Execution: wave w owns the sums id = w, w + 4, ... and computes ALL 256 partials of them itself -- lane l carries the four
// partials l, l + 64, l + 128, l + 192 -- so the two upper tree levels are additions inside a thread, the six lower ones lane
// shifts inside the wave, and no partial ever crosses a wave: the reduction needs no LDS traffic and no block barrier.
// Pairing and order of every addition are those of the plain tree, so the totals are bit-identical to it.
constexpr int tri_row(int id, int n) { int a = 0; while (id >= n - a) { id -= n - a; a++; } return a; }
constexpr int tri_col(int id, int n) { int a = 0; while (id >= n - a) { id -= n - a; a++; } return a + id; }

// one term of sum ID for one point pair (ID is a compile-time constant so that r0 / r1 stay in registers).
// The rows are r0 = (x, y, 1, 0, 0, 0, -xu, -yu) and r1 = (0, 0, 0, x, y, 1, -xv, -yv): most products have a structural 0 or 1 in them.
// Without fast-math the compiler must still multiply by the zeros (0 * t is -0 or NaN for some t); here the inputs are finite, a term
// with a structural zero factor is +-0, adding +-0 to the other product leaves it unchanged unless that is a zero too, and an accumulator
// that starts at +0 can never become -0 -- so the structural zeros are dropped at compile time and every sum keeps its exact value
// (28 multiplications and 40 additions per pair instead of 88 and 88).
constexpr int row_kind0(int k) { return k < 2 ? 2 : (k == 2 ? 1 : (k < 6 ? 0 : 2)); }      // 0: structural zero, 1: one, 2: a variable
constexpr int row_kind1(int k) { return k < 3 ? 0 : (k < 5 ? 2 : (k == 5 ? 1 : 2)); }
template <int KA, int KC>
__device__ __forceinline__ double structured_product(double a, double c) { if constexpr (KA == 1) return c; else if constexpr (KC == 1) return a; else return a * c; }
template <int ID>
constexpr bool full_term_is_zero()
{
    if (ID < 36) { const int a = tri_row(ID, 8), c = tri_col(ID, 8); return (row_kind0(a) == 0 || row_kind0(c) == 0) && (row_kind1(a) == 0 || row_kind1(c) == 0); }
    return false;
}
template <int ID>
__device__ __forceinline__ double full_term(const double (&r0)[8], const double (&r1)[8], double u, double v)
{
    if constexpr (ID < 36)
    {
        constexpr int a = tri_row(ID, 8), c = tri_col(ID, 8);
        constexpr bool z0 = row_kind0(a) == 0 || row_kind0(c) == 0, z1 = row_kind1(a) == 0 || row_kind1(c) == 0;
        if constexpr (z0 && z1) return 0.0;
        else if constexpr (z1) return structured_product<row_kind0(a), row_kind0(c)>(r0[a], r0[c]);
        else if constexpr (z0) return structured_product<row_kind1(a), row_kind1(c)>(r1[a], r1[c]);
        else return structured_product<row_kind0(a), row_kind0(c)>(r0[a], r0[c]) + structured_product<row_kind1(a), row_kind1(c)>(r1[a], r1[c]);
    }
    else
    {
        constexpr int k = ID - 36;
        constexpr bool z0 = row_kind0(k) == 0, z1 = row_kind1(k) == 0;
        if constexpr (z1) return structured_product<row_kind0(k), 2>(r0[k], u);
        else if constexpr (z0) return structured_product<row_kind1(k), 2>(r1[k], v);
        else return structured_product<row_kind0(k), 2>(r0[k], u) + structured_product<row_kind1(k), 2>(r1[k], v);
    }
}
template <int ID>
__device__ __forceinline__ double partial_term(const double (&r0)[4], const double (&r1)[4], double u, double v)
{
    if constexpr (ID < 10) { constexpr int a = tri_row(ID, 4), c = tri_col(ID, 4); return r0[a] * r0[c] + r1[a] * r1[c]; }
    else return r0[ID - 10] * u + r1[ID - 10] * v;
}
constexpr int NW = FT / 64;          // waves of k_ransac_finalize: wave W owns the sums W, W + NW, W + 2 NW, ...
template <int W, int NS, int... K>
__device__ __forceinline__ void add_full_terms(double (&acc)[NS], const double (&r0)[8], const double (&r1)[8], double u, double v, bool on, std::integer_sequence<int, K...>)
{
    auto one = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (!full_term_is_zero<W + NW * k>()) acc[k] = acc[k] + (on ? full_term<W + NW * k>(r0, r1, u, v) : 0.0);      // (a sum of structural zeros stays +0)
    };
    (one(std::integral_constant<int, K>{}), ...);
}
template <int W, int NS, int... K>
__device__ __forceinline__ void add_partial_terms(double (&acc)[NS], const double (&r0)[4], const double (&r1)[4], double u, double v, bool on, std::integer_sequence<int, K...>)
{
    ((acc[K] = acc[K] + (on ? partial_term<W + NW * K>(r0, r1, u, v) : 0.0)), ...);
}
template <int W, int N, int NS, int... K>
__device__ __forceinline__ void store_totals(const double (&acc)[4][NS], int n_tri, double* A, double* b, std::integer_sequence<int, K...>)
{
    const int lane = tid_now() & 63;
    // level by level over ALL sums of this wave, so that the NS independent lane shifts of a level are in flight together (a
    // ds_bpermute round trip is ~100 cycles; sum after sum they would serialise into 6 x NS of them)
    double t[NS];
    ((t[K] = (acc[0][K] + acc[2][K]) + (acc[1][K] + acc[3][K])), ...);      // o = 128: (t, t + 128), (t + 64, t + 192); o = 64: their sum
    ((t[K] = lane_plus<32>(t[K])), ...);
    ((t[K] = lane_plus<16>(t[K])), ...);
    ((t[K] = lane_plus<8>(t[K])), ...);
    ((t[K] = lane_plus<4>(t[K])), ...);
    ((t[K] = lane_plus<2>(t[K])), ...);
    ((t[K] = lane_plus<1>(t[K])), ...);
    if (lane == 0)
    {
        auto one = [&](auto kc) {
            constexpr int k = decltype(kc)::value, id = W + NW * k;
            if constexpr (id < N * (N + 1) / 2) { constexpr int a = tri_row(id, N), c = tri_col(id, N); A[a * N + c] = t[k]; A[c * N + a] = t[k]; }
            else b[id - N * (N + 1) / 2] = t[k];
        };
        (one(std::integral_constant<int, K>{}), ...);
    }
}

// Full homography: 36 upper-triangle entries of the 8 x 8 normal matrix (ids 0 .. 35, row major) + 8 right-hand sides (36 .. 43).
// The loop runs over `it` (uniform trip count) with the lane's four partial slots side by side and NO branch on the mask: a masked-out
// or out-of-range pair contributes `+0.0` to every sum, which leaves an accumulator that is never -0 (it starts at +0, see above)
// bit for bit as it was.  Four independent pair computations per iteration instead of a divergent `continue` per pair: the LDS reads,
// conversions and products of one overlap the others' (in-kernel clocks, n = 700: 3.4 -> ~1.3 us per refit round).
template <int W>
__device__ __forceinline__ void refit_sums_full(const float2* __restrict__ p1, const float2* __restrict__ p2, int n, const uint8_t* mask,
                                                const double* par, double* A, double* b)
{
    const double cx = par[0], cy = par[1], sc = par[2];      // (read from LDS where they are used: kept live across the whole kernel they spill)
    constexpr int NS = (44 - W + NW - 1) / NW;       // ids W, W + NW, ... < 44  (4 waves: 11 each; 8 waves: 6, 6, 6, 6, 5, 5, 5, 5)
    const int lane = tid_now() & 63;
    double acc[4][NS];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int k = 0; k < NS; k++) acc[c][k] = 0.0;
    const int iters = (n + NT - 1) / NT;             // block-uniform
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
            const int i = lane + 64 * c + NT * it;
            const int ic = min(i, n - 1);            // (n >= 1: a refit needs inliers)
            const bool on = i < n && mask[ic] != 0;
            const double x = ((double)p1[ic].x - cx) * sc, y = ((double)p1[ic].y - cy) * sc;
            const double u = ((double)p2[ic].x - cx) * sc, v = ((double)p2[ic].y - cy) * sc;
            const double r0[8] = {x, y, 1, 0, 0, 0, -x * u, -y * u};
            const double r1[8] = {0, 0, 0, x, y, 1, -x * v, -y * v};
            add_full_terms<W>(acc[c], r0, r1, u, v, on, std::make_integer_sequence<int, NS>{});
        }
    }
    store_totals<W, 8>(acc, 36, A, b, std::make_integer_sequence<int, NS>{});
}

// Similarity: 10 upper-triangle entries of the 4 x 4 normal matrix (ids 0 .. 9) + 4 right-hand sides (10 .. 13).
template <int W>
__device__ __forceinline__ void refit_sums_partial(const float2* __restrict__ p1, const float2* __restrict__ p2, int n, const uint8_t* mask,
                                                   const double* par, double* A, double* b)
{
    const double cx = par[0], cy = par[1], sc = par[2];
    constexpr int NS = (14 - W + NW - 1) / NW;       // ids W, W + NW, ... < 14  (4 waves: 4, 4, 3, 3; 8 waves: 2 x 6, 1 x 2)
    const int lane = tid_now() & 63;
    double acc[4][NS];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int k = 0; k < NS; k++) acc[c][k] = 0.0;
    const int iters = (n + NT - 1) / NT;             // as refit_sums_full: branch-free, four slots side by side
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
            const int i = lane + 64 * c + NT * it;
            const int ic = min(i, n - 1);
            const bool on = i < n && mask[ic] != 0;
            const double x = ((double)p1[ic].x - cx) * sc, y = ((double)p1[ic].y - cy) * sc;
            const double u = ((double)p2[ic].x - cx) * sc, v = ((double)p2[ic].y - cy) * sc;
            const double r0[4] = {x, -y, 1, 0};
            const double r1[4] = {y, x, 0, 1};
            add_partial_terms<W>(acc[c], r0, r1, u, v, on, std::make_integer_sequence<int, NS>{});
        }
    }
    store_totals<W, 4>(acc, 10, A, b, std::make_integer_sequence<int, NS>{});
}

__device__ __forceinline__ bool refit(bool full, const float2* __restrict__ p1, const float2* __restrict__ p2, int n, const uint8_t* mask,
                      const double* par, double* A, double* b, double* H)
{
    const int lane = tid_now();
#ifdef LVK_RANSAC_TIMING
    const long long r0 = wall_clock64(); long long r1 = 0, r2 = 0;
#endif
    if (full)
    {
        static_assert(NT == 256, "four partials per lane");
        switch (tid_now() >> 6)
        {
            case 0: refit_sums_full<0>(p1, p2, n, mask, par, A, b); break;
            case 1: refit_sums_full<1>(p1, p2, n, mask, par, A, b); break;
            case 2: refit_sums_full<2>(p1, p2, n, mask, par, A, b); break;
            case 3: refit_sums_full<3>(p1, p2, n, mask, par, A, b); break;
            case 4: if constexpr (NW > 4) refit_sums_full<4 % NW>(p1, p2, n, mask, par, A, b); break;
            case 5: if constexpr (NW > 4) refit_sums_full<5 % NW>(p1, p2, n, mask, par, A, b); break;
            case 6: if constexpr (NW > 4) refit_sums_full<6 % NW>(p1, p2, n, mask, par, A, b); break;
            default: if constexpr (NW > 4) refit_sums_full<7 % NW>(p1, p2, n, mask, par, A, b); break;
        }
        __syncthreads();
#ifdef LVK_RANSAC_TIMING
        r1 = wall_clock64();
#endif
        __shared__ int s_ok;
        __shared__ int s_solved8;
        bool ok = solve_n<8>(A, b, &s_solved8);
#ifdef LVK_RANSAC_TIMING
        r2 = wall_clock64();
#endif
        // H = T^-1 Hn T, normalised by its last entry: entry q by lane q of wave 0 (every entry is its own expression of Hn, T, T^-1; the
        // nine of them one after another in a single lane were 0.5 us of every refit round)
        if (lane < 64)
        {
            if (ok)
            {
                const double cx = par[0], cy = par[1], sc = par[2];
                const double Hn[9] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1.0};
                // T = [sc 0 -cx sc; 0 sc -cy sc; 0 0 1], T^-1 = [1/sc 0 cx; 0 1/sc cy; 0 0 1]: column c of T and row r of T^-1 by selects (same
                // operand values as the indexed arrays of the sequential form, zeros included)
                const double isc = 1.0 / sc, tx = -cx * sc, ty = -cy * sc;
                auto entry = [&](int r, int c) -> double {
                    const double t0 = c == 0 ? sc : (c == 2 ? tx : 0.0), t1 = c == 1 ? sc : (c == 2 ? ty : 0.0), t2 = c == 2 ? 1.0 : 0.0;
                    const double i0 = r == 0 ? isc : 0.0, i1 = r == 1 ? isc : 0.0, i2 = r == 0 ? cx : (r == 1 ? cy : 1.0);
                    double Mc[3];                                                              // column c of M = Hn T
#pragma unroll
                    for (int k = 0; k < 3; k++) Mc[k] = (Hn[k * 3] * t0 + Hn[k * 3 + 1] * t1) + Hn[k * 3 + 2] * t2;
                    return (i0 * Mc[0] + i1 * Mc[1]) + i2 * Mc[2];
                };
                const int q = lane < 9 ? lane : 8;
                const double Rq = entry(q / 3, q % 3), R8 = entry(2, 2);
                if (fabs(R8) < 1e-12) ok = false;
                else if (lane < 9) H[lane] = Rq / R8;
            }
            if (lane == 0) s_ok = ok ? 1 : 0;
        }
        __syncthreads();
#ifdef LVK_RANSAC_TIMING
        if (lane == 0) printf("  refit: sums %lld, solve %lld, normalise %lld\n", r1 - r0, r2 - r1, wall_clock64() - r2);
#endif
        return s_ok != 0;
    }
    else
    {
        switch (tid_now() >> 6)
        {
            case 0: refit_sums_partial<0>(p1, p2, n, mask, par, A, b); break;
            case 1: refit_sums_partial<1>(p1, p2, n, mask, par, A, b); break;
            case 2: refit_sums_partial<2>(p1, p2, n, mask, par, A, b); break;
            case 3: refit_sums_partial<3>(p1, p2, n, mask, par, A, b); break;
            case 4: if constexpr (NW > 4) refit_sums_partial<4 % NW>(p1, p2, n, mask, par, A, b); break;
            case 5: if constexpr (NW > 4) refit_sums_partial<5 % NW>(p1, p2, n, mask, par, A, b); break;
            case 6: if constexpr (NW > 4) refit_sums_partial<6 % NW>(p1, p2, n, mask, par, A, b); break;
            default: if constexpr (NW > 4) refit_sums_partial<7 % NW>(p1, p2, n, mask, par, A, b); break;
        }
        __syncthreads();
        __shared__ int s_solved4, s_ok2;
        const bool ok = solve_n<4>(A, b, &s_solved4);
        if (lane == 0)
        {
            if (ok)
            {
                const double cx = par[0], cy = par[1], sc = par[2];
                const double a = b[0], bb = b[1], tx = b[2], ty = b[3];
                H[0] = a; H[1] = -bb; H[2] = (tx / sc + cx) - (a * cx - bb * cy);
                H[3] = bb; H[4] = a;  H[5] = (ty / sc + cy) - (bb * cx + a * cy);
                // (H[6..8] = 0, 0, 1: constant for this model, written once by the caller)
            }
            s_ok2 = ok ? 1 : 0;
        }
        __syncthreads();
        return s_ok2 != 0;
    }
}

// Every thread's stores so far are visible to the host, THEN the word changes (see LvkHostSignal).  Block-uniform call sites only.
__device__ __forceinline__ void signal_host(LvkHostSignal done)
{
    if (!done.flag) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One block: pick the best hypothesis, run the local optimisation, emit H (9 doubles), #inliers and the mask.
// The block has to fit NEXT TO the co-scheduled remap of the overlap mode (4 waves x 80 VGPRs per SIMD leave 192 of a SIMD's 512): as 4
// waves (one per SIMD) it may take 168 VGPRs, as 8 waves (two per SIMD) 96 each; the unconstrained allocation of 250 made it wait for remap
// workgroups to retire.
template <bool STAGED>
__global__ __launch_bounds__(FT) __attribute__((amdgpu_waves_per_eu(FT == 512 ? 5 : 3)))
void k_ransac_finalize(const float2* __restrict__ g1, const float2* __restrict__ g2, int n, const int* __restrict__ n_dev, double t2, int full,
                       const int* __restrict__ full_dev, double cx, double cy, double sc,
                       const double* __restrict__ hyp_H, const long long* __restrict__ hyp_score,
                       uint8_t* __restrict__ gmask_a, uint8_t* __restrict__ gmask_b,
                       double* __restrict__ out_H, int* __restrict__ out_ninl, uint8_t* __restrict__ out_mask, LvkHostSignal done)
{
    LVK_TL(1);
    LVK_TRACKER_PRIORITY();
    __shared__ double sA[64], sb[8], sH[9], sBest[9];
    __shared__ double s_par[4];                           // cx, cy, sc, t2: read back where they are used (see refit_sums_full)
    __shared__ long long s_scratch[FT / 64];
    __shared__ long long s_best[FT / 64]; __shared__ int s_best_h[FT / 64];
    __shared__ float2 s_p1[STAGED ? LDS_POINTS : 1], s_p2[STAGED ? LDS_POINTS : 1];
    __shared__ uint8_t s_mask[2][STAGED ? LDS_POINTS : 1];
    const int lane = threadIdx.x;
    if (n_dev) n = max(min(*n_dev, n), 0);
    if (full_dev) full = *full_dev;
    if (lane == 0) { s_par[0] = cx; s_par[1] = cy; s_par[2] = sc; s_par[3] = t2; }
    if (STAGED)
        for (int i = lane; i < n; i += FT) { s_p1[i] = g1[i]; s_p2[i] = g2[i]; }
    __syncthreads();
    const float2* p1 = STAGED ? s_p1 : g1;
    const float2* p2 = STAGED ? s_p2 : g2;
    uint8_t* mask_a = STAGED ? s_mask[0] : gmask_a;
    uint8_t* mask_b = STAGED ? s_mask[1] : gmask_b;
    const int m = full ? 4 : 2;
#ifdef LVK_RANSAC_TIMING
    long long tt[16]; int nt_ = 0;
#define RT_MARK() do { if (nt_ < 16) tt[nt_++] = wall_clock64(); } while (0)
#else
#define RT_MARK() do { } while (0)
#endif
    RT_MARK();
    // argmax over the hypotheses: highest score, lowest index on ties
    long long best = -1; int best_h = -1;
    for (int h = lane; h < K_HYPOTHESES; h += FT)
    {
        const long long s = hyp_score[h];
        if (s > best) { best = s; best_h = h; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        const long long os = __shfl_xor(best, o); const int oh = __shfl_xor(best_h, o);
        if (os > best || (os == best && oh >= 0 && (best_h < 0 || oh < best_h))) { best = os; best_h = oh; }
    }
    if ((lane & 63) == 0) { s_best[lane >> 6] = best; s_best_h[lane >> 6] = best_h; }
    __syncthreads();
    best = s_best[0]; best_h = s_best_h[0];
    for (int w = 1; w < FT / 64; w++)
    {
        const long long os = s_best[w]; const int oh = s_best_h[w];
        if (os > best || (os == best && oh >= 0 && (best_h < 0 || oh < best_h))) { best = os; best_h = oh; }
    }
    if (best_h < 0 || best < 0)
    {
        for (int i = lane; i < n; i += FT) out_mask[i] = 0;
        if (lane == 0) { for (int q = 0; q < 9; q++) out_H[q] = (q % 4 == 0) ? 1.0 : 0.0; *out_ninl = -2; }
        signal_host(done);
        return;
    }
    if (lane < 9) sBest[lane] = hyp_H[best_h * 9 + lane];
    __syncthreads();
    uint8_t* cur = mask_a; uint8_t* trial = mask_b;
    int ninl = 0;
    if (!full && lane == 0) { sH[6] = 0.0; sH[7] = 0.0; sH[8] = 1.0; }      // the last row of a similarity (refit() writes sH[0..5]); ordered by the barriers below
    RT_MARK();
    long long best_score = score_model<FT>(sBest, p1, p2, n, s_par[3], cur, &ninl, s_scratch);
    __syncthreads();
    RT_MARK();
    for (int round = 0; round < LO_ROUNDS; round++)
    {
        if (ninl < m) break;
        if (!refit(full != 0, p1, p2, n, cur, s_par, sA, sb, sH)) break;
        RT_MARK();
        int nt = 0;
        const long long s = score_model<FT>(sH, p1, p2, n, s_par[3], trial, &nt, s_scratch);
        __syncthreads();
        RT_MARK();
        if (s <= best_score) break;
        best_score = s; ninl = nt;
        if (lane < 9) sBest[lane] = sH[lane];
        uint8_t* t = cur; cur = trial; trial = t;
        __syncthreads();
    }
    for (int i = lane; i < n; i += FT) out_mask[i] = cur[i];
    if (lane < 9) out_H[lane] = sBest[lane];
    if (lane == 0) *out_ninl = ninl;
    signal_host(done);
#ifdef LVK_RANSAC_TIMING
    RT_MARK();
    if (lane == 0)
    {
        printf("finalize n %d ninl %d (100 MHz ticks): stage+argmax, score, then (refit, score) per round, output:", n, ninl);
        for (int k = 1; k < nt_; k++) printf(" %lld", tt[k] - tt[k - 1]);
        printf("\n");
    }
#endif
}

// fast_filter as a kernel of its own (the field preset, lens pairs beyond the fused variant's capacity).
// Block size: 256 threads for up to 1024 pairs (the usual 600-800), 1024 beyond.  A 1024-thread block needs 4 free wave slots on every
// SIMD of one CU at the same moment; next to the persistent remap grid (4 waves per SIMD) and the flow kernel's last blocks it is placed
// several microseconds late.
template <int CMP_NT>
__global__ __launch_bounds__(CMP_NT)
void k_match_compact(const float2* __restrict__ prev, const float2* __restrict__ matched, const uint8_t* __restrict__ status, int n,
                     float2* __restrict__ p1, float2* __restrict__ p2, int* __restrict__ count, int* __restrict__ host_count,
                     float2* __restrict__ host_matched, uint8_t* __restrict__ host_status,
                     const float2* __restrict__ und, float region_w, float region_h, const int* __restrict__ n_raw_dev)
{
    LVK_TL(2);
    LVK_TRACKER_PRIORITY();
    if (n_raw_dev) n = min(max(*n_raw_dev, 0), n);          // the point count was decided on the device: n is the launch's upper bound
    __shared__ unsigned short s_above[CMP_CAP];            // number of dropped elements with a higher index
    __shared__ uint8_t s_keep[CMP_CAP];
    __shared__ int s_wave[CMP_NT / 64];
    const int t = threadIdx.x;
    // fused lens mode: `und` holds the lens-corrected positions (previous | matched); they are what the motion is estimated from
    const float2* pair_prev = und ? und : prev;
    const float2* pair_next = und ? und + n : matched;
    const int m = compact_plan<CMP_NT>(matched, status, n, und, region_w, region_h, s_above, s_keep, s_wave, true, host_matched, host_status);
    if (t == 0) { *count = m; *host_count = m; }
    for (int i = t; i < m; i += CMP_NT)
    {
        const int src = compact_source(i, n, s_above, s_keep);
        p1[i] = pair_prev[src]; p2[i] = pair_next[src];
    }
}

} // namespace

size_t lvk_ransac_workspace_bytes(int n)
{
    // hypotheses (H + score) | mask_a | mask_b
    return (size_t)K_HYPOTHESES * (9 * sizeof(double) + sizeof(long long)) + 2 * (((size_t)n + 255) & ~(size_t)255);
}

// d_p1/d_p2: n pairs; d_ws: lvk_ransac_workspace_bytes(n); outputs d_H (9 doubles), d_ninl, d_mask (n bytes).
int lvk_launch_ransac(lvk_hip_ctx* ctx, const float2* d_p1, const float2* d_p2, int n, double threshold, double region_w, double region_h,
                      bool full_homography, void* d_ws, double* d_H, int* d_ninl, uint8_t* d_mask, const int* d_n, const int* d_full, LvkHostSignal done)
{
    CompactArgs ca{}; ca.full_dev = d_full;
    LVK_HIP_REQUIRE(ctx, d_p1 && d_p2 && d_ws && d_H && d_ninl && d_mask && (d_n || n >= (full_homography ? 4 : 2)));
    double* hyp_H = (double*)d_ws;
    long long* hyp_score = (long long*)(hyp_H + K_HYPOTHESES * 9);
    uint8_t* mask_a = (uint8_t*)(hyp_score + K_HYPOTHESES);
    uint8_t* mask_b = mask_a + (((size_t)n + 255) & ~(size_t)255);
    const double t2 = threshold * threshold;
    if (n <= LDS_POINTS)
    {
        hipLaunchKernelGGL(k_ransac_hypotheses<true>, dim3(K_HYPOTHESES), dim3(NT), 0, ctx->stream, d_p1, d_p2, n, d_n, t2, full_homography ? 1 : 0, hyp_H, hyp_score, ca);
        hipLaunchKernelGGL(k_ransac_finalize<true>, dim3(1), dim3(FT), 0, ctx->stream, d_p1, d_p2, n, d_n, t2, full_homography ? 1 : 0, d_full,
                           region_w * 0.5, region_h * 0.5, 2.0 / (region_w + region_h), hyp_H, hyp_score, mask_a, mask_b, d_H, d_ninl, d_mask, done);
    }
    else
    {
        hipLaunchKernelGGL(k_ransac_hypotheses<false>, dim3(K_HYPOTHESES), dim3(NT), 0, ctx->stream, d_p1, d_p2, n, d_n, t2, full_homography ? 1 : 0, hyp_H, hyp_score, ca);
        hipLaunchKernelGGL(k_ransac_finalize<false>, dim3(1), dim3(FT), 0, ctx->stream, d_p1, d_p2, n, d_n, t2, full_homography ? 1 : 0, d_full,
                           region_w * 0.5, region_h * 0.5, 2.0 / (region_w + region_h), hyp_H, hyp_score, mask_a, mask_b, d_H, d_ninl, d_mask, done);
    }
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

int lvk_launch_match_compact(lvk_hip_ctx* ctx, const float2* d_prev, const float2* d_matched, const uint8_t* d_status, int n,
                             float2* d_p1, float2* d_p2, int* d_count, int* h_count, float2* h_matched, uint8_t* h_status,
                             const float2* d_und, float region_w, float region_h, const int* d_n_raw)
{
    LVK_HIP_REQUIRE(ctx, d_prev && d_matched && d_status && d_p1 && d_p2 && d_count && h_count && h_matched && h_status && n >= 0 && n <= CMP_CAP);
    if (n <= 1024)
        hipLaunchKernelGGL(k_match_compact<256>, dim3(1), dim3(256), 0, ctx->stream, d_prev, d_matched, d_status, n, d_p1, d_p2, d_count, h_count, h_matched, h_status,
                           d_und, region_w, region_h, d_n_raw);
    else
        hipLaunchKernelGGL(k_match_compact<1024>, dim3(1), dim3(1024), 0, ctx->stream, d_prev, d_matched, d_status, n, d_p1, d_p2, d_count, h_count, h_matched, h_status,
                           d_und, region_w, region_h, d_n_raw);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

// fast_filter + RANSAC in two kernels instead of three: the hypotheses kernel compacts the flow result itself (see k_ransac_hypotheses,
// FUSED).  Same arguments and results as lvk_launch_match_compact followed by lvk_launch_ransac(.., d_count); n <= LVK_COMPACT_RANSAC_MAX pairs.
// d_n_raw / d_full: the number of tracked points and the model choice live on the device (n = upper bound, full_homography ignored).
int lvk_launch_compact_ransac(lvk_hip_ctx* ctx, const float2* d_prev, const float2* d_matched, const uint8_t* d_status, int n,
                              float2* d_p1, float2* d_p2, int* d_count, int* h_count, float2* h_matched, uint8_t* h_status,
                              const float2* d_und, float region_wf, float region_hf,
                              double threshold, double region_w, double region_h, bool full_homography, void* d_ws, double* d_H, int* d_ninl, uint8_t* d_mask,
                              const int* d_n_raw, const int* d_full, LvkHostSignal done)
{
    static_assert(LVK_COMPACT_RANSAC_MAX % NT == 0 && LVK_COMPACT_RANSAC_MAX <= LDS_POINTS, "the fused variant compacts into its LDS copy");
    LVK_HIP_REQUIRE(ctx, d_prev && d_matched && d_status && d_p1 && d_p2 && d_count && h_count && h_matched && h_status && n > 0 && n <= LVK_COMPACT_RANSAC_MAX);
    LVK_HIP_REQUIRE(ctx, d_ws && d_H && d_ninl && d_mask);
    double* hyp_H = (double*)d_ws;
    long long* hyp_score = (long long*)(hyp_H + K_HYPOTHESES * 9);
    uint8_t* mask_a = (uint8_t*)(hyp_score + K_HYPOTHESES);
    uint8_t* mask_b = mask_a + (((size_t)n + 255) & ~(size_t)255);
    const double t2 = threshold * threshold;
    const CompactArgs ca{d_prev, d_matched, d_status, d_und, region_wf, region_hf, d_p1, d_p2, d_count, h_count, h_matched, h_status, d_n_raw, d_full};
    hipLaunchKernelGGL((k_ransac_hypotheses<true, true>), dim3(K_HYPOTHESES), dim3(NT), 0, ctx->stream, (const float2*)nullptr, (const float2*)nullptr, n, (const int*)nullptr,
                       t2, full_homography ? 1 : 0, hyp_H, hyp_score, ca);
    hipLaunchKernelGGL(k_ransac_finalize<true>, dim3(1), dim3(FT), 0, ctx->stream, (const float2*)d_p1, (const float2*)d_p2, n, (const int*)d_count, t2, full_homography ? 1 : 0, d_full,
                       region_w * 0.5, region_h * 0.5, 2.0 / (region_w + region_h), hyp_H, hyp_score, mask_a, mask_b, d_H, d_ninl, d_mask, done);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

extern "C" {

// Synchronous test entry point: host point arrays (n x 2 floats), outputs H (9 doubles, row major), mask (n bytes).
// Returns the inlier count (>= 0) or a negative status: LVK_HIP_ERR_* or -10 - (reason) when no model was found.
int lvk_hip_estimate_global_motion(lvk_hip_ctx* ctx, const float* pts1, const float* pts2, int n, double threshold,
                                   double region_w, double region_h, int full_homography, double H[9], uint8_t* mask)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, pts1 && pts2 && H && mask && n >= 0);
    for (int q = 0; q < 9; q++) H[q] = (q % 4 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < n; i++) mask[i] = 0;
    if (n < (full_homography ? 4 : 2)) return -11;
    float2 *d_p1 = nullptr, *d_p2 = nullptr; void* d_ws = nullptr; double* d_H = nullptr; int* d_n = nullptr; uint8_t* d_mask = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_p1); (void)hipFree(d_p2); (void)hipFree(d_ws); (void)hipFree(d_H); (void)hipFree(d_n); (void)hipFree(d_mask); };
    hipError_t e;
    if ((e = hipMalloc((void**)&d_p1, n * sizeof(float2))) != hipSuccess || (e = hipMalloc((void**)&d_p2, n * sizeof(float2))) != hipSuccess ||
        (e = hipMalloc(&d_ws, lvk_ransac_workspace_bytes(n))) != hipSuccess || (e = hipMalloc((void**)&d_H, 9 * sizeof(double))) != hipSuccess ||
        (e = hipMalloc((void**)&d_n, sizeof(int))) != hipSuccess || (e = hipMalloc((void**)&d_mask, n)) != hipSuccess ||
        (e = hipMemcpyAsync(d_p1, pts1, n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(d_p2, pts2, n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
    { cleanup(); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
    int rc = lvk_launch_ransac(ctx, d_p1, d_p2, n, threshold, region_w, region_h, full_homography != 0, d_ws, d_H, d_n, d_mask);
    int ninl = 0;
    if (rc == LVK_HIP_OK)
    {
        if ((e = hipMemcpyAsync(H, d_H, 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
            (e = hipMemcpyAsync(&ninl, d_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
            (e = hipMemcpyAsync(mask, d_mask, n, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(ctx->stream)) != hipSuccess)
        { cleanup(); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
    }
    cleanup();
    if (rc != LVK_HIP_OK) return rc;
    return ninl >= 0 ? ninl : -10 + ninl;
}

// Synchronous test entry point of the GPU-side fast_filter: host arrays in, compacted pairs + count out.
int lvk_hip_fast_filter(lvk_hip_ctx* ctx, const float* prev, const float* matched, const uint8_t* status, int n, float* out_prev, float* out_matched)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, prev && matched && status && out_prev && out_matched && n >= 0 && n <= CMP_CAP);
    if (n == 0) return 0;
    float2 *d = nullptr, *h_m = nullptr; uint8_t *d_s = nullptr, *h_s = nullptr; int *d_c = nullptr, *h_c = nullptr;
    auto cleanup = [&]() { (void)hipFree(d); (void)hipFree(d_s); (void)hipFree(d_c); (void)hipHostFree(h_m); (void)hipHostFree(h_s); (void)hipHostFree(h_c); };
    hipError_t e;
    if ((e = hipMalloc((void**)&d, 4 * (size_t)n * sizeof(float2))) != hipSuccess || (e = hipMalloc((void**)&d_s, n)) != hipSuccess ||
        (e = hipMalloc((void**)&d_c, sizeof(int))) != hipSuccess || (e = hipHostMalloc((void**)&h_m, n * sizeof(float2), hipHostMallocDefault)) != hipSuccess ||
        (e = hipHostMalloc((void**)&h_s, n, hipHostMallocDefault)) != hipSuccess || (e = hipHostMalloc((void**)&h_c, sizeof(int), hipHostMallocDefault)) != hipSuccess ||
        (e = hipMemcpyAsync(d, prev, n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(d + n, matched, n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(d_s, status, n, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
    { cleanup(); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
    int rc = lvk_launch_match_compact(ctx, d, d + n, d_s, n, d + 2 * n, d + 3 * n, d_c, h_c, h_m, h_s, nullptr, 0.0f, 0.0f);
    int m = 0;
    if (rc == LVK_HIP_OK)
    {
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) { cleanup(); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
        m = *h_c;
        if ((e = hipMemcpy(out_prev, d + 2 * n, (size_t)m * sizeof(float2), hipMemcpyDeviceToHost)) != hipSuccess ||
            (e = hipMemcpy(out_matched, d + 3 * n, (size_t)m * sizeof(float2), hipMemcpyDeviceToHost)) != hipSuccess)
        { cleanup(); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
        for (int i = 0; i < n; i++) if (h_s[i] != status[i] || h_m[i].x != matched[2 * i] || h_m[i].y != matched[2 * i + 1]) rc = LVK_HIP_ERR_RUNTIME;
        if (rc != LVK_HIP_OK) ctx->fail(rc, "host mirror of the flow result differs");
    }
    cleanup();
    return rc == LVK_HIP_OK ? m : rc;
}

} // extern "C"

LVK_TL_EXPORT(motion)
