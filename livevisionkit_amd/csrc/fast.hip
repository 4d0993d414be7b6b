// FAST-9/16 corner detection with 3x3 non-max suppression for gfx950, run per detection region.
//
// Replaces cv::FastFeatureDetector(threshold, nonmax=true, TYPE_9_16)->detect(frame(region.bounds)) as called by
// FeatureDetector::detect (reference: LiveVisionKit/Vision/FeatureDetector.cpp:38-41,130-134; arithmetic: OpenCV 4.8.0
// features2d/fast.cpp FAST_t<16> + cornerScore<16>).  Integer only -> bit-exact.
//
// Structure: k_fast_detect stages a (64+8) x (8+8) image tile (3-px ring apron + 1-px suppression apron) in LDS,
// scores the tile plus its 1-px halo, suppresses non-maxima and emits one wave ballot per 64-pixel row segment.
// k_fast_compact turns the ballots into the row-major keypoint list the CPU path would produce (the order is
// load-bearing: it fixes feature indices downstream).
#include "lvk_hip_internal.hpp"

namespace {

constexpr int TW = 64, TH = 8;              // output tile
constexpr int AP = 4;                       // apron: 3 (ring radius) + 1 (NMS neighbourhood)
constexpr int IW = TW + 2 * AP, IH = TH + 2 * AP;
constexpr int SW = TW + 2, SH = TH + 2;     // score tile incl. 1-px halo

__device__ __forceinline__ int fast_score(const uint8_t* c, int stride, int threshold)
{
    // ring offsets (dx, dy), radius 3, starting at (0, 3) -- fast.cpp makeOffsets
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * stride];          d[1] = v - c[1 + 3 * stride];   d[2] = v - c[2 + 2 * stride];   d[3] = v - c[3 + stride];
    d[4] = v - c[3];                   d[5] = v - c[3 - stride];       d[6] = v - c[2 - 2 * stride];   d[7] = v - c[1 - 3 * stride];
    d[8] = v - c[-3 * stride];         d[9] = v - c[-1 - 3 * stride];  d[10] = v - c[-2 - 2 * stride]; d[11] = v - c[-3 - stride];
    d[12] = v - c[-3];                 d[13] = v - c[-3 + stride];     d[14] = v - c[-2 + 2 * stride]; d[15] = v - c[-1 + 3 * stride];
    // best 9-arc: max over arcs of min(d) (darker ring) and of min(-d) (brighter ring), by doubling
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { lo2[k] = min(d[k], d[(k + 1) & 15]); hi2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { lo4[k] = min(lo2[k], lo2[(k + 2) & 15]); hi4[k] = max(hi2[k], hi2[(k + 2) & 15]); }
    int dark = -256, bright_neg = 256;
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
        const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
        dark = max(dark, lo9);
        bright_neg = min(bright_neg, hi9);
    }
    const int m = max(dark, -bright_neg);
    return m > threshold ? m - 1 : 0;       // cornerScore: (largest threshold at which it is still a corner)
}

// The region descriptors travel BY VALUE (kernel arguments) when there are few of them, instead of every workgroup reading them from pinned
// host memory across the link.
struct FastRegionList { FastRegion r[LVK_FAST_INLINE_REGIONS]; };

// The suppression grid's per-cell reduction, done by the detector itself when `cells.first` is given (see k_fast_insert below): every kept
// corner folds its position key -- (region, row, column) packed in the order FeatureDetector::detect meets the corners in -- into its
// cell's "first corner" (atomicMin) and "strongest corner, earliest on ties" (atomicMax of score << 24 | ~key) slots in global memory.
// best: score << 56 | ~key << 32 | x | y << 12 | score << 24 -- the order is decided by the upper half (score, then the EARLIER key), the lower
// half is the winner's feature record riding along
struct FastCells { const uint16_t* col_of; const uint32_t* row_base; uint32_t* first; unsigned long long* best; int* region_count; };

__global__ __launch_bounds__(TW * TH)
void k_fast_detect(const uint8_t* __restrict__ img, int step, int rows, int cols,
                   const FastRegion* __restrict__ regions, FastRegionList inl, int n_inline, int segs_x,
                   unsigned long long* __restrict__ masks, uint8_t* __restrict__ scores, int max_rh, int max_rw, FastCells cells)
{
    LVK_TRACKER_PRIORITY();
    const FastRegion rg = n_inline ? inl.r[blockIdx.z] : regions[blockIdx.z];
    if (!rg.active) return;
    const int lx0 = blockIdx.x * TW, ly0 = blockIdx.y * TH;
    if (lx0 >= rg.w || ly0 >= rg.h) return;

    __shared__ uint8_t s_img[IH][IW];
    __shared__ uint8_t s_score[SH][SW];
    const int tid = threadIdx.y * TW + threadIdx.x;

    // stage the image tile (addresses clamped to the frame; out-of-region pixels are never used for a valid score)
    for (int i = tid; i < IW * IH; i += TW * TH)
    {
        const int ty = i / IW, tx = i - ty * IW;
        const int gx = min(max(rg.x + lx0 - AP + tx, 0), cols - 1);
        const int gy = min(max(rg.y + ly0 - AP + ty, 0), rows - 1);
        s_img[ty][tx] = img[(long)gy * step + gx];
    }
    // (suppression grid on the device: the cell coordinates of this tile's 64 columns and 8 rows, fetched with the tile -- no round trip of
    //  their own later, when a kept corner looks its cell up)
    __shared__ uint32_t s_cell_col[TW], s_cell_row[TH];
    if (cells.first)
    {
        if (tid < TW) s_cell_col[tid] = cells.col_of[min(rg.x + lx0 + tid, cols - 1)];
        else if (tid < TW + TH) s_cell_row[tid - TW] = cells.row_base[min(rg.y + ly0 + tid - TW, rows - 1)];
    }
    __syncthreads();

    // score the tile and its 1-px halo; positions whose ring leaves the region score 0 (the ROI edge is the image edge)
    for (int i = tid; i < SW * SH; i += TW * TH)
    {
        const int ty = i / SW, tx = i - ty * SW;
        const int lx = lx0 - 1 + tx, ly = ly0 - 1 + ty;
        int s = 0;
        if (lx >= 3 && lx < rg.w - 3 && ly >= 3 && ly < rg.h - 3)
            s = fast_score(&s_img[ty + AP - 1][tx + AP - 1], IW, rg.threshold);
        s_score[ty][tx] = (uint8_t)s;
    }
    __syncthreads();

    const int lx = lx0 + threadIdx.x, ly = ly0 + threadIdx.y;
    const int sx = threadIdx.x + 1, sy = threadIdx.y + 1;
    const int s = s_score[sy][sx];
    const bool keep = s > 0 &&
        s > s_score[sy][sx - 1] && s > s_score[sy][sx + 1] &&
        s > s_score[sy - 1][sx - 1] && s > s_score[sy - 1][sx] && s > s_score[sy - 1][sx + 1] &&
        s > s_score[sy + 1][sx - 1] && s > s_score[sy + 1][sx] && s > s_score[sy + 1][sx + 1];
    const unsigned long long mask = __ballot(keep);
    if (ly < rg.h)
    {
        if (threadIdx.x == 0) masks[((long)blockIdx.z * max_rh + ly) * segs_x + blockIdx.x] = mask;
        if (threadIdx.x == 0 && cells.first && mask) atomicAdd(&cells.region_count[blockIdx.z], __popcll(mask));      // raw corners of the region
        if (keep) scores[((long)blockIdx.z * max_rh + ly) * max_rw + lx] = (uint8_t)s;
        if (keep && cells.first)
        {
            const uint32_t ci = s_cell_row[threadIdx.y] + s_cell_col[threadIdx.x];
            const uint32_t key = (uint32_t)(((int)blockIdx.z * max_rh + ly) * max_rw + lx);
            atomicMin(&cells.first[ci], key);
            const uint32_t rec = (uint32_t)(rg.x + lx) | ((uint32_t)(rg.y + ly) << 12) | ((uint32_t)s << 24);
            atomicMax(&cells.best[ci], ((unsigned long long)(((uint32_t)s << 24) | (0xFFFFFFu - key)) << 32) | rec);
        }
    }
}

// One block per region: exclusive scan of the per-segment popcounts in row-major order, then ordered scatter.
__global__ __launch_bounds__(1024)
void k_fast_compact(const FastRegion* __restrict__ regions, FastRegionList inl, int n_inline, int segs_x,
                    const unsigned long long* __restrict__ masks, const uint8_t* __restrict__ scores, int max_rh, int max_rw,
                    uint32_t* __restrict__ out, int cap, int* __restrict__ counts)
{
    LVK_TRACKER_PRIORITY();
    const int r = blockIdx.x;
    const FastRegion rg = n_inline ? inl.r[r] : regions[r];
    if (!rg.active) { if (threadIdx.x == 0) counts[r] = 0; return; }
    const int rsegs = (rg.w + TW - 1) / TW;
    const int nseg = rg.h * rsegs;
    __shared__ int s_wave[16];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int chunk = 0; chunk < nseg; chunk += 1024)
    {
        const int seg = chunk + threadIdx.x;
        int ly = 0, sg = 0;
        unsigned long long m = 0;
        if (seg < nseg)
        {
            ly = seg / rsegs; sg = seg - ly * rsegs;
            m = masks[((long)r * max_rh + ly) * segs_x + sg];
        }
        const int cnt = __popcll(m);
        // inclusive scan inside the wave
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int w = 0; w < wave; w++) wave_off += s_wave[w];
        int pos = s_base + wave_off + incl - cnt;
        while (m)
        {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int lx = sg * TW + bit;
            if (pos < cap)
                out[(long)r * cap + pos] = (uint32_t)lx | ((uint32_t)ly << 12) | ((uint32_t)scores[((long)r * max_rh + ly) * max_rw + lx] << 24);
            pos++;
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_base = s_base + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[r] = s_base;
}


// ---- the suppression grid on the device --------------------------------------------------------------------------------------------
// FeatureDetector::detect runs every raw corner, region after region in row-major order, through a grid with one slot per cell
// (Vision/FeatureDetector.cpp:138-157): an empty cell takes the corner (appended to the feature list), a cell that holds a DETECTED
// feature keeps the stronger of the two in place (strictly greater response: the earlier corner wins a tie), a cell that holds a
// PROPAGATED feature never changes.  Closed form of what that loop leaves behind: the new features are, for every cell without a
// propagated feature that receives at least one corner, the corner of maximal score (earliest on ties), listed in the order of the
// cells' FIRST corners.  "Earlier" is the order of (region, row, column) -- a position key, no sequence numbers needed:
//   * k_fast_detect folds every kept corner into its cell's two slots (above): the per-corner work runs on the detector's own ~540
//     workgroups, one corner per lane, no dependent memory round trips in a serial loop (a first version that walked the ballots in ONE
//     workgroup spent 35-90 us on exactly those, next to a remap that keeps every memory pipe busy);
//   * k_fast_insert (one workgroup) ranks the cells by their first corner -- a bitmap over the position keys in LDS and a prefix popcount --,
//     appends the winners to the list the optical-flow kernel reads, evaluates SpatialMap::distribution_quality (Data/SpatialMap.tpp:589-625)
//     over propagated + new cells, the two early-outs of FrameTracker::track (Vision/FrameTracker.cpp:127-131) and the homography /
//     similarity choice (:170) for the kernels that follow, and clears the slots for the next frame.
struct FastInsertArgs
{
    const uint8_t* bucket;                // device table of the grid (FeatureGridH): distribution bucket of every cell
    uint32_t* first; unsigned long long* best;      // the per-cell slots k_fast_detect filled (device memory; left cleared)
    int* region_count;                    // raw corners per region, summed by k_fast_detect (device memory; left cleared)
    int capacity, small_grid, n_held, min_samples; float uniformity, homography_threshold;
    int nkeys;                            // position keys: nregions * max_rh * max_rw
    float2* pts;                          // the optical flow's point list: new points go to pts[n_held ...]
    uint32_t* new_kp;                     // x | y << 12 | score << 24 (frame coordinates) of the new features, list order
    int* result;                          // [0] new features, [1] points to track (0: early out), [2] quality (float bits), [3] full homography, [4] corners
    int* d_n; int* d_full;                // [1] and [3] again in device memory, for the kernels of the chain
    int* counts;                          // raw corner count per region
    uint32_t occ[128];                    // cells that hold a propagated feature, one bit each (by value: no read across the host link)
    int occ_bucket[16];                   // ... and how many of them lie in each distribution bucket
};

// 1024 threads x <= 48 VGPRs: four wavefronts per SIMD fit into the 192 VGPRs the persistent remap grid of the overlap mode leaves free, and
// the SIMDs have four of this kernel's (prioritised) waves to issue from -- next to the remap a lone wave issues an instruction every 5-10
// cycles, and the kernel is a few thousand instructions of bookkeeping, not arithmetic.
constexpr int INS_NT = 1024, INS_MAX_CELLS = 4096, INS_CPT = INS_MAX_CELLS / INS_NT, INS_MAX_KEYS = 1 << 18;

// LDS (dynamic): bitmap over the position keys (nkeys / 32 words) + exclusive prefix popcount per bitmap word (u16)
__host__ __device__ inline size_t ins_lds_bytes(int nkeys) { const size_t words = ((size_t)nkeys + 31) / 32; return words * 4 + ((words + 1) & ~(size_t)1) * 2; }

__global__ __launch_bounds__(INS_NT)
void k_fast_insert(int nregions, FastInsertArgs a)
{
    LVK_TRACKER_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) uint8_t ins_smem[];
    const int bwords = (a.nkeys + 31) / 32;
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(ins_smem);
    unsigned short* s_pre = reinterpret_cast<unsigned short*>(s_bits + bwords);
    __shared__ int s_wave[INS_NT / 64], s_bucket[16], s_count[LVK_FAST_INLINE_REGIONS];
    __shared__ uint32_t s_occ[INS_MAX_CELLS / 32];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // everything this thread needs from global memory, all loads in flight together: next to a remap that keeps the memory pipes busy a
    // dependent round trip costs microseconds, and this kernel has exactly one
    uint32_t f[INS_CPT]; unsigned long long bk[INS_CPT]; uint32_t bu[INS_CPT];
#pragma unroll
    for (int k = 0; k < INS_CPT; k++)
    {
        const int ci = t + k * INS_NT;
        f[k] = ci < a.capacity ? a.first[ci] : 0xFFFFFFFFu;
        bk[k] = ci < a.capacity ? a.best[ci] : 0ull;
        bu[k] = ci < a.capacity ? a.bucket[ci] : 0u;
    }
    const int my_count = t < nregions ? a.region_count[t] : 0;
#pragma unroll 1
    for (int i = t; i < bwords; i += INS_NT) s_bits[i] = 0u;
    if (t < INS_MAX_CELLS / 32) s_occ[t] = a.occ[t];
    if (t < LVK_FAST_INLINE_REGIONS) s_count[t] = my_count;
    if (t < nregions && my_count) a.region_count[t] = 0;
    if (t < 16) s_bucket[t] = a.occ_bucket[t];
    __syncthreads();

    // the cells that received a corner and hold no propagated feature: their first corner's key into the bitmap; slots cleared for the next
    // frame.  The distribution buckets of the PROPAGATED cells come counted from the host (it marked them); only the new cells are counted
    // here (a few hundred LDS atomics instead of the thousand that cost the first version 4 us)
#pragma unroll
    for (int k = 0; k < INS_CPT; k++)
    {
        const int ci = t + k * INS_NT;
        if (ci < a.capacity && f[k] != 0xFFFFFFFFu)
        {
            a.first[ci] = 0xFFFFFFFFu; a.best[ci] = 0ull;
            if ((s_occ[ci >> 5] >> (ci & 31)) & 1u) f[k] = 0xFFFFFFFFu;      // a propagated feature: whatever the detector found there is ignored
            else { atomicOr(&s_bits[f[k] >> 5], 1u << (f[k] & 31)); atomicAdd(&s_bucket[bu[k] & 15u], 1); }
        }
    }
    __syncthreads();
    // exclusive prefix popcount over the bitmap words: consecutive words per thread, one wave scan, one barrier
    const int wpt = (bwords + INS_NT - 1) / INS_NT, wb0 = t * wpt, wb1 = min(wb0 + wpt, bwords);
    int mine = 0;
#pragma unroll 1
    for (int w = wb0; w < wb1; w++) mine += __popc(s_bits[w]);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, n_new = 0;
#pragma unroll 2
    for (int q = 0; q < INS_NT / 64; q++) { const int u = s_wave[q]; n_new += u; if (q < wave) base += u; }      // (not all 16 at once: the kernel has 48 VGPRs)
    int run = base + incl - mine;
#pragma unroll 1
    for (int w = wb0; w < wb1; w++) { s_pre[w] = (unsigned short)run; run += __popc(s_bits[w]); }
    __syncthreads();

    // the winners, in the order of their cells' first corners: the record rode along in the lower half of the slot
#pragma unroll
    for (int k = 0; k < INS_CPT; k++)
        if (f[k] != 0xFFFFFFFFu)
        {
            const uint32_t w = f[k] >> 5, bit = f[k] & 31;
            const int pos = (int)s_pre[w] + __popc(s_bits[w] & ((1u << bit) - 1u));
            const uint32_t rec = (uint32_t)bk[k];
            a.pts[a.n_held + pos] = make_float2((float)(rec & 0xFFFu), (float)((rec >> 12) & 0xFFFu));
            a.new_kp[pos] = rec;
        }

    // distribution quality over the occupied cells, the early-outs and the model choice
    if (t < nregions) a.counts[t] = s_count[t];
    if (t == 0)
    {
        int m_used = 0;
        for (int b = 0; b < 16; b++) m_used += s_bucket[b];
        float q = 1.0f;
        if (m_used != 0)
        {
            if (a.small_grid) q = (float)m_used / (float)a.capacity;
            else
            {
                const int ideal = (int)((float)m_used / 16.0f);
                int excess = 0;
                for (int b = 0; b < 16; b++) excess += max(s_bucket[b] - ideal, 0);
                q = 1.0f - ((float)excess / (float)(m_used - ideal));
            }
        }
        int total = 0;
        for (int r = 0; r < nregions; r++) total += s_count[r];
        const int n_total = a.n_held + n_new;
        const int n_eff = (n_total < a.min_samples || q < a.uniformity) ? 0 : n_total;
        const int full = q > a.homography_threshold ? 1 : 0;
        a.result[0] = n_new; a.result[1] = n_eff; a.result[2] = __float_as_int(q); a.result[3] = full; a.result[4] = total;
        *a.d_n = n_eff; *a.d_full = full;
    }
}

} // namespace

int lvk_fast_workspace_bytes(int nregions, int max_rw, int max_rh, size_t* masks_bytes, size_t* scores_bytes)
{
    const int segs_x = (max_rw + TW - 1) / TW;
    *masks_bytes = (size_t)nregions * max_rh * segs_x * sizeof(unsigned long long);
    *scores_bytes = (size_t)nregions * max_rh * max_rw;
    return segs_x;
}

// Enqueues detection for `nregions` regions (device array d_regions).  d_out: nregions x cap packed keypoints
// (x | y << 12 | score << 24, region-local), d_counts: nregions totals (may exceed cap; entries beyond cap are dropped).
int lvk_launch_fast(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols,
                    const FastRegion* d_regions, int nregions, int max_rw, int max_rh,
                    void* d_masks, void* d_scores, uint32_t* d_out, int cap, int* d_counts, const FastRegion* host_regions)
{
    LVK_HIP_REQUIRE(ctx, d_img && d_regions && nregions > 0 && max_rw > 0 && max_rh > 0 && max_rw < 4096 && max_rh < 4096);
    const int segs_x = (max_rw + TW - 1) / TW;
    const dim3 block(TW, TH), grid(segs_x, (max_rh + TH - 1) / TH, nregions);
    FastRegionList inl{};
    const int n_inline = (host_regions && nregions <= LVK_FAST_INLINE_REGIONS) ? nregions : 0;      // host_regions: the same descriptors, readable by the host
    for (int i = 0; i < n_inline; i++) inl.r[i] = host_regions[i];
    hipLaunchKernelGGL(k_fast_detect, grid, block, 0, ctx->stream, (const uint8_t*)d_img, step, rows, cols, d_regions, inl, n_inline, segs_x,
                       (unsigned long long*)d_masks, (uint8_t*)d_scores, max_rh, max_rw, FastCells{});
    hipLaunchKernelGGL(k_fast_compact, dim3(nregions), dim3(1024), 0, ctx->stream, d_regions, inl, n_inline, segs_x,
                       (const unsigned long long*)d_masks, (const uint8_t*)d_scores, max_rh, max_rw, d_out, cap, d_counts);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}


// Detection + the suppression grid on the device (see k_fast_insert): the detect kernel folds the corners into the per-cell slots, ONE
// workgroup leaves the new features behind the held ones in `pts` and the counts / flags the rest of the chain reads.  Regions travel as
// kernel arguments (<= 8).
int lvk_launch_fast_insert(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols, const FastRegion* host_regions, int nregions,
                           int max_rw, int max_rh, void* d_masks, void* d_scores, const FastInsertDesc& d)
{
    LVK_HIP_REQUIRE(ctx, d_img && host_regions && nregions > 0 && nregions <= LVK_FAST_INLINE_REGIONS && max_rw > 0 && max_rh > 0);
    LVK_HIP_REQUIRE(ctx, lvk_fast_insert_fits(d.capacity, nregions, max_rw, max_rh, cols, rows));
    LVK_HIP_REQUIRE(ctx, d.pts && d.new_kp && d.result && d.d_n && d.d_full && d.counts && d.occ && d.cell_first && d.cell_best && d.region_count && d.occ_bucket && d.col_of && d.row_base && d.bucket);
    const int segs_x = (max_rw + TW - 1) / TW;
    FastRegionList inl{};
    for (int i = 0; i < nregions; i++) inl.r[i] = host_regions[i];
    const int nkeys = nregions * max_rh * max_rw;
    const size_t lds = ins_lds_bytes(nkeys);
    if (lds > 48 * 1024)
        LVK_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_fast_insert), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 block(TW, TH), grid(segs_x, (max_rh + TH - 1) / TH, nregions);
    hipLaunchKernelGGL(k_fast_detect, grid, block, 0, ctx->stream, (const uint8_t*)d_img, step, rows, cols, (const FastRegion*)nullptr, inl, nregions, segs_x,
                       (unsigned long long*)d_masks, (uint8_t*)d_scores, max_rh, max_rw, FastCells{d.col_of, d.row_base, d.cell_first, (unsigned long long*)d.cell_best, d.region_count});
    FastInsertArgs a{d.bucket, d.cell_first, (unsigned long long*)d.cell_best, d.region_count, d.capacity, d.small_grid ? 1 : 0, d.n_held, d.min_samples, d.uniformity, d.homography_threshold, nkeys,
                     d.pts, d.new_kp, d.result, d.d_n, d.d_full, d.counts, {}, {}};
    for (int i = 0; i < (d.capacity + 31) / 32; i++) a.occ[i] = d.occ[i];
    for (int i = 0; i < 16; i++) a.occ_bucket[i] = d.occ_bucket[i];
    hipLaunchKernelGGL(k_fast_insert, dim3(1), dim3(INS_NT), lds, ctx->stream, nregions, a);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

// what the kernels cover, for the caller's choice between them and the host loop; and the initial state of the per-cell slots
bool lvk_fast_insert_fits(int cells, int nregions, int max_rw, int max_rh, int cols, int rows)
{
    if (!(cells > 0 && cells <= INS_MAX_CELLS && nregions > 0 && nregions <= LVK_FAST_INLINE_REGIONS && cols < 4096 && rows < 4096 && max_rw < 4096 && max_rh < 4096)) return false;
    const long nkeys = (long)nregions * max_rh * max_rw;
    return nkeys <= INS_MAX_KEYS && ins_lds_bytes((int)nkeys) <= 60 * 1024;
}
int lvk_fast_cells_reset(lvk_hip_ctx* ctx, uint32_t* d_first, void* d_best, int cells, int* d_region_count)
{
    LVK_HIP_CHECK(ctx, hipMemsetAsync(d_region_count, 0, LVK_FAST_INLINE_REGIONS * sizeof(int), ctx->stream));
    LVK_HIP_CHECK(ctx, hipMemsetAsync(d_first, 0xFF, (size_t)cells * sizeof(uint32_t), ctx->stream));
    LVK_HIP_CHECK(ctx, hipMemsetAsync(d_best, 0, (size_t)cells * sizeof(unsigned long long), ctx->stream));
    return LVK_HIP_OK;
}

extern "C" {

// Synchronous test entry point: regions = nregions x {x, y, w, h, threshold, active} ints (host); out = nregions x cap
// packed keypoints (host), counts = nregions ints (host).
int lvk_hip_fast_detect(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols,
                        const int* regions, int nregions, uint32_t* out, int cap, int* counts)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, regions && out && counts && nregions > 0 && cap > 0);
    std::vector<FastRegion> rg((size_t)nregions);
    int max_rw = 1, max_rh = 1;
    for (int i = 0; i < nregions; i++)
    {
        rg[i] = FastRegion{regions[6 * i], regions[6 * i + 1], regions[6 * i + 2], regions[6 * i + 3], regions[6 * i + 4], regions[6 * i + 5]};
        LVK_HIP_REQUIRE(ctx, rg[i].x >= 0 && rg[i].y >= 0 && rg[i].w > 0 && rg[i].h > 0 && rg[i].x + rg[i].w <= cols && rg[i].y + rg[i].h <= rows);
        max_rw = std::max(max_rw, rg[i].w); max_rh = std::max(max_rh, rg[i].h);
    }
    size_t mb, sb;
    lvk_fast_workspace_bytes(nregions, max_rw, max_rh, &mb, &sb);
    void *d_masks = nullptr, *d_scores = nullptr, *d_regions = nullptr; uint32_t* d_out = nullptr; int* d_counts = nullptr;
    LVK_HIP_CHECK(ctx, hipMalloc(&d_masks, mb));
    LVK_HIP_CHECK(ctx, hipMalloc(&d_scores, sb));
    LVK_HIP_CHECK(ctx, hipMalloc(&d_regions, rg.size() * sizeof(FastRegion)));
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_out, (size_t)nregions * cap * sizeof(uint32_t)));
    LVK_HIP_CHECK(ctx, hipMalloc((void**)&d_counts, nregions * sizeof(int)));
    LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_regions, rg.data(), rg.size() * sizeof(FastRegion), hipMemcpyHostToDevice, ctx->stream));
    int rc = lvk_launch_fast(ctx, d_img, step, rows, cols, (const FastRegion*)d_regions, nregions, max_rw, max_rh, d_masks, d_scores, d_out, cap, d_counts);
    if (rc == LVK_HIP_OK)
    {
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, (size_t)nregions * cap * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(counts, d_counts, nregions * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    (void)hipFree(d_masks); (void)hipFree(d_scores); (void)hipFree(d_regions); (void)hipFree(d_out); (void)hipFree(d_counts);
    return rc;
}

} // extern "C"
