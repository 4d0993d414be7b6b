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

__global__ __launch_bounds__(TW * TH)
void k_fast_detect(const uint8_t* __restrict__ img, int step, int rows, int cols,
                   const FastRegion* __restrict__ regions, FastRegionList inl, int n_inline, int segs_x,
                   unsigned long long* __restrict__ masks, uint8_t* __restrict__ scores, int max_rh, int max_rw)
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
        if (keep) scores[((long)blockIdx.z * max_rh + ly) * max_rw + lx] = (uint8_t)s;
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
// cells' FIRST corners.  One workgroup computes that straight from k_fast_detect's ballots -- no ordered corner list is materialised, no
// host round trip splits the frame's chain of kernels -- and appends the new points to the list the optical-flow kernel reads:
//   1. exclusive scan of the per-word popcounts (words in region / row / segment order) -> the sequence number of every corner;
//   2. per corner: atomicMin of the sequence number and atomicMax of (score << 24 | ~sequence) on its cell (LDS);
//   3. per corner again: a corner that IS its cell's first one opens a list position (scan of those counts per word); the winner of
//      the cell is looked up by its sequence number (binary search over the scan, k-th set bit of the word);
//   4. SpatialMap::distribution_quality (Data/SpatialMap.tpp:589-625) over propagated + new cells, the two early-outs of
//      FrameTracker::track (Vision/FrameTracker.cpp:127-131) and the homography / similarity choice (:170) for the kernels that follow.
struct FastInsertArgs
{
    const uint16_t* col_of; const uint32_t* row_base; const uint8_t* bucket;      // device tables of the grid (FeatureGridH)
    const uint32_t* occ;                  // cells that hold a propagated feature, one bit each (device-visible host memory)
    int capacity, small_grid, n_held, min_samples; float uniformity, homography_threshold;
    float2* pts;                          // the optical flow's point list: new points go to pts[n_held ...]
    uint32_t* new_kp;                     // x | y << 12 | score << 24 (frame coordinates) of the new features, list order
    int* result;                          // [0] new features, [1] points to track (0: early out), [2] quality (float bits), [3] full homography, [4] corners
    int* d_n; int* d_full;                // [1] and [3] again in device memory, for the kernels of the chain
    int* counts;                          // raw corner count per region
};

constexpr int INS_NT = 1024, INS_MAX_WORDS = 4096, INS_MAX_CELLS = 4096;

__global__ __launch_bounds__(INS_NT)
void k_fast_insert(FastRegionList inl, int nregions, int segs_x, const unsigned long long* __restrict__ masks, const uint8_t* __restrict__ scores,
                   int max_rh, int max_rw, FastInsertArgs a)
{
    LVK_TRACKER_PRIORITY();
    __shared__ uint32_t s_first[INS_MAX_CELLS], s_best[INS_MAX_CELLS];
    __shared__ int s_wbase[INS_MAX_WORDS + 1];
    __shared__ unsigned short s_fbase[INS_MAX_WORDS];
    __shared__ uint32_t s_occ[INS_MAX_CELLS / 32];
    __shared__ int s_wave[INS_NT / 64], s_carry, s_woff[LVK_FAST_INLINE_REGIONS + 1], s_bucket[16], s_used;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    for (int i = t; i < a.capacity; i += INS_NT) { s_first[i] = 0xFFFFFFFFu; s_best[i] = 0u; }
    for (int i = t; i < (a.capacity + 31) / 32; i += INS_NT) s_occ[i] = a.occ[i];
    if (t < 16) s_bucket[t] = 0;
    if (t == 0)
    {
        int off = 0;
        for (int r = 0; r < nregions; r++) { s_woff[r] = off; if (inl.r[r].active) off += inl.r[r].h * ((inl.r[r].w + TW - 1) / TW); }
        s_woff[nregions] = off; s_carry = 0; s_used = 0;
    }
    __syncthreads();
    const int nwords = s_woff[nregions];
    // word w of the flattened order -> (region, row, segment) and the ballot word itself
    auto word_at = [&](int w, int& r, int& ly, int& sg) -> unsigned long long {
        r = 0;
        while (r + 1 < nregions && w >= s_woff[r + 1]) r++;
        while (!inl.r[r].active) r++;                                        // (an inactive region owns no words: skip to the owner)
        const int rsegs = (inl.r[r].w + TW - 1) / TW, k = w - s_woff[r];
        ly = k / rsegs; sg = k - ly * rsegs;
        return masks[((long)r * max_rh + ly) * segs_x + sg];
    };

    // 1. sequence numbers: exclusive scan of the popcounts
    for (int chunk = 0; chunk < nwords; chunk += INS_NT)
    {
        const int w = chunk + t;
        int r, ly, sg;
        const int cnt = w < nwords ? __popcll(word_at(w, r, ly, sg)) : 0;
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int q = 0; q < wave; q++) wave_off += s_wave[q];
        if (w < nwords) s_wbase[w] = s_carry + wave_off + incl - cnt;
        __syncthreads();
        if (t == INS_NT - 1) s_carry = s_carry + wave_off + incl;
        __syncthreads();
    }
    if (t == 0) s_wbase[nwords] = s_carry;
    __syncthreads();
    const int total = s_wbase[nwords];
    if (t < nregions) a.counts[t] = inl.r[t].active ? s_wbase[s_woff[t + 1]] - s_wbase[s_woff[t]] : 0;

    // 2. first corner and strongest corner of every cell
    for (int w = t; w < nwords; w += INS_NT)
    {
        int r, ly, sg;
        unsigned long long m = word_at(w, r, ly, sg);
        uint32_t seq = (uint32_t)s_wbase[w];
        const int gy = inl.r[r].y + ly;
        while (m)
        {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int lx = sg * TW + bit;
            const uint32_t ci = a.row_base[gy] + a.col_of[inl.r[r].x + lx];
            if (!((s_occ[ci >> 5] >> (ci & 31)) & 1u))
            {
                const uint32_t sc = scores[((long)r * max_rh + ly) * max_rw + lx];
                atomicMin(&s_first[ci], seq);
                atomicMax(&s_best[ci], (sc << 24) | (0xFFFFFFu - seq));
            }
            seq++;
        }
    }
    __syncthreads();

    // 3. list positions: corners that are their cell's first one, in sequence order
    s_carry = 0;                                       // (every thread read `total` above; the barrier below orders this store)
    __syncthreads();
    for (int chunk = 0; chunk < nwords; chunk += INS_NT)
    {
        const int w = chunk + t;
        int cnt = 0;
        if (w < nwords)
        {
            int r, ly, sg;
            unsigned long long m = word_at(w, r, ly, sg);
            uint32_t seq = (uint32_t)s_wbase[w];
            const int gy = inl.r[r].y + ly;
            while (m)
            {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t ci = a.row_base[gy] + a.col_of[inl.r[r].x + sg * TW + bit];
                cnt += s_first[ci] == seq ? 1 : 0;     // (an occupied cell keeps first = ~0, which is no sequence number)
                seq++;
            }
        }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int q = 0; q < wave; q++) wave_off += s_wave[q];
        if (w < nwords) s_fbase[w] = (unsigned short)(s_carry + wave_off + incl - cnt);
        __syncthreads();
        if (t == INS_NT - 1) s_carry = s_carry + wave_off + incl;
        __syncthreads();
    }
    const int n_new = s_carry;
    for (int w = t; w < nwords; w += INS_NT)
    {
        int r, ly, sg;
        unsigned long long m = word_at(w, r, ly, sg);
        uint32_t seq = (uint32_t)s_wbase[w];
        int pos = s_fbase[w];
        const int gy = inl.r[r].y + ly;
        while (m)
        {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int gx = inl.r[r].x + sg * TW + bit;
            const uint32_t ci = a.row_base[gy] + a.col_of[gx];
            if (s_first[ci] == seq)
            {
                const uint32_t key = s_best[ci], sb = 0xFFFFFFu - (key & 0xFFFFFFu);
                int bx = gx, by = gy;
                if (sb != seq)
                {
                    // the strongest corner of the cell is a later one: find its word (largest wb with wbase[wb] <= sb), then its bit
                    int lo = 0, hi = nwords - 1;
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((uint32_t)s_wbase[mid] <= sb) lo = mid; else hi = mid - 1; }
                    int r2, ly2, sg2;
                    unsigned long long m2 = word_at(lo, r2, ly2, sg2);
                    for (int k = (int)sb - s_wbase[lo]; k > 0; k--) m2 &= m2 - 1;
                    bx = inl.r[r2].x + sg2 * TW + (__ffsll((long long)m2) - 1); by = inl.r[r2].y + ly2;
                }
                a.pts[a.n_held + pos] = make_float2((float)bx, (float)by);
                a.new_kp[pos] = (uint32_t)bx | ((uint32_t)by << 12) | (key & 0xFF000000u);
                pos++;
            }
            seq++;
        }
    }

    // 4. distribution quality over the occupied cells, the early-outs and the model choice
    int used = 0;
    for (int ci = t; ci < a.capacity; ci += INS_NT)
        if (((s_occ[ci >> 5] >> (ci & 31)) & 1u) || s_first[ci] != 0xFFFFFFFFu) { used++; if (!a.small_grid) atomicAdd(&s_bucket[a.bucket[ci]], 1); }
    if (used) atomicAdd(&s_used, used);
    __syncthreads();
    if (t == 0)
    {
        const int m_used = s_used;
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
                       (unsigned long long*)d_masks, (uint8_t*)d_scores, max_rh, max_rw);
    hipLaunchKernelGGL(k_fast_compact, dim3(nregions), dim3(1024), 0, ctx->stream, d_regions, inl, n_inline, segs_x,
                       (const unsigned long long*)d_masks, (const uint8_t*)d_scores, max_rh, max_rw, d_out, cap, d_counts);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}


// Detection + the suppression grid on the device (see k_fast_insert): same detect kernel, then ONE workgroup that leaves the new features
// behind the held ones in `pts` and the counts / flags the rest of the chain reads.  Regions travel as kernel arguments (<= 8).
int lvk_launch_fast_insert(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols, const FastRegion* host_regions, int nregions,
                           int max_rw, int max_rh, void* d_masks, void* d_scores, const FastInsertDesc& d)
{
    LVK_HIP_REQUIRE(ctx, d_img && host_regions && nregions > 0 && nregions <= LVK_FAST_INLINE_REGIONS && max_rw > 0 && max_rh > 0 && max_rw < 4096 && max_rh < 4096);
    LVK_HIP_REQUIRE(ctx, d.capacity > 0 && d.capacity <= INS_MAX_CELLS && d.pts && d.new_kp && d.result && d.d_n && d.d_full && d.counts && d.occ);
    const int segs_x = (max_rw + TW - 1) / TW;
    int nwords = 0;
    FastRegionList inl{};
    for (int i = 0; i < nregions; i++) { inl.r[i] = host_regions[i]; if (inl.r[i].active) nwords += inl.r[i].h * ((inl.r[i].w + TW - 1) / TW); }
    LVK_HIP_REQUIRE(ctx, nwords <= INS_MAX_WORDS);
    const dim3 block(TW, TH), grid(segs_x, (max_rh + TH - 1) / TH, nregions);
    hipLaunchKernelGGL(k_fast_detect, grid, block, 0, ctx->stream, (const uint8_t*)d_img, step, rows, cols, (const FastRegion*)nullptr, inl, nregions, segs_x,
                       (unsigned long long*)d_masks, (uint8_t*)d_scores, max_rh, max_rw);
    const FastInsertArgs a{d.col_of, d.row_base, d.bucket, d.occ, d.capacity, d.small_grid ? 1 : 0, d.n_held, d.min_samples, d.uniformity, d.homography_threshold,
                           d.pts, d.new_kp, d.result, d.d_n, d.d_full, d.counts};
    hipLaunchKernelGGL(k_fast_insert, dim3(1), dim3(INS_NT), 0, ctx->stream, inl, nregions, segs_x, (const unsigned long long*)d_masks, (const uint8_t*)d_scores,
                       max_rh, max_rw, a);
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

int lvk_fast_insert_limits(int* max_cells, int* max_words) { *max_cells = INS_MAX_CELLS; *max_words = INS_MAX_WORDS; return TW; }

extern "C" {

// Synchronous test entry point: regions = nregions x {x, y, w, h, threshold, active} ints (host); out = nregions x cap
// packed keypoints (host), counts = nregions ints (host).
int lvk_hip_fast_detect(lvk_hip_ctx* ctx, const void* d_img, int step, int rows, int cols,
                        const int* regions, int nregions, uint32_t* out, int cap, int* counts)
{
    if (!ctx) return LVK_HIP_ERR_ARG;
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
