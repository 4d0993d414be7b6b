// The stabilization filter proper: per-frame orchestration of the HIP kernels plus the host state machines.
//
// Replaces (reference, paths relative to LiveVisionKit/):
//   StabilizationFilter::{configure,filter,restart,ready,reset_context,frame_delay,stable_region}  Filters/StabilizationFilter.cpp:42-206
//   FrameTracker::{configure,track,restart,estimate_global_motion}                                  Vision/FrameTracker.cpp:57-196,325-375
//   FeatureDetector::{detect,propagate,reset} bookkeeping                                           Vision/FeatureDetector.cpp:114-214
//
// Per frame, on the context's stream:
//   luma (Y / BGR / RGB -> gray) + INTER_AREA downscale -> pyramid (pyrDown x3, Scharr x4)   [imgproc.hip]
//   FAST-9/16 + NMS per due region, ordered compaction                                  [fast.hip]      -> host: suppression grid
//   pyramidal LK, one block per feature, one wave per level (points read from pinned host memory) [pyrlk.hip]
//   fast_filter in the reference's swap-erase order                                     [motion.hip k_match_compact]
//   RANSAC hypotheses + local optimisation, pair count from the device                  [motion.hip]    -> ONE sync; host: ageing, propagate, QA, smoothing
//   (vector-field preset / lens modes: sync after LK, point filter + mesh solve on the host)
// and on the bulk stream (overlap mode; else the same stream):
//   4:2:0 ingest of the new frame, EASU remap of the delayed frame (homography or in-kernel mesh, optionally with the lens
//   pre-warp composed in, optionally with the 4:2:0 egress fused)                         [ingest.hip, remap.hip]
// Packed frames are never copied: the filter borrows the caller's device buffer until that frame has been emitted
// (the reference moves the input frame into its queue, StabilizationFilter.cpp:118).
#include "stab_state.hpp"

using namespace lvkstab;

// FrameTracker::track (FrameTracker.cpp:108-196)
int lvk_hip_stab::track(const QueuedFrame& f, const void* luma, int luma_step, int luma_pix, int luma_channel, WarpMeshF& motion, bool& have_motion)
{
    have_motion = false;
    tracking_stability = 0.0f;
    last_detected = last_matched = 0; last_distribution = 0.0f;
    hipStream_t st = ctx->stream;
    int rc;

    cur ^= 1;
    std::swap(prev_w, cur_w); std::swap(prev_h, cur_h);
    cur_w = s.detection_width; cur_h = s.detection_height;
    DevicePyramid& C = pyr[cur];
    DevicePyramid& P = pyr[cur ^ 1];
    int pe = 0;
    // (look-ahead: this frame's downscale and pyramid were put behind the previous push's chain -- same planes, same geometry, announced for
    //  exactly this push --, so `C` holds them already, in stream order)
    const bool built_ahead = ahead_built_for == push_seq && ahead_built.same(luma, luma_step, luma_pix, luma_channel, f.rows, f.cols) && pyr_w == cur_w && pyr_h == cur_h;
    ahead_built = LumaAhead(); ahead_built_for = 0;
    if (built_ahead) lookahead_frames++;
    else
    {
        pe = prof_begin(LVK_STAGE_DOWNSCALE);
        if ((rc = lvk_launch_luma_area_resize(ctx, luma, luma_step, luma_pix, luma_channel, f.rows, f.cols, const_cast<uint8_t*>(C.args.lv[0].img), C.args.lv[0].step, cur_h, cur_w)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        pe = prof_begin(LVK_STAGE_PYRAMID);
        if ((rc = C.build(ctx)) != LVK_HIP_OK) return rc;
        prof_end(pe);
    }
    trace.mark(HostTrace::DOWN_PYR_LAUNCH);
    // The previous frame's list bookkeeping (fast_filter, ageing, the suppression grid's re-seed) runs HERE, in the shadow of the two kernels
    // just launched: the next kernel of this frame (optical flow) cannot start before they are done anyway, while at the end of the
    // previous push those 9 us delayed this frame's first launch -- the tracker chain and the host take turns, and that turn-taking, not
    // either of them, bounds the frame rate.
    finish_post();
    if (post_error) { post_error = false; return fail(LVK_HIP_ERR_RUNTIME, "GPU-side fast_filter disagrees with the host's"); }
    // A push that ends before the chain's synchronisation below still has the downscale reading the caller's luma (the plane entries: luma_pix 1 for
    // planar formats, 2 / 4 for packed 4:2:2 / AYUV -- "the input planes are consumed before the call returns"): wait for it.  Packed frames (luma_pix
    // 3) stay borrowed until they are released.
    auto leave_early = [&]() -> int { if (luma_pix != 3) LVK_HIP_CHECK(ctx, hipStreamSynchronize(st)); return LVK_HIP_OK; };
    if (!initialized || cur_w != prev_w || cur_h != prev_h) { initialized = true; return leave_early(); }

    // ---- FeatureDetector::detect
    grid.plan(plan);
    bool any = false;
    for (size_t i = 0; i < plan.size(); i++) { h_regions[i] = plan[i]; any = any || plan[i].active; }
    // On a frame on which the detector runs, the corners go through the suppression grid ON THE DEVICE, inside the chain (k_fast_insert): the
    // new features land behind the held ones in the flow kernel's point list, and the kernels that follow take the point count and the
    // model choice from device memory -- one chain and one synchronisation per frame, like a frame without detection.  The host loop stays
    // for what the kernel does not cover (huge grids, regions off the pixel grid, more points than the on-device fast_filter takes).
    const size_t n_held = grid.held.size();
    const size_t n_bound = std::min(cap_features, n_held + (grid.capacity() - grid.used_cells()));      // every free cell takes at most one corner
    const bool dev_insert = any && device_grid && grid.device_insert_ok() && (int)plan.size() <= LVK_FAST_INLINE_REGIONS &&
                            lvk_fast_insert_fits((int)grid.capacity(), (int)plan.size(), fast_max_rw, fast_max_rh, cur_w, cur_h) && n_bound >= 1 && n_bound <= 4096;
    float distribution = 0.0f;
    if (dev_insert)
    {
        for (size_t i = 0; i < n_held; i++) h_pts[i] = make_float2(grid.held[i].x, grid.held[i].y);
        int occ_bucket[16];
        grid.occupancy(h_occ, occ_bucket);
        const FastInsertDesc d{d_grid_col, d_grid_row, d_grid_bucket, d_cell_first, d_cell_best, d_region_count, h_occ, occ_bucket, (int)grid.capacity(), grid.grid_cols() <= 4 || grid.grid_rows() <= 4, (int)n_held,
                               s.min_motion_samples, s.uniformity_threshold, HOMOGRAPHY_DISTRIBUTION_THRESHOLD, h_pts, h_new_kp, h_insert, d_n_points, d_full, h_fast_counts};
        pe = prof_begin(LVK_STAGE_FAST);
        if ((rc = lvk_launch_fast_insert(ctx, C.args.lv[0].img, C.args.lv[0].step, cur_h, cur_w, h_regions, (int)plan.size(), fast_max_rw, fast_max_rh,
                                         d_fast_masks, d_fast_scores, d)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        device_grid_frames++;
    }
    else
    {
    if (any)
    {
        // The small per-frame parameter / result blocks live in pinned, device-visible host memory: the kernels read the
        // region descriptors from it and write the keypoint list straight into it (posted PCIe writes), so the only host
        // call besides the launches is the stream synchronisation.
        pe = prof_begin(LVK_STAGE_FAST);
        if ((rc = lvk_launch_fast(ctx, C.args.lv[0].img, C.args.lv[0].step, cur_h, cur_w, h_regions, (int)plan.size(), fast_max_rw, fast_max_rh,
                                  d_fast_masks, d_fast_scores, h_fast_out, fast_cap, h_fast_counts, h_regions)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
        trace.mark(HostTrace::FAST_SYNC);
        host_grid_frames++;
    }
    for (size_t i = 0; i < plan.size(); i++)
        if (plan[i].active) grid.absorb(i, h_fast_out + i * (size_t)fast_cap, std::min(h_fast_counts[i], fast_cap));
    distribution = grid.finish(tracked);
    last_distribution = distribution; last_detected = (int)tracked.size();
    if (tracked.size() < (size_t)s.min_motion_samples || distribution < s.uniformity_threshold) { tracked.clear(); return leave_early(); }
    if (tracked.size() > cap_features) return fail(LVK_HIP_ERR_RUNTIME, "feature count exceeds the suppression grid capacity");
    }

    trace.mark(HostTrace::GRID);
    // ---- sparse optical flow prev -> cur
    // (device grid: n is the upper bound the kernels are launched for; they read the count itself from d_n_points)
    int n = dev_insert ? (int)n_bound : (int)tracked.size();
    if (!dev_insert) for (int i = 0; i < n; i++) h_pts[i] = make_float2(tracked[i].x, tracked[i].y);
    const int* dn = dev_insert ? d_n_points : nullptr;
    const int* dfull = dev_insert ? d_full : nullptr;
    const bool full = distribution > HOMOGRAPHY_DISTRIBUTION_THRESHOLD;         // (device grid: decided by the kernel, d_full)
    // Global-motion mode without a lens model: the whole chain optical flow -> fast_filter -> RANSAC runs on the GPU without a
    // host round trip in between (the flow kernel reads the points from pinned host memory, k_match_compact reproduces the host's
    // swap-erase order); the host synchronises once and then repeats the cheap bookkeeping on its own copies.
    const bool chained = n <= 4096;
    const bool field = s.track_local_motions != 0;
    // Global-motion chain of a caller that WAITS for every frame (the OBS plugin's pattern): k_ransac_finalize announces its results with a word
    // in host memory, and the wait below spins on it.  A free-running caller keeps the completion event: there the remap must not be launched
    // earlier than it is -- its first wave of blocks would land on the conversion behind the chain (profiles/r05_ab_host_signal_word.txt).
    LvkHostSignal done{nullptr, 0};
    if (chained && !field && !caller_runs_free)
    {
        if (!h_chain_flag) { LVK_HIP_CHECK(ctx, hipHostMalloc((void**)&h_chain_flag, 64, hipHostMallocCoherent)); *h_chain_flag = 0; }
        done = LvkHostSignal{h_chain_flag, ++chain_seq};
    }
    pe = prof_begin(LVK_STAGE_PYRLK);
    if (chained)
    {
        // fused lens mode: the motion is estimated between lens-corrected positions (what the reference chain LC -> VS tracks); the flow
        // kernel writes them itself (d_und: previous | matched)
        if ((rc = lvk_launch_pyrlk(ctx, P.args, C.args, h_pts, n, d_matched, d_status, LK_WIN, LK_WIN, LK_ITERS, LK_EPS, LK_MIN_EIG, d_pts,
                                   lens ? &lens_model : nullptr, (double)f.cols / (double)cur_w, (double)f.rows / (double)cur_h, lens ? d_und : nullptr, dn)) != LVK_HIP_OK) return rc;
        const bool fused_compact = !field && n <= LVK_COMPACT_RANSAC_MAX;        // the RANSAC's first kernel compacts the flow result itself
        if (!fused_compact && (rc = lvk_launch_match_compact(ctx, d_pts, d_matched, d_status, n, d_p1, d_p1 + cap_features, d_count, h_count, h_matched, h_status,
                                                             lens ? d_und : nullptr, (float)cur_w, (float)cur_h, dn)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        pe = prof_begin(LVK_STAGE_MOTION);
        if (field)
        {
            // estimate_local_motions (FrameTracker.cpp:200-321): least-squares mesh through the matches, solved on the device
            if ((rc = lvk_launch_mesh_solve(mesh_dev, st, d_mesh_scratch, d_p1, d_p1 + cap_features, d_count, n, s.min_motion_samples, (float)cur_w, (float)cur_h,
                                            s.temporal_smoothing, s.acceptance_threshold, h_offsets, h_mask, h_mesh_status)) != LVK_HIP_OK) return rc;
        }
        else if (fused_compact)
        {
            if ((rc = lvk_launch_compact_ransac(ctx, d_pts, d_matched, d_status, n, d_p1, d_p1 + cap_features, d_count, h_count, h_matched, h_status,
                                                lens ? d_und : nullptr, (float)cur_w, (float)cur_h,
                                                s.acceptance_threshold, (double)cur_w, (double)cur_h, full, d_ransac_ws, h_H, h_ninl, h_mask, dn, dfull, done)) != LVK_HIP_OK) return rc;
        }
        else if ((rc = lvk_launch_ransac(ctx, d_p1, d_p1 + cap_features, n, s.acceptance_threshold, (double)cur_w, (double)cur_h, full, d_ransac_ws, h_H, h_ninl, h_mask, d_count, dfull, done)) != LVK_HIP_OK) return rc;
        prof_end(pe);
    }
    else
    {
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_pts, h_pts, n * sizeof(float2), hipMemcpyHostToDevice, st));
        if ((rc = lvk_launch_pyrlk(ctx, P.args, C.args, d_pts, n, h_matched, h_status, LK_WIN, LK_WIN, LK_ITERS, LK_EPS, LK_MIN_EIG)) != LVK_HIP_OK) return rc;
        prof_end(pe);
    }
    trace.mark(HostTrace::LK_LAUNCH);
    bool chain_event_armed = false;              // local: an error return below must not leave a stale event armed for the next push
    if (deferred_ingest && tracker_ingest_capable)
    {
        // Where the conversion goes: a bulk stream that is idle (a caller that synchronises every frame) takes it now, next to the
        // chain -- the remap that follows then has the GPU to itself; a bulk stream that is still busy with the previous remap (a
        // free-running caller) would only get to it after that, so it goes behind the chain on this stream, and the push waits for
        // the chain through an event instead of for the stream.
        const hipError_t q = hipStreamQuery(remap_stream);
        if (q != hipSuccess) (void)hipGetLastError();
        // (host-resident frames whose chroma planes are still on the link: behind the chain as well -- on the bulk stream the conversion would
        //  hold the output remap back until they have arrived)
        // (a remap whose stores cross the host link leaves the bulk stream time to spare -- 283 us of link time per frame against ~235 us of a
        //  slowed-down chain --: an announced host frame, whose planes have arrived, is converted there; 3 320 against 3 215 frames/s, and the
        //  pushes on which the detector runs no longer stand out: p90 0.312 instead of 0.364 ms)
        ingest_on_tracker = ingest_placement == 1 || (ingest_placement == 0 && ((q == hipErrorNotReady && !host_direct_now) || ingest_wait[0] != nullptr));
    }
    // (an announced next frame: its downscale + pyramid go behind the chain too, so the push waits for the chain through the event as well)
    const bool build_ahead = chained && ahead_announced.luma != nullptr && pyr_w == cur_w && pyr_h == cur_h;
    if (chained && ((deferred_ingest && tracker_ingest_capable && ingest_on_tracker) || build_ahead))
    {
        if (!chain_done) LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&chain_done, hipEventDisableTiming));
        LVK_HIP_CHECK(ctx, hipEventRecord(chain_done, st)); chain_event_armed = true;
    }
    if (deferred_ingest && (rc = run_deferred_ingest()) != LVK_HIP_OK) return rc;
    if (build_ahead && (rc = launch_build_ahead(P, cur_w, cur_h)) != LVK_HIP_OK) return rc;
    if (lens && !chained)
    {
        if ((rc = lvk_launch_lens_undistort(ctx, st, lens_model, (double)f.cols / (double)cur_w, (double)f.rows / (double)cur_h,
                                            d_pts, n, h_matched, n, h_und)) != LVK_HIP_OK) return rc;
    }
    // (Round 5, measured and rejected: putting this push's output remap on the bulk stream HERE, behind a hipStreamWaitValue32 the host releases
    //  with one store once the smoother has the correction -- the launch left the host's turn (7.4 -> 0.3 us) and the rate fell 2 %: the chain
    //  ran 3.7 us slower and the push waited 5.5 us longer for the new frame's conversion, which sits behind the chain on this stream.  The
    //  cycle is bound by this STREAM, not by the host's turn.  profiles/r05_ab_prelaunch_remap.txt, scripts/probes/waitvalue_probe.hip.)
    bool have_results = false;
    if (done.flag)
    {
        // The kernel's completion SIGNAL comes ~3 us after its results (end-of-kernel write-back, the command processor's signal, the runtime's
        // wake-up); what follows here needs the results, not the signal, and everything that goes on a stream is ordered by the stream.
        // What the host may read once it has seen the word (acquire): k_ransac_finalize's own results (h_H, h_ninl, h_mask -- its
        // __threadfence_system + release store put them ahead of the word) AND the host mirrors the EARLIER kernels of the chain wrote (h_count,
        // h_matched, h_status, h_insert, h_new_kp, h_fast_counts): those kernels had completed -- stream order; a kernel's end is a release at
        // system scope for coherent host memory, which is what every one of these blocks is allocated as (hipHostMallocCoherent,
        // stab_configure.hip) -- before the finalize kernel started, so they are ordered ahead of its fence as well.
        // The spin is BOUNDED by signal_spin_us (default 400 us, a few chain times: K streams per GPU share the chip): a word that has not
        // changed by then -- a slow neighbour, a faulted kernel, a lost device -- falls through to the wait that blocks in the runtime and
        // reports errors; the thread no longer burns its core for the rest of a long chain.
        const unsigned expect = signal_test_lose ? done.seq + 0x40000000u : done.seq;      // (tests: a word that never arrives)
        const auto t0 = std::chrono::steady_clock::now();
        const auto budget = std::chrono::microseconds(signal_spin_us);
        for (unsigned spins = 0;; spins++)
        {
            if (__atomic_load_n(done.flag, __ATOMIC_ACQUIRE) == expect) { have_results = true; break; }
            if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > budget) break;
            lvk_cpu_relax();
        }
        sched[have_results ? LVK_SCHED_WAIT_SIGNAL_WORD : LVK_SCHED_WAIT_WORD_TIMEOUT]++;
    }
    if (!have_results)
    {
        if (chain_event_armed) LVK_HIP_CHECK(ctx, hipEventSynchronize(chain_done));
        else LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
        sched[LVK_SCHED_WAIT_EVENT]++;
    }
    trace.mark(HostTrace::LK_SYNC);

    if (dev_insert)
    {
        // what the host loop would have left behind: thresholds, the feature list (held + new), the distribution quality -- recomputed here
        // from the new features and held against the kernel's own figure
        const int n_new = h_insert[0];
        float q_dev; std::memcpy(&q_dev, &h_insert[2], sizeof(float));
        if (n_new < 0 || (size_t)n_new > grid.capacity()) return fail(LVK_HIP_ERR_RUNTIME, "device suppression grid returned an impossible count");
        distribution = grid.finish_device(tracked, h_new_kp, n_new, h_fast_counts, fast_cap);
        last_distribution = distribution; last_detected = (int)tracked.size();
        if (distribution != q_dev || tracked.size() != n_held + (size_t)n_new)
            return fail(LVK_HIP_ERR_RUNTIME, "device suppression grid disagrees with the host's bookkeeping");
        // FrameTracker.cpp:127-131 (the kernels of the chain saw a point count of zero and did nothing)
        if (tracked.size() < (size_t)s.min_motion_samples || distribution < s.uniformity_threshold) { tracked.clear(); return LVK_HIP_OK; }
        if (tracked.size() > cap_features) return fail(LVK_HIP_ERR_RUNTIME, "feature count exceeds the suppression grid capacity");
        n = (int)tracked.size();
    }
    if (chained)
    {
        // everything the remap launch needs is in the pinned result block; the list bookkeeping follows in finish_post()
        const int m = *h_count;
        last_matched = m;
        if (m < 0 || m > n) return fail(LVK_HIP_ERR_RUNTIME, "GPU-side fast_filter returned an impossible count");
        if ((size_t)m < (size_t)s.min_motion_samples) { tracked.clear(); return LVK_HIP_OK; }
        motion = WarpMeshF(s.motion_height, s.motion_width);
        if (field)
        {
            if (*h_mesh_status != 0) { tracked.clear(); return LVK_HIP_OK; }             // no estimate this frame (identity motion)
            std::memcpy(motion.off.data(), h_offsets, motion.off.size() * sizeof(float));
        }
        else
        {
            std::memcpy(last_H, h_H, sizeof(last_H));
            motion.from_homography(last_H, (float)cur_w, (float)cur_h);
        }
        size_t inliers = 0;
        for (int i = 0; i < m; i++) inliers += h_mask[i] ? 1 : 0;
        tracking_stability = (float)inliers / (float)m;                                  // ratio_of(inlier_status, 1)
        have_motion = true;
        post_n = n; post_m = m;
        trace.mark(HostTrace::POST);
        return LVK_HIP_OK;
    }

    if (lens && !chained)                     // (the chained path has folded this test into the status flags on the GPU)
    {
        // a match whose corrected positions leave the tracking region is not visible in the lens-corrected frame: drop it
        const float w = (float)cur_w, h = (float)cur_h;
        auto inside = [&](const float2& p) { return p.x >= 0.0f && p.x < w && p.y >= 0.0f && p.y < h; };
        for (int k = 0; k < n; k++)
            if (!(inside(h_und[k]) && inside(h_und[n + k]))) h_status[k] = 0;
    }
    // fast_filter(features, tracked points, matched points; keep = status): back-to-front swap-erase (Container.tpp:97-121)
    int m = n;
    for (int k = n - 1; k >= 0; k--)
        if (!h_status[k])
        {
            m--;
            std::swap(tracked[k], tracked[m]);
            std::swap(h_pts[k], h_pts[m]);
            std::swap(h_matched[k], h_matched[m]);
            if (lens && !chained) { std::swap(h_und[k], h_und[m]); std::swap(h_und[n + k], h_und[n + m]); }
        }
    tracked.resize(m);
    last_matched = m;
    if (chained && *h_count != m) return fail(LVK_HIP_ERR_RUNTIME, "GPU-side fast_filter disagrees with the host's");
    if ((size_t)m < (size_t)s.min_motion_samples) { tracked.clear(); return LVK_HIP_OK; }

    trace.mark(HostTrace::FILTER);
    // ---- motion estimate
    motion = WarpMeshF(s.motion_height, s.motion_width);
    const float2* e1 = lens ? h_und : h_pts;
    const float2* e2 = lens ? h_und + n : h_matched;
    if (s.track_local_motions)
    {
        // estimate_local_motions (FrameTracker.cpp:200-321): least-squares mesh through the feature matches (more matches than the GPU-side
        // fast_filter handles: the pairs go up in one copy, as for the RANSAC below)
        std::memcpy(h_p1, e1, m * sizeof(float2));
        std::memcpy(h_p1 + m, e2, m * sizeof(float2));
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_p1, h_p1, 2 * (size_t)m * sizeof(float2), hipMemcpyHostToDevice, st));
        if ((rc = lvk_launch_mesh_solve(mesh_dev, st, d_mesh_scratch, d_p1, d_p1 + m, nullptr, m, 0, (float)cur_w, (float)cur_h,
                                        s.temporal_smoothing, s.acceptance_threshold, h_offsets, h_mask, h_mesh_status)) != LVK_HIP_OK) return rc;
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
        if (*h_mesh_status != 0)
        {
            tracked.clear();                                                          // like the other no-motion exits
            return LVK_HIP_OK;                                                        // no estimate this frame (identity motion)
        }
        std::memcpy(motion.off.data(), h_offsets, motion.off.size() * sizeof(float));
        size_t inl = 0;
        for (int i = 0; i < m; i++) inl += h_mask[i] ? 1 : 0;
        tracking_stability = (float)inl / (float)m;
        for (int i = m - 1; i >= 0; i--)
        {
            if (h_mask[i]) { tracked[i].age++; tracked[i].x = h_matched[i].x; tracked[i].y = h_matched[i].y; }
            else { std::swap(tracked[i], tracked.back()); tracked.pop_back(); }
        }
        grid.propagate(tracked);
        have_motion = true;
        return LVK_HIP_OK;
    }
    // both point sets travel in one copy (h_p1 and d_p1 each hold p1 | p2 in one allocation); results come back through
    // the pinned host block the kernel writes directly
    if (!chained)
    {
        std::memcpy(h_p1, e1, m * sizeof(float2));
        std::memcpy(h_p1 + m, e2, m * sizeof(float2));
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(d_p1, h_p1, 2 * (size_t)m * sizeof(float2), hipMemcpyHostToDevice, st));
        pe = prof_begin(LVK_STAGE_MOTION);
        if ((rc = lvk_launch_ransac(ctx, d_p1, d_p1 + m, m, s.acceptance_threshold, (double)cur_w, (double)cur_h, full, d_ransac_ws, h_H, h_ninl, h_mask)) != LVK_HIP_OK) return rc;
        prof_end(pe);
        trace.mark(HostTrace::RANSAC_LAUNCH);
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
        trace.mark(HostTrace::RANSAC_SYNC);
    }
    std::memcpy(last_H, h_H, sizeof(last_H));
    motion.from_homography(last_H, (float)cur_w, (float)cur_h);

    size_t inliers = 0;
    for (int i = 0; i < m; i++) inliers += h_mask[i] ? 1 : 0;
    tracking_stability = (float)inliers / (float)m;                                      // ratio_of(inlier_status, 1)

    for (int i = m - 1; i >= 0; i--)                                                     // FrameTracker.cpp:183-192
    {
        if (h_mask[i]) { tracked[i].age++; tracked[i].x = h_matched[i].x; tracked[i].y = h_matched[i].y; }
        else { std::swap(tracked[i], tracked.back()); tracked.pop_back(); }
    }
    grid.propagate(tracked);
    have_motion = true;
    trace.mark(HostTrace::POST);
    return LVK_HIP_OK;
}

// StabilizationFilter::filter (StabilizationFilter.cpp:69-135).  (luma, luma_step, luma_pix): where the tracker reads the
// luma of this frame from -- the packed frame itself (pix 3) or, on the YUV420 path, the caller's planar Y (pix 1).

int lvk_stab_push_impl(lvk_hip_stab* st, const void* d_frame, int step, int rows, int cols, uint64_t timestamp, int format,
                     const void* luma, int luma_step, int luma_pix,
                     void* d_out, int out_step, int out_rows, int* produced, uint64_t* out_timestamp, const void** released, OutPlanes420* o420,
                     lvk_frame_info* emitted)
{
    lvk_hip_ctx* ctx = st->ctx;
    if (produced) *produced = 0;
    if (released) *released = nullptr;
    LVK_HIP_REQUIRE(ctx, d_frame && rows > 0 && cols > 0 && step >= 3 * cols);           // !input.empty()
    if (!st->buffers_ok) return ctx->fail(LVK_HIP_ERR_RUNTIME, "the last configure() failed while allocating the tracker's buffers: configure again");
    // 3-channel VideoFrame formats (VideoFrame.cpp:170-306): YUV tracks channel 0, BGR / RGB track cvtColor(..2GRAY); the remap
    // runs the YUV or the RGB EASU program by the frame's format (Image.cpp:36-41).  GRAY / 4-channel frames are not on this path.
    LVK_HIP_REQUIRE(ctx, format == LVK_FORMAT_YUV || format == LVK_FORMAT_BGR || format == LVK_FORMAT_RGB);
    const int luma_channel = format == LVK_FORMAT_YUV ? 0 : (format == LVK_FORMAT_BGR ? -1 : -2);
    LVK_HIP_REQUIRE(ctx, luma_pix == 3 || format == LVK_FORMAT_YUV);
    const QueuedFrame in{d_frame, step, rows, cols, timestamp, format};
    // The frame this push will emit is the DELAYED one, at its own size (the queue holds whole frames, StabilizationFilter.cpp:118-131; dst is
    // allocated from the delayed source, WarpMesh.cpp:183-223 -> Image.cpp:53,116): what cannot be written is refused HERE, before the tracker
    // runs and before the queue moves -- a refused push leaves the filter as it was, and no borrowed frame is stranded.
    {
        QueuedFrame due;
        if (st->next_output(in, &due))
        {
            const bool to_planes = o420 && o420->y && (st->s.stabilize_output || st->s.crop_to_stable_region || st->lens);      // (the fused remap + egress kernel)
            // (4:2:0 entries: the planes take the frame whichever route it leaves by -- the fused kernel, or the packed buffer + the egress kernel)
            const bool planes_fit = !o420 || o420->vf != 0 ||
                                    (o420->y && o420->y_step >= due.cols && o420->u_step >= (o420->nv12 ? due.cols : due.cols / 2) &&
                                     (o420->nv12 || o420->v_step >= due.cols / 2) && o420->rows_cap >= due.rows);
            const bool fits = planes_fit && (to_planes || (d_out != nullptr && out_step >= 3 * due.cols && out_rows >= due.rows));
            if (!fits)
                return ctx->fail(LVK_HIP_ERR_ARG, "the output buffer does not hold the frame this push emits: " + std::to_string(due.cols) + " x " + std::to_string(due.rows) +
                                                      " (the DELAYED frame's own size -- lvk_hip_stab_next_output); nothing was queued");
        }
    }
    // (from here on the push happens: which schedule it takes is decided -- and counted -- now, not for a push that was refused above)
    st->caller_runs_free = st->caller_free_running_now() || st->host_free_running_hint;
    // One push of grace: a caller that has been running free for a while and synchronises ONCE (the end of a batch, a barrier in front of a timed
    // region) is most likely still a free-running caller -- its first push after the synchronisation keeps the free-running schedule (persistent
    // remap grid, so that the chain of the push behind it finds room; round 6: the two pushes after a device-wide synchronisation 0.16-0.29 + 0.13-0.15 ms
    // -> 0.10-0.21 + 0.10-0.12).  A second push in a row that looks synchronous switches the mode; a caller that waits for every frame never sees this.
    if (st->caller_runs_free) { st->free_streak++; st->sync_streak = 0; }
    else
    {
        const bool grace = st->free_streak >= 8 && st->sync_streak == 0;
        st->sync_streak++;
        if (grace) st->caller_runs_free = true; else st->free_streak = 0;
    }
    st->sched[st->caller_runs_free ? LVK_SCHED_PUSH_FREE_RUNNING : LVK_SCHED_PUSH_SYNCHRONISED]++;
    { const int lrc = st->ensure_lens(rows, cols); if (lrc != LVK_HIP_OK) return lrc; }
    static const WarpMeshF identity_mesh(2, 2);
    const uint8_t bg[3] = {(uint8_t)st->s.background[0], (uint8_t)st->s.background[1], (uint8_t)st->s.background[2]};

    auto enqueue = [&]() {
        if (st->queue.size() == st->queue_capacity) { if (released) *released = st->queue.front().d_ptr; st->queue.pop_front(); }
        st->queue.push_back(in);
    };
    auto emit = [&](const WarpMeshF* mesh) -> int {
        const QueuedFrame f = st->queue.front();
        st->queue.pop_front();
        int rc = LVK_HIP_OK;
        // overlap mode: the delayed frame was pushed >= 1 push ago and the tracker has synchronised the main stream since,
        // so the remap may run on its own stream concurrently with the next frame's tracking
        const bool side = st->overlap && mesh && st->s.stabilize_output;
        hipStream_t rs = side ? st->remap_stream : ctx->stream;
        // The persistent grid (4 remap blocks per CU) leaves room for the tracker's blocks of the NEXT frame; 5 or 6 starve them (8 450 /
        // 7 700 instead of 8 780 frames/s).  That only matters to a caller that runs free: one that waits for every frame (the previous
        // push ended long ago and the bulk stream is idle) gets the full grid -- the remap then has the GPU to itself (p50 latency -6 %).
        const bool persistent = side && st->caller_runs_free;
        if (mesh) st->sched[persistent ? LVK_SCHED_REMAP_PERSISTENT : LVK_SCHED_REMAP_FULL]++;
        // a remap whose stores cross the host link (lvk_hip_stab_push_yuv420_host) is bound by the link, not by the chip: ONE block per CU
        // for a free-running caller -- measured (two upload streams at the time) 2 800 frames/s against 2 560 with the 4 blocks per CU of a device-resident stream (the
        // stores of more blocks only fill the link's write queue sooner, which stalls the tracker's kernels), 2 450 with one per two CUs
        ctx->co_blocks_per_cu = (persistent && st->host_direct_now) ? 1 : 0;
        if (side && (rc = st->bulk_stream_sees_caller_work()) != LVK_HIP_OK) return rc;
        if (st->remap_wait) { const hipEvent_t e = st->remap_wait; st->remap_wait = nullptr; LVK_HIP_CHECK(ctx, hipStreamWaitEvent(rs, e, 0)); }
        st->trace.mark(HostTrace::EMIT_WAITS);
        const int pe = st->prof_begin(LVK_STAGE_REMAP, rs);
        // fused lens mode: the pre-warp of the DELAYED frame's own size (the tracker's model is that of the incoming frame)
        LensArgs lens_other; const LensArgs* lens_args = nullptr;
        if (st->lens)
        {
            lens_args = &st->lens_args;
            if (f.rows != st->lens_rows || f.cols != st->lens_cols)
            {
                LensModel m;
                if (lvk_lens_model_build(st->lens_params, f.rows, f.cols, m) != LVK_HIP_OK) return ctx->fail(LVK_HIP_ERR_ARG, "invalid camera profile for the delayed frame's size");
                std::memcpy(lens_other.f, m.f, sizeof(lens_other.f)); lens_args = &lens_other;
            }
        }
        if (mesh && o420 && o420->y && o420->vf != 0)
        {
            rc = lvk_launch_warpmesh_apply_obs(ctx, rs, o420->vf, f.d_ptr, f.step, f.rows, f.cols, o420->p, o420->s, mesh->off.data(), mesh->rows, mesh->cols, bg, lens_args, persistent);
            o420->used = true;
        }
        else if (mesh && o420 && o420->y)
        {
            rc = lvk_launch_warpmesh_apply_420(ctx, rs, f.d_ptr, f.step, f.rows, f.cols, o420->y, o420->y_step, o420->u, o420->u_step, o420->v, o420->v_step,
                                               o420->nv12, mesh->off.data(), mesh->rows, mesh->cols, bg, lens_args, persistent);
            o420->used = true;
        }
        else if (mesh) rc = lvk_launch_warpmesh_apply_lens(ctx, rs, f.d_ptr, f.step, f.rows, f.cols, d_out, out_step, mesh->off.data(), mesh->rows, mesh->cols, bg,
                                                      f.format == LVK_FORMAT_YUV ? 1 : 0, lens_args, persistent);
        else
        {
            hipError_t e = hipMemcpy2DAsync(d_out, out_step, f.d_ptr, f.step, (size_t)f.cols * 3, f.rows, hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) rc = ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
        }
        st->prof_end(pe, rs);
        st->trace.mark(HostTrace::EMIT_KERNEL);
        if (rc != LVK_HIP_OK) return rc;
        if (produced) *produced = 1;
        if (out_timestamp) *out_timestamp = f.ts;                                         // WarpMesh.cpp:221-222
        if (emitted) *emitted = lvk_frame_info{f.rows, f.cols, f.format};
        if (side && st->pool_frames)
        {
            // 4:2:0 path in overlap mode: the slot is next written by an ingest -- on this same stream, i.e. after the remap that is
            // reading it now (stream order is all the protection it needs), or on the tracking stream behind this event
            if (st->tracker_ingest_capable)
            {
                const int si = st->slot_index(f.d_ptr);
                if (si >= 0)
                {
                    if (!st->slot_read_done[(size_t)si]) { hipError_t e = hipEventCreateWithFlags(&st->slot_read_done[(size_t)si], hipEventDisableTiming); if (e != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); }
                    hipError_t e = hipEventRecord(st->slot_read_done[(size_t)si], rs);
                    if (e != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
                    st->slot_read_armed[(size_t)si] = 1;
                }
            }
            if (released) *released = f.d_ptr;
        }
        else if (side)
        {
            // the frame stays borrowed until its remap has finished: hand back the previous one instead
            const int slot = st->remap_slot; st->remap_slot ^= 1;
            hipError_t e = hipEventRecord(st->remap_done[slot], rs);
            if (e != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
            if (st->pending_release)
            {
                if ((e = hipEventSynchronize(st->remap_done[st->pending_slot])) != hipSuccess) return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e));
                if (released) *released = st->pending_release;
            }
            st->pending_release = f.d_ptr; st->pending_slot = slot;
        }
        else if (released) *released = f.d_ptr;
        return LVK_HIP_OK;
    };

    if (!st->s.stabilize_output)                                                          // StabilizationFilter.cpp:77-95
    {
        enqueue();
        if (st->queue.size() != st->queue_capacity) return LVK_HIP_OK;
        return emit(st->s.crop_to_stable_region ? &st->smoother.scene_crop() : st->lens ? &identity_mesh : nullptr);
    }

    st->trace.mark(HostTrace::ENTER);
    WarpMeshF motion(st->s.motion_height, st->s.motion_width);                            // m_NullMotion
    WarpMeshF est; bool have = false;
    if (st->post_error) { st->post_error = false; return st->fail(LVK_HIP_ERR_RUNTIME, "GPU-side fast_filter disagrees with the host's"); }
    int rc = st->track(in, luma, luma_step, luma_pix, luma_channel, est, have);
    if (rc != LVK_HIP_OK) { st->finish_post(); return rc; }
    if (have) motion = est;

    // quality assurance (StabilizationFilter.cpp:101-115)
    const float tq = st->tracking_stability;
    st->scene_quality = st->scene_quality + QA_UPDATE_RATE * (tq - st->scene_quality);
    if (tq < st->s.min_tracking_quality) st->trust = 0.0f;
    else if (st->scene_quality < st->s.min_scene_quality) st->trust = step_toward(st->trust, 0.0f, QA_BLEND_STEP);
    else st->trust = step_toward(st->trust, 1.0f, QA_BLEND_STEP);
    motion.scale(st->trust);
    st->last_motion = motion;

    enqueue();
    WarpMeshF correction = st->smoother.next(motion);
    if (st->queue.size() != st->queue_capacity) return LVK_HIP_OK;                          // !ready(): output.release()
    if (st->s.crop_to_stable_region) correction += st->smoother.scene_crop();
    st->last_correction = correction;
    st->trace.mark(HostTrace::SMOOTH);
    const int erc = emit(&correction);
    st->trace.mark(HostTrace::REMAP_LAUNCH);
    return erc;                                                                             // finish_post(): at the next push, or when the lists are read
}

// a free-running caller: the bulk stream still busy with the previous remap, or this push beginning within 15 us of the previous one's return
// (a caller that waits for its frames synchronises and reads back in between: at least a remap's duration)
bool lvk_hip_stab::caller_free_running_now()
{
    bulk_busy_at_push = false;
    if (overlap && remap_stream)
    {
        const hipError_t q = hipStreamQuery(remap_stream);
        if (q != hipSuccess) (void)hipGetLastError();
        bulk_busy_at_push = q == hipErrorNotReady;
    }
    if (assume_caller) return assume_caller == 1;
    return bulk_busy_at_push || (last_push_end.time_since_epoch().count() != 0 && std::chrono::steady_clock::now() - last_push_end < std::chrono::microseconds(15));
}

int lvk_hip_stab::ensure_pool(int rows, int cols)
{
    const size_t want = (size_t)s.predictive_samples + 4;
    if (rows == pool_rows && cols == pool_cols && pool_all.size() >= want) return LVK_HIP_OK;
    if (rows != pool_rows || cols != pool_cols)
    {
        LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (remap_stream) LVK_HIP_CHECK(ctx, hipStreamSynchronize(remap_stream));
        // A resolution change in the middle of a 4:2:0 stream.  The frames still queued live in pool slots of the old geometry; like the reference,
        // whose queue holds whole VideoFrames (StabilizationFilter.cpp:118-131), they STAY queued and leave at their own size over the next
        // `frame_delay` pushes (rounds 2-5 dropped them): their slots are retired -- kept until their frame has been emitted -- and the rest of the
        // old pool is freed.  Tracker and path smoother carry on.
        std::vector<void*> keep;
        if (queue_kind == 2) for (const QueuedFrame& q : queue) keep.push_back(const_cast<void*>(q.d_ptr));
        for (void* p : pool_all)
        {
            if (std::find(keep.begin(), keep.end(), p) != keep.end()) pool_retired.push_back(p);
            else (void)hipFree(p);
        }
        for (hipEvent_t e : slot_read_done) if (e) (void)hipEventDestroy(e);
        pool_all.clear(); pool_free.clear(); slot_read_done.clear(); slot_read_armed.clear();
        pending_release = nullptr; pending_slot = -1;
        pool_rows = rows; pool_cols = cols;
        // the packed output of the un-fused egress route holds the largest frame that can still be emitted
        size_t out_bytes = (size_t)rows * cols * 3;
        if (queue_kind == 2) for (const QueuedFrame& q : queue) out_bytes = std::max(out_bytes, (size_t)q.rows * q.cols * 3);
        if (out_bytes > pool_out_bytes)
        {
            if (pool_out) { (void)hipFree(pool_out); pool_out = nullptr; pool_out_bytes = 0; }
            LVK_HIP_CHECK(ctx, hipMalloc(&pool_out, out_bytes));
            pool_out_bytes = out_bytes;
        }
    }
    while (pool_all.size() < want)
    {
        void* p = nullptr;
        LVK_HIP_CHECK(ctx, hipMalloc(&p, (size_t)rows * cols * 3));
        pool_all.push_back(p); pool_free.push_back(p); slot_read_done.push_back(nullptr); slot_read_armed.push_back(0);
    }
    return LVK_HIP_OK;
}

int lvk_hip_stab::release_retired(const void* p)
{
    auto it = std::find(pool_retired.begin(), pool_retired.end(), p);
    if (it == pool_retired.end()) return LVK_HIP_OK;
    // its remap was enqueued a moment ago (bulk or tracking stream): wait for it -- a handful of times per resize, never in steady state
    LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (remap_stream) LVK_HIP_CHECK(ctx, hipStreamSynchronize(remap_stream));
    (void)hipFree(*it);
    pool_retired.erase(it);
    return LVK_HIP_OK;
}

int lvk_hip_stab::sweep_retired()
{
    for (size_t i = 0; i < pool_retired.size();)
    {
        bool queued = false;
        for (const QueuedFrame& q : queue) queued = queued || q.d_ptr == pool_retired[i];
        if (queued) { i++; continue; }
        const int rc = release_retired(pool_retired[i]);
        if (rc != LVK_HIP_OK) return rc;
    }
    return LVK_HIP_OK;
}

void lvk_hip_stab::free_pool()
{
    for (void* p : pool_all) (void)hipFree(p);
    for (void* p : pool_retired) (void)hipFree(p);
    for (hipEvent_t e : slot_read_done) if (e) (void)hipEventDestroy(e);
    pool_all.clear(); pool_free.clear(); pool_retired.clear(); slot_read_done.clear(); slot_read_armed.clear();
    if (pool_out) { (void)hipFree(pool_out); pool_out = nullptr; }
    pool_out_bytes = 0;
    pool_rows = pool_cols = 0;
}

extern "C" {

int lvk_hip_stab_next_output(const lvk_hip_stab* st, int rows, int cols, int format, lvk_frame_info* out)
{
    if (!st) return LVK_HIP_ERR_ARG;
    QueuedFrame due{};
    if (!st->next_output(QueuedFrame{nullptr, 3 * cols, rows, cols, 0, format}, &due)) return 0;
    if (out) *out = lvk_frame_info{due.rows, due.cols, due.format};
    return 1;
}

int lvk_hip_stab_push(lvk_hip_stab* st, const void* d_frame, int step, int rows, int cols, uint64_t timestamp, int format,
                      void* d_out, int out_step, int out_rows, int* produced, uint64_t* out_timestamp, const void** released, lvk_frame_info* emitted)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    st->trace.begin();
    st->prof_tick++;
    st->push_seq++;
    struct AnnouncementEnds { lvk_hip_stab* s; ~AnnouncementEnds() { s->ahead_announced = lvk_hip_stab::LumaAhead(); } } announcement_ends{st};
    if (st->queue.empty()) st->queue_kind = 0;
    if (st->queue_kind == 2) return st->fail(LVK_HIP_ERR_ARG, "frames of lvk_hip_stab_push_yuv420 are still queued: restart() before switching to lvk_hip_stab_push");
    st->queue_kind = 1;
    st->pool_frames = false;
    int rc = st->mark_caller_work();
    if (rc != LVK_HIP_OK) return rc;
    rc = lvk_stab_push_impl(st, d_frame, step, rows, cols, timestamp, format, d_frame, step, 3, d_out, out_step, out_rows, produced, out_timestamp, released, nullptr, emitted);
    if (released && !*released && !st->orphaned.empty()) { *released = st->orphaned.front(); st->orphaned.pop_front(); }
    st->trace.mark(HostTrace::EXIT);
    st->last_push_end = std::chrono::steady_clock::now();
    return rc;
}

// The OBS asynchronous path in one call: I4XXIngest / NV12Ingest::to_ocl -> StabilizationFilter::filter -> ::to_obs
// (Modules/OBS-Plugin/Interop/VisionFilter.cpp:151-212, FrameIngest.cpp:494-602).  Planar (or NV12) 4:2:0 in, 4:2:0 out;
// the packed 8UC3 frames the filter works on live in an internal pool (predictive_samples + 4 frames).  The input planes
// are consumed before the call returns; the output planes are complete after lvk_hip_sync().
// (shared by lvk_hip_stab_push_yuv420 and lvk_hip_stab_push_obs: `vf` = the OBS video format of the planes; the 4:2:0 formats leave through the fused
//  remap + egress kernel, every other format through the packed buffer and its egress kernel)
static int lvk_stab_push_planes(lvk_hip_stab* st, int vf, const void* const in_planes[3], const int in_steps[3], int rows, int cols, uint64_t timestamp,
                                void* const out_planes[3], const int out_steps[3], int o_rows,
                                int* produced, uint64_t* out_timestamp, lvk_frame_info* emitted)
{
    if (!st) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(st->ctx);
    lvk_hip_ctx* ctx = st->ctx;
    const bool is420 = vf == LVK_VIDEO_FORMAT_I420 || vf == LVK_VIDEO_FORMAT_I40A || vf == LVK_VIDEO_FORMAT_NV12;
    const int nv12 = vf == LVK_VIDEO_FORMAT_NV12 ? 1 : 0;
    const int frame_format = lvk_hip_obs_frame_format(vf);
    if (frame_format < 0 || frame_format == LVK_FORMAT_GRAY || !in_planes || !in_steps || !in_planes[0])
        return st->fail(LVK_HIP_ERR_ARG, "lvk_hip_stab_push_obs: video format " + std::to_string(vf) + " has no three-channel frame the filter could take (FrameIngest::Select, lvk::remap: CV_8UC3)");
    const std::array<const void*, 3> ip{in_planes[0], in_planes[1], in_planes[2]};
    const std::array<int, 3> is_{in_steps[0], in_steps[1], in_steps[2]};
    std::array<void*, 3> op{nullptr, nullptr, nullptr}; std::array<int, 3> os{0, 0, 0};
    if (out_planes && out_steps) { op = {out_planes[0], out_planes[1], out_planes[2]}; os = {out_steps[0], out_steps[1], out_steps[2]}; }
    // (the names of the 4:2:0 route)
    void* o_y = op[0]; void* o_u = op[1]; void* o_v = nv12 ? op[1] : op[2];
    const int oy_step = os[0], ou_step = os[1], ov_step = nv12 ? os[1] : os[2];
    // where the tracker reads its luma (VideoFrame::viewAsFormat(GRAY), VideoFrame.cpp:260): the caller's Y plane / the Y bytes of the packed formats; the
    // BGR / RGB frames of DirectIngest are tracked from the converted copy (cvtColor needs all three channels)
    const bool direct = frame_format != LVK_FORMAT_YUV;
    const uint8_t* luma = static_cast<const uint8_t*>(ip[0]); int luma_step = is_[0], luma_pix = 1;
    if (vf == LVK_VIDEO_FORMAT_YUY2 || vf == LVK_VIDEO_FORMAT_YVYU) luma_pix = 2;
    else if (vf == LVK_VIDEO_FORMAT_UYVY) { luma += 1; luma_pix = 2; }
    else if (vf == LVK_VIDEO_FORMAT_AYUV) { luma += 1; luma_pix = 4; }
    st->trace.begin();
    st->prof_tick++;
    st->push_seq++;
    struct AnnouncementEnds { lvk_hip_stab* s; ~AnnouncementEnds() { s->ahead_announced = lvk_hip_stab::LumaAhead(); } } announcement_ends{st};
    if (produced) *produced = 0;
    if (st->queue.empty()) st->queue_kind = 0;
    if (st->queue_kind == 1) return st->fail(LVK_HIP_ERR_ARG, "borrowed frames of lvk_hip_stab_push are still queued: restart() before switching to lvk_hip_stab_push_yuv420");
    if (!is420)
    {
        // the planes of the frame this push emits (the DELAYED one, at its own size) must hold it: refused before anything changes, like the 4:2:0 planes
        QueuedFrame due{};
        if (st->next_output(QueuedFrame{nullptr, 3 * cols, rows, cols, timestamp, frame_format}, &due))
        {
            const int packed = (vf == LVK_VIDEO_FORMAT_YUY2 || vf == LVK_VIDEO_FORMAT_YVYU || vf == LVK_VIDEO_FORMAT_UYVY) ? 2 :
                               (vf == LVK_VIDEO_FORMAT_AYUV || vf == LVK_VIDEO_FORMAT_RGBA || vf == LVK_VIDEO_FORMAT_BGRA || vf == LVK_VIDEO_FORMAT_BGRX) ? 4 :
                               vf == LVK_VIDEO_FORMAT_BGR3 ? 3 : 0;
            const int cw = (vf == LVK_VIDEO_FORMAT_I422 || vf == LVK_VIDEO_FORMAT_I42A) ? due.cols / 2 : due.cols;
            const bool fits = op[0] && o_rows >= due.rows &&
                              (packed ? os[0] >= packed * due.cols : (op[1] && op[2] && os[0] >= due.cols && os[1] >= cw && os[2] >= cw));
            if (!fits)
                return st->fail(LVK_HIP_ERR_ARG, "the output planes do not hold the frame this push emits: " + std::to_string(due.cols) + " x " + std::to_string(due.rows) +
                                                     " (the DELAYED frame's own size -- lvk_hip_stab_next_output); nothing was queued");
        }
    }
    st->queue_kind = 2;
    int rc = st->ensure_pool(rows, cols);
    if (rc != LVK_HIP_OK) return rc;
    if (!st->pool_retired.empty() && (rc = st->sweep_retired()) != LVK_HIP_OK) return rc;
    if ((rc = st->mark_caller_work()) != LVK_HIP_OK) return rc;
    if (st->pool_free.empty())
    {
        // frames dropped by restart() / a shrinking queue never came back through *released: reclaim them
        for (void* p : st->pool_all)
        {
            bool used = (p == st->pending_release);
            for (const QueuedFrame& q : st->queue) used = used || (q.d_ptr == p);
            if (!used) st->pool_free.push_back(p);
        }
        LVK_HIP_REQUIRE(ctx, !st->pool_free.empty());
    }
    void* slot = st->pool_free.front(); st->pool_free.pop_front();
    // The packed frame is only read by the remap `predictive_samples` pushes later (the tracker reads the luma plane itself), so in
    // overlap mode the conversion runs on the remap stream, off the tracker's critical path; same-stream order protects the slot.
    const bool side_ingest = st->overlap && st->s.stabilize_output && !direct;
    // a delayed frame (its remap is launched after later synchronisations of the tracking stream) may be converted on either stream: track() decides
    st->tracker_ingest_capable = side_ingest && st->queue_capacity > 1;
    st->ingest_on_tracker = false;
    int pe = 0;
    auto do_ingest = [=]() -> int {
        const bool on_tracker = side_ingest && st->ingest_on_tracker;
        st->sched[on_tracker ? LVK_SCHED_INGEST_ON_TRACKER : (side_ingest ? LVK_SCHED_INGEST_ON_BULK : LVK_SCHED_INGEST_INLINE)]++;
        hipStream_t is = (side_ingest && !on_tracker) ? st->remap_stream : ctx->stream;
        if (side_ingest && !on_tracker) { const int w = st->bulk_stream_sees_caller_work(); if (w != LVK_HIP_OK) return w; }
        if (on_tracker)
        {
            const int si = st->slot_index(slot);
            if (si >= 0 && st->slot_read_armed[(size_t)si]) { st->slot_read_armed[(size_t)si] = 0; LVK_HIP_CHECK(ctx, hipStreamWaitEvent(is, st->slot_read_done[(size_t)si], 0)); }
        }
        // host-resident frames: the planes are still arriving on the upload stream
        for (hipEvent_t& e : st->ingest_wait) if (e) { LVK_HIP_CHECK(ctx, hipStreamWaitEvent(is, e, 0)); e = nullptr; }
        const int pi = st->prof_begin(LVK_STAGE_INGEST, is);
        const int r = lvk_launch_ingest_obs(ctx, is, vf, ip.data(), is_.data(), rows, cols, slot, 3 * cols);
        st->prof_end(pi, is);
        if (r != LVK_HIP_OK) return r;
        if (side_ingest)
        {
            if (!st->ingest_done) LVK_HIP_CHECK(ctx, hipEventCreateWithFlags(&st->ingest_done, hipEventDisableTiming));
            LVK_HIP_CHECK(ctx, hipEventRecord(st->ingest_done, is));
            st->ingest_recorded_for = st->push_seq;
        }
        return LVK_HIP_OK;
    };
    // Overlap mode: the conversion is not on the tracker's critical path (it runs behind the previous remap on the bulk stream), so its
    // launch and its event record wait until the tracker's kernels are on their way -- track() calls it after its last launch.
    // (8.05k -> 8.17k frames/s, p50 latency -7 us.)
    if (side_ingest) st->deferred_ingest = do_ingest;
    else { rc = do_ingest(); if (rc != LVK_HIP_OK) { st->pool_free.push_back(slot); return rc; } }
    int prod = 0; const void* released = nullptr;
    st->pool_frames = side_ingest;
    OutPlanes420 o420{o_y, oy_step, o_u, ou_step, o_v, ov_step, nv12, false, o_rows};
    if (!(o_y && o_u && (nv12 || o_v))) o420.y = nullptr;
    const bool fused_obs = !is420 && lvk_remap_obs_fusable(vf) && op[0];
    if (fused_obs) { o420.vf = vf; o420.y = op[0]; for (int i = 0; i < 3; i++) { o420.p[i] = op[i]; o420.s[i] = os[i]; } }
    // (the packed route's buffer: pool_out, tight rows, as many as its allocation holds at the widest queued frame)
    lvk_frame_info info{0, 0, 0};
    {
        QueuedFrame due{};
        const bool will = st->next_output(QueuedFrame{slot, 3 * cols, rows, cols, timestamp, frame_format}, &due);
        const int out_cols = will ? due.cols : cols;
        rc = lvk_stab_push_impl(st, slot, 3 * cols, rows, cols, timestamp, frame_format, direct ? slot : (const void*)luma, direct ? 3 * cols : luma_step, direct ? 3 : luma_pix,
                                st->pool_out, 3 * out_cols, (int)(st->pool_out_bytes / ((size_t)3 * out_cols)), &prod, out_timestamp, &released,
                                (is420 || fused_obs) ? &o420 : nullptr, &info);
    }
    {
        // a push that was refused (or failed before the frame was queued) has not taken the slot: its conversion is not launched, the slot is free again
        bool taken = released == slot;
        for (const QueuedFrame& q : st->queue) taken = taken || q.d_ptr == slot;
        if (!taken) { st->deferred_ingest = nullptr; st->pool_free.push_front(slot); }
    }
    if (st->deferred_ingest) { const int r2 = st->run_deferred_ingest(); if (rc == LVK_HIP_OK) rc = r2; }      // (track() returned before its launches)
    if (released)
    {
        if (st->is_retired(released)) { const int r3 = st->release_retired(released); if (rc == LVK_HIP_OK) rc = r3; }      // a frame of an earlier size has left
        else st->pool_free.push_back(const_cast<void*>(released));
    }
    if (emitted && prod) *emitted = info;
    if (side_ingest)
    {
        // (the next frame's pyramid is on its way behind this chain: nothing will shadow the list bookkeeping at the start of the next push --
        //  track() does it behind the two launches that are no longer there --, so it runs here, while the conversion is still running anyway)
        if (st->ahead_built_for == st->push_seq + 1) st->finish_post();
        st->trace.mark(HostTrace::EXIT_PRE);
        // contract: the caller's planes are consumed when the call returns (the conversion started ~a tracking pass ago).  An event, not
        // hipStreamSynchronize: synchronising the bulk stream itself costs ~10 us of host time even when it is idle (measured).
        // (only a conversion that THIS push launched: a push that was refused -- or failed -- before its frame was queued has launched none, and when it is
        //  the first overlap-mode push of the filter's life the event does not exist yet: the wait failed with "invalid resource handle", hid the refusal's own
        //  message and left a sticky runtime error for the next launch to report -- found by fuzz seeds 389 / 390 of the round-6 sweep)
        if (st->ingest_recorded_for == st->push_seq) LVK_HIP_CHECK(ctx, hipEventSynchronize(st->ingest_done));
        st->trace.mark(HostTrace::EXIT_WAIT);
    }
    // same contract without a tracker pass (delay-only mode, stabilize_output off): nothing has synchronised behind the conversion yet
    // (and for DirectIngest's formats, whose copy runs on the tracking stream and whose tracker reads the copy: an early return of track() has not waited)
    else if (!st->s.stabilize_output || direct) LVK_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (rc != LVK_HIP_OK) return rc;
    if (prod && o420.used) { if (produced) *produced = 1; }                // the fused remap + egress kernel has written the planes
    else if (prod)
    {
        // the emitted frame has ITS OWN geometry (a frame queued before the size changed leaves at the old size)
        if (is420)
        {
            LVK_HIP_REQUIRE(ctx, o_y && o_u && (nv12 || o_v));
            LVK_HIP_REQUIRE(ctx, oy_step >= info.cols && ou_step >= (nv12 ? info.cols : info.cols / 2) && (nv12 || ov_step >= info.cols / 2) && o_rows >= info.rows);
        }
        else LVK_HIP_REQUIRE(ctx, op[0] && o_rows >= info.rows);
        hipStream_t es = (st->overlap && st->s.stabilize_output) ? st->remap_stream : ctx->stream;
        if (st->remap_wait) { const hipEvent_t e = st->remap_wait; st->remap_wait = nullptr; LVK_HIP_CHECK(ctx, hipStreamWaitEvent(es, e, 0)); }
        pe = st->prof_begin(LVK_STAGE_EGRESS, es);
        rc = lvk_launch_egress_obs(ctx, es, vf, st->pool_out, 3 * info.cols, info.rows, info.cols, op.data(), os.data());
        st->prof_end(pe, es);
        if (rc != LVK_HIP_OK) return rc;
        if (produced) *produced = 1;
    }
    st->trace.mark(HostTrace::EXIT);
    st->last_push_end = std::chrono::steady_clock::now();
    return LVK_HIP_OK;
}

int lvk_hip_stab_push_yuv420(lvk_hip_stab* st, const void* d_y, int y_step, const void* d_u, int u_step, const void* d_v, int v_step, int nv12,
                             int rows, int cols, uint64_t timestamp,
                             void* o_y, int oy_step, void* o_u, int ou_step, void* o_v, int ov_step, int o_rows,
                             int* produced, uint64_t* out_timestamp, lvk_frame_info* emitted)
{
    const void* in_planes[3] = {d_y, d_u, nv12 ? nullptr : d_v}; const int in_steps[3] = {y_step, u_step, nv12 ? 0 : v_step};
    void* out_planes[3] = {o_y, o_u, nv12 ? nullptr : o_v}; const int out_steps[3] = {oy_step, ou_step, nv12 ? 0 : ov_step};
    return lvk_stab_push_planes(st, nv12 ? LVK_VIDEO_FORMAT_NV12 : LVK_VIDEO_FORMAT_I420, in_planes, in_steps, rows, cols, timestamp,
                                out_planes, out_steps, o_rows, produced, out_timestamp, emitted);
}

// The plugin's asynchronous path for ANY format FrameIngest::Select accepts (except Y800), one call per frame: to_ocl -> StabilizationFilter::filter
// -> to_obs (Modules/OBS-Plugin/Interop/VisionFilter.cpp:151-212, FrameIngest.cpp:36-75).
int lvk_hip_stab_push_obs(lvk_hip_stab* st, int video_format, const void* const d_planes[3], const int steps[3], int rows, int cols, uint64_t timestamp,
                          void* const o_planes[3], const int o_steps[3], int o_rows, int* produced, uint64_t* out_timestamp, lvk_frame_info* emitted)
{
    return lvk_stab_push_planes(st, video_format, d_planes, steps, rows, cols, timestamp, o_planes, o_steps, o_rows, produced, out_timestamp, emitted);
}

} // extern "C"
