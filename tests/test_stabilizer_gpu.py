"""End-to-end GPU parity: the HIP stabilization filter vs the CPU oracle on the same synthetic clips, frame by frame,
through the C-ABI.  Bar: every emitted frame bit-identical, every per-frame statistic / mesh identical."""
import numpy as np
import pytest

from tests import oracle_lib, synth

pytestmark = pytest.mark.gpu


def _to_settings(o):
    """oracle_lib.StabSettings -> livevisionkit_amd.StabilizationFilterSettings (same field order)."""
    import ctypes
    import livevisionkit_amd as lvk
    s = lvk.StabilizationFilterSettings()
    assert ctypes.sizeof(s) == ctypes.sizeof(o)
    ctypes.memmove(ctypes.byref(s), ctypes.byref(o), ctypes.sizeof(o))
    return s


def _apply(H, p):
    q = np.c_[p, np.ones(len(p))] @ H.T
    return q[:, :2] / q[:, 2:]


@pytest.mark.parametrize("partial", [False, True])
def test_global_motion_bit_exact(ctx, oracle, partial):
    rng = np.random.default_rng(5)
    for trial, n in enumerate([4, 7, 75, 300, 1856, 3000]):
        H = np.array([[1.01, 0.012, 3.1], [-0.011, 0.995, -2.2], [2e-5 * (not partial), -1e-5 * (not partial), 1.0]])
        p1 = np.c_[rng.uniform(0, 480, n), rng.uniform(0, 270, n)].astype(np.float32)
        p2 = _apply(H, p1) + rng.normal(0, 0.2, p1.shape)
        out = rng.random(n) < 0.3
        p2[out] += rng.uniform(-40, 40, (out.sum(), 2))
        p2 = p2.astype(np.float32)
        rc_o, H_o, m_o = oracle.find_homography(p1, p2, 3.0, partial=partial)
        rc_g, H_g, m_g = ctx.estimate_global_motion(p1, p2, 3.0, full_homography=not partial)
        assert (rc_o < 0) == (rc_g < 0), (n, rc_o, rc_g)
        if rc_o >= 0:
            assert rc_o == rc_g, (n, rc_o, rc_g)
        assert np.array_equal(m_o, m_g), n
        assert np.array_equal(H_o.view(np.uint64), H_g.view(np.uint64)), (n, np.abs(H_o - H_g).max())


def test_global_motion_degenerate_inputs(ctx, oracle):
    same = np.tile(np.array([[10.0, 20.0]], np.float32), (50, 1))                  # all points identical -> no model
    line = np.c_[np.arange(50), 2 * np.arange(50)].astype(np.float32)              # collinear -> singular 4-point systems
    for p in (same, line):
        for partial in (False, True):
            rc_o, H_o, m_o = oracle.find_homography(p, p, 3.0, partial=partial)
            rc_g, H_g, m_g = ctx.estimate_global_motion(p, p, 3.0, full_homography=not partial)
            assert (rc_o < 0) == (rc_g < 0) and np.array_equal(m_o, m_g)
            assert np.array_equal(H_o, H_g)
    rc_g, H_g, m_g = ctx.estimate_global_motion(same[:3], same[:3], 3.0)
    assert rc_g < 0 and np.array_equal(H_g, np.eye(3))


def _run_pair(oracle, ctx, frames, settings, n_check_frames=None, reconfigure_at=None, then_configure=None, overlap=False):
    import torch
    import livevisionkit_amd as lvk
    ost = oracle_lib.OracleStabilizer(oracle, settings)
    gst = lvk.StabilizationFilter(_to_settings(settings), context=ctx)
    if overlap:
        gst.set_overlap(True)
    if then_configure is not None:                      # OBS flow: default-constructed filter, then reconfigure(preset)
        ost.configure(then_configure); gst.configure(_to_settings(then_configure))
    produced = 0
    for i, f in enumerate(frames):
        if reconfigure_at and i in reconfigure_at:
            new = reconfigure_at[i]
            ost.configure(new); gst.configure(_to_settings(new))
        want, wts = ost.push(f, ts=100 + i)
        got, gts = gst.apply(torch.from_numpy(f).cuda(), timestamp=100 + i)
        ctx.sync()
        so, sg = ost.stats(), gst.stats()
        for k in ("n_detected", "n_matched", "n_tracked"):
            assert getattr(so, k) == getattr(sg, k), (i, k, getattr(so, k), getattr(sg, k))
        for k in ("tracking_stability", "scene_quality", "trust", "distribution", "smoothing_factor"):
            assert getattr(so, k) == getattr(sg, k), (i, k, getattr(so, k), getattr(sg, k))
        assert list(so.homography) == list(sg.homography), (i, list(so.homography), list(sg.homography))
        mo, co = ost.meshes(); mg, cg = gst.meshes()
        assert np.array_equal(mo.view(np.uint32), mg.view(np.uint32)), i
        fo, fg = ost.features(), gst.features()
        assert np.array_equal(fo.view(np.uint32), fg.view(np.uint32)), i
        assert (want is None) == (got is None), i
        if want is not None:
            assert wts == gts
            assert np.array_equal(co.view(np.uint32), cg.view(np.uint32)), i
            g = got.cpu().numpy()
            if not np.array_equal(g, want):
                d = np.abs(g.astype(int) - want.astype(int))
                raise AssertionError(f"frame {i}: {np.count_nonzero(d.max(axis=2))} pixels differ, max {d.max()}")
            produced += 1
    ost.close(); gst.close()
    return produced


@pytest.fixture(scope="module")
def clip():
    return synth.make_clip(360, 640, 30, seed=11, jitter=1.0)


def test_stabilizer_homography_preset_bit_exact(ctx, oracle, clip):
    frames, _ = clip
    s = oracle_lib.preset("homography", predictive_samples=4)
    assert _run_pair(oracle, ctx, frames, s) == len(frames) - 4


def test_stabilizer_overlap_mode_bit_exact(ctx, oracle, clip):
    """Remap on the second stream (overlapping the next frame's tracking): same frames, same order, same bits."""
    frames, _ = clip
    s = oracle_lib.preset("homography", predictive_samples=4)
    assert _run_pair(oracle, ctx, frames, s, overlap=True) == len(frames) - 4
    field = oracle_lib.preset("field", predictive_samples=3, min_scene_quality=0.4, min_tracking_quality=0.2)
    assert _run_pair(oracle, ctx, frames[:20], oracle_lib.preset("default"), then_configure=field, overlap=True) == 17


def test_overlap_mode_without_per_frame_sync(ctx, oracle, clip):
    """Free-running overlap mode (no sync between pushes, outputs in distinct buffers, borrowed inputs reused only after
    release): every output still equals the oracle's."""
    import torch
    import livevisionkit_amd as lvk
    frames, _ = clip
    s = oracle_lib.preset("homography", predictive_samples=3)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    gst.set_overlap(True)
    wants, gots = [], []
    dev = [torch.from_numpy(f).cuda() for f in frames]
    for i, f in enumerate(frames):
        want, _ = ost.push(f, ts=i)
        got, _ = gst.apply(dev[i], timestamp=i)
        if want is not None:
            wants.append(want); gots.append(got)
    ctx.sync()
    assert len(gots) == len(frames) - 3
    for i, (w, g) in enumerate(zip(wants, gots)):
        assert np.array_equal(g.cpu().numpy(), w), i
    ost.close(); gst.close()


def test_stabilizer_relaxed_qa_no_crop_bit_exact(ctx, oracle, clip):
    frames, _ = clip
    s = oracle_lib.preset("homography", predictive_samples=3, min_scene_quality=0.4, min_tracking_quality=0.2,
                          crop_to_stable_region=0, corrective_limit_x=0.1, corrective_limit_y=0.08)
    assert _run_pair(oracle, ctx, frames[:20], s) == 17


def test_stabilizer_library_defaults_global_motion(ctx, oracle, clip):
    """Library default geometry (256x256 tracking, 2x2 regions, density 0.2) with the global-motion estimator."""
    frames, _ = clip
    s = oracle_lib.preset("default", track_local_motions=0, predictive_samples=3)
    assert _run_pair(oracle, ctx, frames[:14], s) == 11


def test_stabilizer_field_preset_obs_flow_bit_exact(ctx, oracle, clip):
    """Vector-field preset the way the OBS plugin reaches it: default-constructed filter, then reconfigure (VSFilter.cpp:235-294).
    16x16 mesh -> least-squares local motion (row a10) + the in-kernel mesh remap."""
    frames, _ = clip
    field = oracle_lib.preset("field", predictive_samples=4, min_scene_quality=0.4, min_tracking_quality=0.2)
    assert _run_pair(oracle, ctx, frames[:26], oracle_lib.preset("default"), then_configure=field) == 22


def test_stabilizer_field_preset_direct_construction_quirk(ctx, oracle, clip):
    """Constructed directly with a 16x16 mesh the reference keeps the constraints of FrameTracker's default 256x256 region
    (FrameTracker.cpp:74-82); the HIP path reproduces that too."""
    frames, _ = clip
    assert _run_pair(oracle, ctx, frames[:10], oracle_lib.preset("field", predictive_samples=3)) == 7


@pytest.mark.parametrize("mesh", [(17, 17), (32, 32)])
def test_stabilizer_field_preset_large_motion_resolution(ctx, oracle, clip, mesh):
    """motion_resolution beyond the register-window mesh solver (Math/WarpMesh.cpp:34-41,79-90 and FrameTracker.cpp:57-92 take any size):
    the generic device kernels, same specification -- whole filter bit-exact against the oracle, overlap mode included."""
    frames, _ = clip
    field = oracle_lib.preset("field", predictive_samples=3, min_scene_quality=0.4, min_tracking_quality=0.2, motion_width=mesh[0], motion_height=mesh[1])
    assert _run_pair(oracle, ctx, frames[:12], oracle_lib.preset("default"), then_configure=field, overlap=mesh == (32, 32)) == 9


@pytest.mark.parametrize("size", [(180, 320), (300, 400)])
def test_frames_below_the_detection_resolution_bit_exact(ctx, oracle, size):
    """FrameTracker.cpp:117 resizes whatever it is given to detection_resolution: a frame SMALLER than 480 x 270 on one or both axes goes through
    cv::resize's bilinear emulation of INTER_AREA (k_area_enlarge); the stream matches the oracle frame by frame, packed and 4:2:0."""
    import torch
    import livevisionkit_amd as lvk
    rows, cols = size
    frames, _ = synth.make_clip(rows, cols, 12, seed=rows, jitter=0.6)
    s = oracle_lib.preset("homography", predictive_samples=2, min_scene_quality=0.3, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    ost2 = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    gst2 = lvk.StabilizationFilter(_to_settings(s), context=ctx); gst2.set_overlap(True)
    emitted = tracked = 0
    for i, f in enumerate(frames):
        want, _ = ost.push(f, ts=i)
        got, _ = gst.apply(torch.from_numpy(f).cuda(), timestamp=i)
        ctx.sync()
        assert (want is None) == (got is None), i
        assert np.array_equal(gst.features(), ost.features()), i
        tracked = max(tracked, len(ost.features()))
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want), i
            emitted += 1
        planes = oracle.egress_yuv420(f)
        want2, _ = ost2.push(oracle.ingest_yuv420(*planes), ts=i)
        got2, _ = gst2.apply_yuv420(tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in planes), timestamp=i)
        ctx.sync()
        if want2 is not None:
            for a, b in zip(got2, oracle.egress_yuv420(want2)):
                assert np.array_equal(a.cpu().numpy(), b), i
    assert emitted == 10 and tracked > 80
    for x in (ost, ost2, gst, gst2):
        x.close()


def test_refused_configure_leaves_the_filter_untouched(ctx, oracle, clip):
    """A configure() the library refuses (here: a motion mesh wider than the device solver's 167 columns; also an out-of-range quality)
    returns an error and changes NOTHING: the following pushes match an oracle that never saw the call (round-2 ADVICE: the state used
    to be committed before the solver could fail)."""
    import torch
    import livevisionkit_amd as lvk
    frames, _ = clip
    s = oracle_lib.preset("field", predictive_samples=3, min_scene_quality=0.4, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    gst = lvk.StabilizationFilter(_to_settings(oracle_lib.preset("default")), context=ctx); gst.configure(_to_settings(s))
    gst.set_overlap(True)
    for i, f in enumerate(frames[:14]):
        if i == 6:
            for bad in (oracle_lib.preset("field", motion_width=200, motion_height=40, detection_width=320, detection_height=180, predictive_samples=7),
                        oracle_lib.preset("field", min_scene_quality=1.5, predictive_samples=2)):
                with pytest.raises(Exception):
                    gst.configure(_to_settings(bad))
        want, _ = ost.push(f, ts=i)
        got, _ = gst.apply(torch.from_numpy(f).cuda(), timestamp=i)
        ctx.sync()
        assert (want is None) == (got is None), i
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want), i
        mo, _ = ost.meshes(); mg, _ = gst.meshes()
        assert np.array_equal(mo.view(np.uint32), mg.view(np.uint32)), i
    assert gst.frame_delay() == 3
    ost.close(); gst.close()


def test_stabilizer_library_defaults_local_motion_2x2(ctx, oracle, clip):
    """The CLI's configuration: library defaults = local-motion least squares on a 2x2 mesh (8 unknowns), 256x256 tracking."""
    frames, _ = clip
    assert _run_pair(oracle, ctx, frames[:16], oracle_lib.preset("default", predictive_samples=3)) == 13


def test_stabilizer_affine_fallback_when_badly_distributed(ctx, oracle):
    """Texture only in one corner: distribution quality <= 0.6 -> partial-affine estimator (FrameTracker.cpp:359-374)."""
    frames, _ = synth.make_clip(360, 640, 12, seed=13, jitter=0.6)
    frames = frames.copy()
    frames[:, :, 330:, :] = 128
    frames[:, 200:, :, :] = 128
    s = oracle_lib.preset("homography", predictive_samples=2, uniformity_threshold=0.0, min_motion_samples=20)
    assert _run_pair(oracle, ctx, frames, s) == 10


def test_stabilizer_scene_cut_and_flat_frames(ctx, oracle, clip):
    """A scene cut drops the trust factor; featureless frames take the nullopt path."""
    frames, _ = clip
    other, _ = synth.make_clip(360, 640, 8, seed=99, jitter=1.0)
    flat = np.full((3, 360, 640, 3), 90, np.uint8)
    seq = np.concatenate([frames[:10], other, flat, frames[10:16]])
    s = oracle_lib.preset("homography", predictive_samples=3)
    assert _run_pair(oracle, ctx, seq, s) == len(seq) - 3


def test_stabilizer_reconfigure_and_passthrough(ctx, oracle, clip):
    frames, _ = clip
    a = oracle_lib.preset("homography", predictive_samples=3)
    b = oracle_lib.preset("homography", predictive_samples=3, stabilize_output=0)         # passthrough with delay (+ crop)
    c = oracle_lib.preset("homography", predictive_samples=5, corrective_limit_x=0.08, corrective_limit_y=0.08)
    _run_pair(oracle, ctx, frames[:24], a, reconfigure_at={8: b, 13: a, 18: c})


def test_stabilizer_restart(ctx, oracle, clip):
    import torch
    import livevisionkit_amd as lvk
    frames, _ = clip
    s = oracle_lib.preset("homography", predictive_samples=2)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    for i, f in enumerate(frames[:16]):
        if i == 7:
            ost.restart(); gst.restart()
        want, _ = ost.push(f, ts=i)
        got, _ = gst.apply(torch.from_numpy(f).cuda(), timestamp=i)
        ctx.sync()
        assert (want is None) == (got is None), i
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want), i
        assert ost.stats().scene_quality == gst.stats().scene_quality
    assert gst.frame_delay() == 2 and gst.stable_region(360, 640) == (16, 9, 608, 342)
    ost.close(); gst.close()


def test_1080p_and_4k_streams_match_oracle(ctx, oracle):
    """BASELINE configs 1-3 (720p -- the non-integer INTER_AREA downscale --, 1080p, 4K packed YUV): a short clip each, every emitted
    frame bit-identical."""
    import torch
    import livevisionkit_amd as lvk
    for (rows, cols, n) in [(720, 1280, 9), (1080, 1920, 9), (2160, 3840, 9)]:
        small, _ = synth.make_clip(rows // 4, cols // 4, n, seed=rows, jitter=1.0)
        frames = np.ascontiguousarray(small.repeat(4, axis=1).repeat(4, axis=2))            # cheap full-size frames with corners
        # relaxed quality assurance: the trust factor leaves zero within the clip (require_live_warp below), so the compared frames carry
        # the warp the tracker estimated, not the crop alone
        s = oracle_lib.preset("homography", predictive_samples=2, min_scene_quality=0.4, min_tracking_quality=0.2)
        ost = oracle_lib.OracleStabilizer(oracle, s)
        gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
        emitted = 0
        for i, f in enumerate(frames):
            want, _ = ost.push(f, ts=i, nthreads=32)
            got, _ = gst.apply(torch.from_numpy(f).cuda(), timestamp=i)
            ctx.sync()
            assert ost.stats().n_tracked == gst.stats().n_tracked
            assert (want is None) == (got is None)
            if want is not None:
                assert np.array_equal(got.cpu().numpy(), want), (rows, i)
                emitted += 1
        assert emitted == n - 2
        oracle_lib.require_live_warp(ost, f"{cols}x{rows} packed")
        assert ost.stats().trust == gst.stats().trust
        ost.close(); gst.close()


# ---- YUV420 in / out -----------------------------------------------------------------------
@pytest.mark.parametrize("nv12", [False, True])
@pytest.mark.parametrize("size", [(36, 48), (270, 480), (1080, 1920), (38, 50)])
def test_ingest_egress_yuv420_bit_exact(ctx, oracle, nv12, size):
    import torch
    rows, cols = size
    rng = np.random.default_rng(rows + cols)
    y = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    u = rng.integers(0, 256, (rows // 2, cols // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (rows // 2, cols // 2), dtype=np.uint8)
    if nv12:
        uv = np.ascontiguousarray(np.stack([u, v], -1))
        want = oracle.ingest_yuv420(y, uv)
        got = ctx.ingest_yuv420(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda())
    else:
        want = oracle.ingest_yuv420(y, u, v)
        got = ctx.ingest_yuv420(torch.from_numpy(y).cuda(), torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda())
    ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)
    frame = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    wp = oracle.egress_yuv420(frame, nv12=nv12)
    gp = ctx.egress_yuv420(torch.from_numpy(frame).cuda(), nv12=nv12)
    ctx.sync()
    for a, b in zip(gp, wp):
        assert np.array_equal(a.cpu().numpy(), b)


@pytest.mark.parametrize("nv12,overlap", [(False, False), (True, True)])
def test_stabilizer_yuv420_in_out_bit_exact(ctx, oracle, clip, nv12, overlap):
    """The OBS async path in one call: 4:2:0 planes -> ingest -> filter -> egress -> 4:2:0 planes, vs the oracle chain."""
    import torch
    import livevisionkit_amd as lvk
    frames, _ = clip
    s = oracle_lib.preset("homography", predictive_samples=3)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    if overlap:
        gst.set_overlap(True)
    emitted = 0
    for i, f in enumerate(frames[:18]):
        planes = oracle.egress_yuv420(f, nv12=nv12)                    # the 4:2:0 source material
        packed = oracle.ingest_yuv420(*planes)
        want, wts = ost.push(packed, ts=i)
        got, gts = gst.apply_yuv420(tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in planes), timestamp=i)
        ctx.sync()
        assert ost.stats().n_tracked == gst.stats().n_tracked, i
        assert (want is None) == (got is None), i
        if want is not None:
            assert wts == gts
            for a, b in zip(got, oracle.egress_yuv420(want, nv12=nv12)):
                assert np.array_equal(a.cpu().numpy(), b), i
            emitted += 1
    assert emitted == 15
    ost.close(); gst.close()


@pytest.mark.parametrize("strict", [False, True])
def test_4k_generator_clip_overlap_yuv420_bit_exact(ctx, oracle, strict):
    """Full size on NON-degenerate texture (round-5 VERDICT: the other full-size clips are 4 x pixel-replicated small clips, every 4 x 4 block flat):
    24 (relaxed QA) / 38 (strict QA) frames of SURVEY 8d's generator (tests/clipgen.py: gratings + rectangles + noise, smooth pan + AR(1) jitter) rendered at 3840 x 2160 on the
    GPU, copied down once, through the I420 overlap path free-running -- the persistent remap grid next to the tracker -- with the shipped strict
    QA preset and the relaxed one; every emitted plane bit-identical to the oracle chain, and the warp must be live (not crop only)."""
    import torch
    import livevisionkit_amd as lvk
    from tests import clipgen
    # (strict QA: the scene quality -- a moving average that starts at zero -- passes 0.95 at the 29th frame, the trust factor leaves zero after that)
    rows, cols, n = 2160, 3840, (38 if strict else 24)
    clip = clipgen.Clip(rows, cols, n, seed=0x4C564B31 + 17, device="cuda", cut_at=None)
    planes_d = [clip.render_i420(i) for i in range(n)]
    torch.cuda.synchronize()
    planes_h = [tuple(p.cpu().numpy() for p in pl) for pl in planes_d]
    del clip
    qa = {} if strict else dict(min_scene_quality=0.3, min_tracking_quality=0.2)
    s = oracle_lib.preset("homography", predictive_samples=4, **qa)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(_to_settings(s))
    gst.set_overlap(True)
    wants, gots = [], []
    for i in range(n):                                                              # the GPU stream first, back to back: a free-running caller
        got, gts = gst.apply_yuv420(planes_d[i], timestamp=i)                      # (no synchronisation between pushes, every output its own planes)
        if got is not None:
            gots.append((gts, got))
    ctx.sync()
    for i in range(n):
        want, wts = ost.push(oracle.ingest_yuv420(*planes_h[i]), ts=i, nthreads=32)
        if want is not None:
            wants.append((wts, oracle.egress_yuv420(want)))
    assert len(wants) == n - 4 and [t for t, _ in wants] == [t for t, _ in gots]
    wants, gots = [w for _, w in wants], [g for _, g in gots]
    oracle_lib.require_live_warp(ost, f"4K generator clip, strict={strict}")
    so, sg = ost.stats(), gst.stats()
    assert (so.trust, so.n_matched, so.n_tracked, list(so.homography)) == (sg.trust, sg.n_matched, sg.n_tracked, list(sg.homography))
    assert gst.schedule_counters()["remap_persistent"] >= n - 8                     # free-running: the co-scheduled grid is what ran
    for i, (w, g) in enumerate(zip(wants, gots)):
        for a, b in zip(g, w):
            assert np.array_equal(a.cpu().numpy(), b), (strict, i)
    ost.close(); gst.close()


@pytest.mark.parametrize("size,nv12", [((1080, 1920), False), ((1080, 1920), True), ((4320, 7680), False)])
def test_overlap_yuv420_full_size_persistent_grid_bit_exact(ctx, oracle, size, nv12):
    """Overlap mode at 1080p / 8K (the largest OBS canvas; 4K: test_4k_generator_clip_overlap_yuv420_bit_exact): the fused remap + 4:2:0 egress runs on the persistent grid (several strips per block, double-buffered
    chroma exchange), free-running next to the tracker; every emitted plane bit-identical to the oracle chain."""
    import torch
    import livevisionkit_amd as lvk
    rows, cols = size
    n = 9
    small, _ = synth.make_clip(rows // 4, cols // 4, n, seed=rows + 1, jitter=1.0)
    frames = np.ascontiguousarray(small.repeat(4, axis=1).repeat(4, axis=2))
    # relaxed quality assurance: the trust factor leaves zero at the fifth frame, the later remaps apply a real homography
    s = oracle_lib.preset("homography", predictive_samples=2, min_scene_quality=0.3, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    gst.set_overlap(True)
    wants, gots = [], []
    outs = []
    for i, f in enumerate(frames):
        planes = oracle.egress_yuv420(f, nv12=nv12)
        want, _ = ost.push(oracle.ingest_yuv420(*planes), ts=i, nthreads=32)
        got, _ = gst.apply_yuv420(tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in planes), timestamp=i)   # no sync between pushes
        assert (want is None) == (got is None), i
        if want is not None:
            wants.append(oracle.egress_yuv420(want, nv12=nv12)); gots.append(got)
    ctx.sync()
    assert len(wants) == n - 2
    oracle_lib.require_live_warp(ost, f"{cols}x{rows} overlap nv12={nv12}")
    for i, (w, g) in enumerate(zip(wants, gots)):
        for a, b in zip(g, w):
            assert np.array_equal(a.cpu().numpy(), b), (size, i)
    ost.close(); gst.close()


@pytest.mark.parametrize("fmt", [0, 2])
def test_stabilizer_bgr_rgb_frames_bit_exact(ctx, oracle, clip, fmt):
    """VideoFrame formats BGR (0) / RGB (2): tracking luma = cvtColor(..2GRAY), remap = the RGB EASU program."""
    import torch
    import livevisionkit_amd as lvk
    frames, _ = clip
    s = oracle_lib.preset("homography", predictive_samples=3)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    produced = 0
    for i, f in enumerate(frames[:12]):
        want, wts = ost.push(f, ts=i, fmt=fmt)
        got, gts = gst.apply(torch.from_numpy(f).cuda(), timestamp=i, fmt=fmt)
        ctx.sync()
        so, sg = ost.stats(), gst.stats()
        assert (so.n_detected, so.n_matched, so.n_tracked, so.tracking_stability) == (sg.n_detected, sg.n_matched, sg.n_tracked, sg.tracking_stability), i
        assert (want is None) == (got is None)
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want), i
            produced += 1
    assert produced == 9
    # the formats really differ: the same clip pushed as YUV tracks another luma and resamples with another program
    ost2 = oracle_lib.OracleStabilizer(oracle, s)
    outs = [ost2.push(f, ts=i)[0] for i, f in enumerate(frames[:5])]
    ost3 = oracle_lib.OracleStabilizer(oracle, s)
    outs_f = [ost3.push(f, ts=i, fmt=fmt)[0] for i, f in enumerate(frames[:5])]
    assert not np.array_equal(outs[4], outs_f[4])
    ost.close(); gst.close(); ost2.close(); ost3.close()


def test_mixing_push_flavours_needs_a_restart(ctx):
    """Borrowed packed frames and pooled 4:2:0 frames cannot share one queue: the switch is refused until restart()."""
    import torch
    import livevisionkit_amd as lvk
    s = _to_settings(oracle_lib.preset("homography", predictive_samples=3))
    f = lvk.StabilizationFilter(s, context=ctx)
    packed = torch.zeros((270, 480, 3), dtype=torch.uint8, device="cuda")
    planes = (torch.zeros((270, 480), dtype=torch.uint8, device="cuda"), torch.zeros((135, 240), dtype=torch.uint8, device="cuda"),
              torch.zeros((135, 240), dtype=torch.uint8, device="cuda"))
    f.apply(packed)
    with pytest.raises(lvk.LvkHipError):
        f.apply_yuv420(planes)
    f.restart()
    f.apply_yuv420(planes)
    with pytest.raises(lvk.LvkHipError):
        f.apply(packed)
    f.restart()
    f.apply(packed)
    ctx.sync(); f.close()


@pytest.mark.parametrize("entry", ["device", "host", "host-copy"])
def test_yuv420_resolution_change_mid_stream(ctx, oracle, entry, monkeypatch):
    """A 4:2:0 stream whose frame size changes (an OBS source that is resized; VSFilter.cpp does not restart its filter): tracker and path
    smoother carry on, and -- like the reference, whose queue holds whole VideoFrames (StabilizationFilter.cpp:118-131) -- the frames still queued
    at the old size LEAVE AT THEIR OWN SIZE over the next frame_delay pushes (round 6; rounds 2-5 dropped them).  lvk_hip_stab_next_output sizes the
    output planes; planes sized from the incoming frame are refused when the delayed one is larger, before anything changes; every emitted frame
    equals the oracle's frame of the same timestamp.  Device planes, pinned host planes (direct sink) and the download route (copy sink)."""
    import torch
    import livevisionkit_amd as lvk
    if entry == "host-copy":
        monkeypatch.setenv("LVK_HIP_HOST_SINK", "copy")
    a, _ = synth.make_clip(360, 640, 10, seed=31, jitter=1.0)
    b, _ = synth.make_clip(270, 480, 8, seed=31, jitter=1.0)
    c, _ = synth.make_clip(360, 640, 8, seed=31, jitter=1.0)
    frames = list(a) + list(b) + list(c[:8])
    s = oracle_lib.preset("homography", predictive_samples=3, min_scene_quality=0.3, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx); gst.set_overlap(True)
    want, got, refused = {}, {}, 0
    for i, f in enumerate(frames):
        planes = oracle.egress_yuv420(f)
        big = np.zeros((360, 640, 3), np.uint8)                                  # the oracle may emit a frame of another size than it is given
        w, wts = ost.push(oracle.ingest_yuv420(*planes), ts=i, out=big)
        if w is not None:
            r, c_ = frames[wts].shape[:2]
            want[wts] = oracle.egress_yuv420(np.ascontiguousarray(big[:r, :c_]))
        due = gst.next_output(f.shape[0], f.shape[1])
        assert (due is None) == (w is None), i
        if entry == "device":
            dp = tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in planes)
            if due is not None and due[0] > f.shape[0]:
                with pytest.raises(lvk.LvkHipError, match="DELAYED"):             # output planes sized from the INCOMING frame: refused, nothing changes
                    gst.apply_yuv420(dp, timestamp=i, out=tuple(torch.empty_like(p) for p in dp))
                refused += 1
            g, gts = gst.apply_yuv420(dp, timestamp=i)
            ctx.sync()
            if g is not None:
                got[gts] = [p.cpu().numpy() for p in g]
        else:
            hin = gst.host_planes(*f.shape[:2])
            for d, p in zip(hin, planes):
                d[...] = p
            if due is not None and due[0] > f.shape[0]:
                with pytest.raises(lvk.LvkHipError, match="DELAYED"):
                    gst.apply_yuv420_host_prepared(gst.prepare_yuv420_host(hin), i, gst.prepare_yuv420_host(gst.host_planes(*f.shape[:2])))
                refused += 1
            hout = gst.host_planes(*(due[:2] if due else f.shape[:2]))
            g, _ = gst.apply_yuv420_host_prepared(gst.prepare_yuv420_host(hin), i, gst.prepare_yuv420_host(hout))
            ctx.sync()
            if g is not None:
                got[gst._ots.value] = [np.array(p) for p in hout]
        assert np.array_equal(gst.features(), ost.features()), i
    assert sorted(want) == list(range(0, len(frames) - 3))                      # the oracle emits every frame, old sizes included ...
    assert sorted(got) == sorted(want)                                          # ... and so does the library
    assert refused == 3                                                         # the three 640x360 frames that left while 480x270 frames came in
    for ts, planes in got.items():
        for p, q in zip(planes, want[ts]):
            assert p.shape == q.shape and np.array_equal(p, q), ts
    ost.close(); gst.close()


def _tie_clip(rows, cols, n):
    """A checkerboard of flat squares under a slow drift: hundreds of corners with IDENTICAL scores, several to a suppression-grid cell --
    the grid keeps the FIRST of the strongest ones (FeatureDetector.cpp:150 `>`), so the order the corners are met in decides."""
    frames = []
    yy, xx = np.mgrid[0:rows, 0:cols]
    for i in range(n):
        f = np.zeros((rows, cols, 3), np.uint8)
        sx, sy = xx + 2 * i, yy + (i % 3)
        f[..., 0] = np.where(((sx // 9) + (sy // 7)) % 2 == 1, 200, 40) + ((sx // 45 + sy // 35) % 3) * 8
        f[..., 1] = 128; f[..., 2] = 120
        frames.append(f)
    return frames


@pytest.mark.parametrize("preset", ["homography", "field"])
def test_suppression_grid_on_the_device(ctx, oracle, clip, preset, monkeypatch):
    """Frames on which the detector runs put their corners through the suppression grid INSIDE the chain (k_fast_insert): no host loop, no
    second synchronisation.  The feature list (positions, responses, ages, ORDER), the counts and the distribution quality must equal the
    oracle's and the host loop's (LVK_HIP_HOST_GRID=1) frame by frame -- on the jittering clip, on a clip full of equal-score corners,
    through a restart, and with the relaxed and strict quality thresholds (early-outs decided by the kernel)."""
    import torch
    import livevisionkit_amd as lvk
    for frames, over in ((clip[0], dict(min_scene_quality=0.4, min_tracking_quality=0.2)), (_tie_clip(360, 640, 16), dict()),
                         (clip[0][:12], dict(uniformity_threshold=0.95, min_motion_samples=75))):          # the last: distribution < threshold, nothing tracked
        so = oracle_lib.preset(preset, predictive_samples=2, **over)
        ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(so)
        gdev = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gdev.configure(_to_settings(so))
        monkeypatch.setenv("LVK_HIP_HOST_GRID", "1")
        ghost = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); ghost.configure(_to_settings(so))
        monkeypatch.delenv("LVK_HIP_HOST_GRID")
        for i, f in enumerate(frames):
            if i == 9:
                ost.restart(); gdev.restart(); ghost.restart()
            d = torch.from_numpy(f).cuda()
            ost.push(f, ts=i); gdev.apply(d, timestamp=i); ghost.apply(d.clone(), timestamp=i)
            ctx.sync()
            s0, s1, s2 = ost.stats(), gdev.stats(), ghost.stats()
            for k in ("n_detected", "n_matched", "n_tracked", "distribution", "tracking_stability", "trust"):
                assert getattr(s0, k) == getattr(s1, k) == getattr(s2, k), (preset, i, k, getattr(s0, k), getattr(s1, k), getattr(s2, k))
            f0, f1, f2 = ost.features(), gdev.features(), ghost.features()
            assert np.array_equal(f0.view(np.uint32), f1.view(np.uint32)), (preset, i)
            assert np.array_equal(f0.view(np.uint32), f2.view(np.uint32)), (preset, i)
            assert np.array_equal(ost.meshes()[0].view(np.uint32), gdev.meshes()[0].view(np.uint32)), (preset, i)
        dev, host = gdev.detector_frames()
        assert dev >= 3 and host == 0, (preset, dev, host)
        dev, host = ghost.detector_frames()
        assert dev == 0 and host >= 3, (preset, dev, host)
        ost.close(); gdev.close(); ghost.close()


def _run_planes(gst, ctx, clips, order, announce, sync_every=False, out_sets=4):
    """Push `order` (indices into the prepared plane sets `clips`) through apply_yuv420_prepared; `announce(k)` = index to announce before the
    k-th push, or None.  Returns the emitted planes as numpy arrays."""
    import torch
    outs = [tuple(torch.empty_like(p) for p in clips[0]["planes"]) for _ in range(out_sets)]
    outs_p = [gst.prepare_yuv420(o) for o in outs]
    emitted = []
    for k, idx in enumerate(order):
        a = announce(k)
        if a is not None:
            gst.prefetch_yuv420_prepared(clips[a])
        got, _ = gst.apply_yuv420_prepared(clips[idx], k, outs_p[k % out_sets])
        if sync_every or got is not None:
            ctx.sync()                                                   # (the output planes are reused every out_sets pushes: read them now)
        if got is not None:
            emitted.append(tuple(p.cpu().numpy().copy() for p in got))
    ctx.sync()
    return emitted


@pytest.mark.parametrize("preset,nv12", [("homography", False), ("field", True)])
def test_device_lookahead_same_frames(ctx, oracle, clip, preset, nv12):
    """lvk_hip_stab_prefetch_yuv420: announcing frame n + 1 before pushing frame n moves its downscale + pyramid behind frame n's chain and nothing
    else -- every emitted plane equals the run without announcements (which the tests above hold to the oracle); a wrong announcement, a
    restart in between and buffers whose CONTENT changes behind the same addresses fall back to the plain path."""
    import torch
    import livevisionkit_amd as lvk
    frames, _ = clip
    n = 24
    s = oracle_lib.preset(preset, predictive_samples=3, min_scene_quality=0.3, min_tracking_quality=0.2)

    def make():
        g = lvk.StabilizationFilter(_to_settings(s), context=ctx); g.set_overlap(True); return g

    plane_sets = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in oracle.egress_yuv420(f, nv12=nv12)) for f in frames[:n]]
    gst = make(); clips = [gst.prepare_yuv420(p) for p in plane_sets]
    plain = _run_planes(gst, ctx, clips, list(range(n)), lambda k: None)
    assert gst.lookahead_frames() == 0
    gst.close()
    assert len(plain) == n - 3

    gst = make(); clips = [gst.prepare_yuv420(p) for p in plane_sets]
    ahead = _run_planes(gst, ctx, clips, list(range(n)), lambda k: k + 1 if k + 1 < n else None)
    hits = gst.lookahead_frames()
    gst.close()
    assert hits >= n - 3, hits                                            # (the first push has no chain to hide anything behind)
    assert len(ahead) == len(plain)
    for i, (a, b) in enumerate(zip(ahead, plain)):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (preset, i)

    # wrong announcements (another frame than the one pushed next), every other push
    gst = make(); clips = [gst.prepare_yuv420(p) for p in plane_sets]
    wrong = _run_planes(gst, ctx, clips, list(range(n)), lambda k: (k + 5) % n if k % 2 else (k + 1 if k + 1 < n else None))
    assert 0 < gst.lookahead_frames() < hits
    gst.close()
    for i, (a, b) in enumerate(zip(wrong, plain)):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), ("wrong announcement", i)

    # two buffers only, rewritten between pushes: an announcement is good for the very next push and never outlives it
    gst = make()
    bufs = [tuple(torch.empty_like(p) for p in plane_sets[0]) for _ in range(2)]
    bufs_p = [gst.prepare_yuv420(b) for b in bufs]
    outs = [tuple(torch.empty_like(p) for p in plane_sets[0]) for _ in range(2)]
    outs_p = [gst.prepare_yuv420(o) for o in outs]
    reused = []
    for k in range(n):
        if k == 0:
            for d, src in zip(bufs[0], plane_sets[0]): d.copy_(src)
        if k + 1 < n and k % 3 != 2:                                     # announce two of three: fill the other buffer first, then announce it
            ctx.sync()
            for d, src in zip(bufs[(k + 1) % 2], plane_sets[k + 1]): d.copy_(src)
            torch.cuda.synchronize()
            gst.prefetch_yuv420_prepared(bufs_p[(k + 1) % 2])
        got, _ = gst.apply_yuv420_prepared(bufs_p[k % 2], k, outs_p[k % 2])
        ctx.sync()
        if got is not None:
            reused.append(tuple(p.cpu().numpy().copy() for p in got))
        if k + 1 < n and k % 3 == 2:                                      # not announced: the buffer gets its new content only now
            for d, src in zip(bufs[(k + 1) % 2], plane_sets[k + 1]): d.copy_(src)
            torch.cuda.synchronize()
    gst.close()
    assert len(reused) == len(plain)
    for i, (a, b) in enumerate(zip(reused, plain)):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), ("reused buffers", i)

    # a restart between the announcement and the push forgets it; so does prefetch_cancel
    def with_restart(announcing):
        g = make(); c = [g.prepare_yuv420(p) for p in plane_sets]
        _run_planes(g, ctx, c, list(range(6)), (lambda k: k + 1) if announcing else (lambda k: None))
        before = g.lookahead_frames()
        if announcing:
            g.prefetch_yuv420_prepared(c[7])
        g.restart()
        out = _run_planes(g, ctx, c, list(range(n)), (lambda k: k + 1 if k + 1 < n else None) if announcing else (lambda k: None))
        if announcing:
            g.prefetch_yuv420_prepared(c[0]); g.prefetch_cancel()
            assert g.lookahead_frames() >= before + n - 3
        g.close()
        return out
    a_run, b_run = with_restart(True), with_restart(False)
    assert len(a_run) == len(b_run) and len(a_run) >= n - 3
    for i, (a, b) in enumerate(zip(a_run, b_run)):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), ("after restart", i)


def test_device_lookahead_packed_frames(ctx, oracle, clip):
    """lvk_hip_stab_prefetch (packed BGR frames: the grey value is what is read ahead) against the plain run."""
    import torch
    import livevisionkit_amd as lvk
    frames, _ = clip
    n = 16
    s = oracle_lib.preset("homography", predictive_samples=2, min_scene_quality=0.3, min_tracking_quality=0.2)
    dev = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in frames[:n]]
    runs = []
    for announce in (False, True):
        gst = lvk.StabilizationFilter(_to_settings(s), context=ctx); gst.set_overlap(True)
        outs = []
        for k in range(n):
            if announce and k + 1 < n:
                gst.prefetch(dev[k + 1], fmt=0)
            got, _ = gst.apply(dev[k], timestamp=k, fmt=0)
            if got is not None:
                ctx.sync(); outs.append(got.cpu().numpy().copy())
        ctx.sync()
        assert (gst.lookahead_frames() >= n - 2) == announce
        runs.append(outs); gst.close()
    assert len(runs[0]) == len(runs[1]) == n - 2
    for i, (a, b) in enumerate(zip(*runs)):
        assert np.array_equal(a, b), i


def test_refused_first_overlap_push_reports_the_refusal_and_leaves_no_error_behind(ctx):
    """A push that is refused (output planes too small for the DELAYED frame) launches no conversion.  When it is the first push of the filter's life that
    would have put one on the bulk stream -- the frames before it went through the passthrough (stabilize_output off: conversion inline) --, the event the
    push waits for at its end does not exist yet: until round 6 that wait failed with "invalid resource handle", replaced the refusal's message and left a
    sticky runtime error for the next (valid) launch to trip over (found by seeds 389 / 390 of the OBS-format fuzz sweep).  Reference behaviour this
    protects: a resize in the middle of a stream emits the queued frames at their own size (StabilizationFilter.cpp:118-131, WarpMesh.cpp:183-223)."""
    import torch
    import livevisionkit_amd as lvk
    big, small = (432, 768), (360, 640)
    s = oracle_lib.preset("homography", predictive_samples=2, stabilize_output=0)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    gst.set_overlap(True)

    def planes(size, v):
        r, c = size
        return (torch.full((r, c), v, dtype=torch.uint8, device="cuda"), torch.full((r // 2, c // 2), 128, dtype=torch.uint8, device="cuda"),
                torch.full((r // 2, c // 2), 128, dtype=torch.uint8, device="cuda"))

    for i in range(2):
        out, _ = gst.apply_yuv420(planes(big, 40 + i), timestamp=i)
        assert out is None
    s.stabilize_output = 1
    gst.configure(_to_settings(s))                                            # from here on the conversion is a bulk-stream kernel with an event behind it
    assert gst.next_output(*small)[:2] == big                                 # the third push emits the first frame, at ITS size
    too_small = planes(small, 0)
    with pytest.raises(lvk.LvkHipError, match="DELAYED"):
        gst.apply_yuv420(planes(small, 50), timestamp=2, out=too_small)
    assert gst.next_output(*small)[:2] == big                                 # nothing was queued
    out, ts = gst.apply_yuv420(planes(small, 50), timestamp=2)                # planes of the right size: the push goes through ...
    ctx.sync()
    assert ts == 0 and tuple(out[0].shape) == big
    out, ts = gst.apply_yuv420(planes(small, 51), timestamp=3)                # ... and so does the next one: no stale runtime error
    ctx.sync()
    assert ts == 1 and tuple(out[0].shape) == big
    gst.close()
