"""Input-borrow mode (lvk_hip_stab_set_input_borrow, round 6): the caller lends the device planes of lvk_hip_stab_push_yuv420 until the NEXT push has
returned -- the reference's ownership, VideoFilter::apply(std::move(frame), ..) moves the input into the filter's queue (Filters/StabilizationFilter.cpp:118)
-- and a free-running caller's 4:2:0 -> 4:4:4 conversion (I4XXIngest / NV12Ingest::to_ocl, Modules/OBS-Plugin/Interop/FrameIngest.cpp:494-557) rides as
side work inside the output remap of the delayed frame (k_remap_*_420_ingest) instead of being a kernel of its own.  The pixels must not know: the fused
kernel against the two separate ones, and whole streams against the oracle chain (bit-exact: integer conversion, the remap's binary32 sequence)."""
import ctypes

import numpy as np
import pytest

from tests import oracle_lib, synth

pytestmark = pytest.mark.gpu


def _conv(o):
    import livevisionkit_amd as lvk
    s = lvk.StabilizationFilterSettings()
    ctypes.memmove(ctypes.byref(s), ctypes.byref(o), ctypes.sizeof(o))
    return s


@pytest.mark.parametrize("co", [False, True])
@pytest.mark.parametrize("nv12", [False, True])
@pytest.mark.parametrize("size,new_size", [((36, 48), (36, 48)), ((270, 480), (270, 480)), ((1080, 1920), (1080, 1920)), ((38, 52), (64, 20)), ((2160, 3840), (720, 1280))])
def test_fused_remap_ingest_kernel_equals_the_two_kernels(ctx, oracle, size, new_size, nv12, co):
    """lvk_hip_warpmesh_apply_yuv420_ingest == lvk_hip_warpmesh_apply_yuv420 + lvk_hip_ingest_yuv420, and the conversion == the oracle's: full grid and the
    persistent grid, homography and mesh kernels, a new frame of another size than the remapped one (a stream in the middle of a resize), sizes whose
    conversion has fewer / more units than the launch has blocks."""
    import torch
    rows, cols = size
    nrows, ncols = new_size
    rng = np.random.default_rng(rows * 7 + ncols)
    src = torch.from_numpy(rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)).cuda()
    y = rng.integers(0, 256, (nrows, ncols), dtype=np.uint8)
    u = rng.integers(0, 256, (nrows // 2, ncols // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (nrows // 2, ncols // 2), dtype=np.uint8)
    planes_h = (y, np.ascontiguousarray(np.stack([u, v], -1))) if nv12 else (y, u, v)
    planes = tuple(torch.from_numpy(p).cuda() for p in planes_h)
    want_new = oracle.ingest_yuv420(*planes_h)
    for mesh in (np.array([[[0.004, -0.003], [-0.002, 0.004]], [[0.003, 0.002], [-0.004, -0.002]]], np.float32), synth.random_mesh(16, 16, rng, amp=0.01)):
        ref_out = ctx.warpmesh_apply_yuv420(src, mesh, nv12=nv12)
        guard = torch.full((nrows + 2, ncols * 3 + 8), 0xA5, dtype=torch.uint8, device="cuda")
        new = guard[1:nrows + 1, 4:4 + 3 * ncols].unflatten(1, (ncols, 3))
        assert new.data_ptr() % 4 == 0 and new.stride(0) % 4 == 0
        out, _ = ctx.warpmesh_apply_yuv420_ingest(src, mesh, planes, new_frame=new, nv12=nv12, co=co)
        ctx.sync()
        for a, b in zip(out, ref_out):
            assert torch.equal(a, b), (size, mesh.shape)
        assert np.array_equal(new.cpu().numpy(), want_new), (size, new_size, mesh.shape)
        g = guard.cpu().numpy()
        assert (g[0] == 0xA5).all() and (g[-1] == 0xA5).all() and (g[:, :4] == 0xA5).all() and (g[:, 4 + 3 * ncols:] == 0xA5).all()      # nothing outside the new frame


@pytest.mark.parametrize("nv12,preset,lens", [(False, "homography", False), (True, "homography", False), (False, "field", False), (False, "homography", True)])
def test_borrow_mode_free_running_stream_bit_exact(ctx, oracle, nv12, preset, lens):
    """A free-running 4K stream in input-borrow mode (4K: the remap outlasts the host's turn between two pushes, the caller is seen running free): every emitted plane bit-identical to the oracle chain, the conversions ran inside the remaps
    (schedule counters), with a ring of TWO input plane sets that is overwritten as early as the contract allows -- set k is refilled right after the push
    that follows its own has returned (device-to-device copies of ~50 us: the bulk stream is still busy, the caller still runs free).  The same stream without the mode gives the same planes."""
    import torch
    import livevisionkit_amd as lvk
    rows, cols, n, delay = 2160, 3840, 14, 3
    small, _ = synth.make_clip(rows // 4, cols // 4, n, seed=77, jitter=1.0)
    frames = np.ascontiguousarray(small.repeat(4, axis=1).repeat(4, axis=2))
    planes_h = [oracle.egress_yuv420(f, nv12=nv12) for f in frames]
    s = oracle_lib.preset(preset, predictive_samples=delay, min_scene_quality=0.3, min_tracking_quality=0.2)
    prof = np.array([0.8 * cols, 0.8 * cols, cols / 2.0, rows / 2.0, -0.12, 0.03, 0.0, 0.0, 0.0]) if lens else None
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    if lens:
        ost.set_lens(prof)
    wants = []
    for i in range(n):
        w, wts = ost.push(oracle.ingest_yuv420(*planes_h[i]), ts=i, nthreads=32)
        if w is not None:
            wants.append((wts, oracle.egress_yuv420(w, nv12=nv12)))
    oracle_lib.require_live_warp(ost, "borrow-mode stream")
    staged = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pl) for pl in planes_h]       # where the ring is refilled from (device copies)
    results = {}
    ts_ = torch.cuda.Stream()
    own = lvk.Context(0, stream=ts_)                       # the context's stream IS the torch stream the ring is refilled on: no host synchronisation between pushes
    for borrow in (True, False):
        gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=own); gst.configure(_conv(s))
        gst.set_overlap(True); gst.set_input_borrow(borrow)
        if lens:
            gst.set_lens(prof)
        ring = [tuple(torch.empty_like(p) for p in staged[0]) for _ in range(2)]
        ring_args = [gst.prepare_yuv420(r) for r in ring]
        outs = [tuple(torch.empty_like(p) for p in staged[0]) for _ in range(n)]
        out_args = [gst.prepare_yuv420(o) for o in outs]
        gots = []
        torch.cuda.synchronize()
        for i in range(n):
            # set i % 2 was lent by push i - 2 and came back when push i - 1 returned: refill it now, on the context's stream (the push orders itself behind what that stream holds)
            with torch.cuda.stream(ts_):
                for d, p in zip(ring[i % 2], staged[i]):
                    d.copy_(p, non_blocking=True)
            got, gts = gst.apply_yuv420_prepared(ring_args[i % 2], i, out_args[i])
            if got is not None:
                gots.append((gts, got))
        own.sync()
        c = gst.schedule_counters()
        if borrow:
            assert c["ingest_fused"] >= n - delay - 3, sorted(c.items())                     # (the first pushes after the start fill the delay / are seen as synchronous)
            assert c["ingest_fused"] + c["ingest_on_tracker"] + c["ingest_on_bulk"] + c["ingest_inline"] == n, c
        else:
            assert c["ingest_fused"] == 0, c
        assert [t for t, _ in gots] == [t for t, _ in wants]
        for (ts, g), (_, w) in zip(gots, wants):
            for k, (a, b) in enumerate(zip(g, w)):
                assert np.array_equal(a.cpu().numpy(), b), (borrow, ts, k)
        results[borrow] = gots
        so, sg = ost.stats(), gst.stats()
        assert (so.trust, so.n_matched, so.n_tracked) == (sg.trust, sg.n_matched, sg.n_tracked)
        gst.close()
    ost.close(); own.close()


def test_borrow_mode_synchronised_caller_and_switch_off(ctx, oracle):
    """A caller that waits for every frame is scheduled as without the mode (no conversion inside a remap, planes consumed at return: they are overwritten
    right after every push here); switching the mode off in the middle of a free-running stream hands the lent planes back before it returns."""
    import torch
    import livevisionkit_amd as lvk
    rows, cols, n, delay = 432, 768, 14, 2
    frames, _ = synth.make_clip(rows, cols, n, seed=5, jitter=1.0)
    planes_h = [oracle.egress_yuv420(f) for f in frames]
    s = oracle_lib.preset("homography", predictive_samples=delay, min_scene_quality=0.3, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    wants = {}
    for i in range(n):
        w, wts = ost.push(oracle.ingest_yuv420(*planes_h[i]), ts=i)
        if w is not None:
            wants[wts] = oracle.egress_yuv420(w)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(_conv(s))
    gst.set_overlap(True); gst.set_input_borrow(True)
    one = tuple(torch.empty_like(torch.from_numpy(p)).cuda() for p in planes_h[0])
    gots = {}
    import time
    for i in range(7):                                                       # synchronised: ONE plane set, refilled after every push
        for d, p in zip(one, planes_h[i]):
            d.copy_(torch.from_numpy(p))
        got, gts = gst.apply_yuv420(one, timestamp=i)
        ctx.sync(); time.sleep(0.002)
        if got is not None:
            gots[gts] = got
    assert gst.schedule_counters()["ingest_fused"] == 0, gst.schedule_counters()
    ring = [tuple(torch.empty_like(p) for p in one) for _ in range(3)]
    for i in range(7, n):                                                    # free-running; the mode goes off after push 10
        for d, p in zip(ring[i % 3], planes_h[i]):
            d.copy_(torch.from_numpy(p), non_blocking=False)
        got, gts = gst.apply_yuv420(ring[i % 3], timestamp=i)
        if got is not None:
            gots[gts] = got
        if i == 10:
            gst.set_input_borrow(False)
            for t in ring[i % 3]:
                t.fill_(0)                                                   # the planes of push 10 are the caller's again
    ctx.sync()
    assert sorted(gots) == sorted(wants)
    for ts in sorted(gots):
        for a, b in zip(gots[ts], wants[ts]):
            assert np.array_equal(a.cpu().numpy(), b), ts
    ost.close(); gst.close()
