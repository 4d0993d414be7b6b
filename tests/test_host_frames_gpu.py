"""Host-resident frames (SURVEY.md section 8d's metric; FrameIngest::upload_planes / download_planes, Modules/OBS-Plugin/Interop/
FrameIngest.cpp:415-474,567-602): lvk_hip_stab_push_yuv420_host against the oracle and against the device-resident entry point.
Both output routes -- the remap kernel writing the pinned host planes itself ("direct") and remap -> device planes -> download ("copy")
-- and the per-push choice between them must give the same bytes; contiguous (one copy) and pitched (2-D copies) planes; I420 and NV12."""
import ctypes as _c

import numpy as np
import pytest

from tests import clipgen, oracle_lib, synth

pytestmark = pytest.mark.gpu


def _settings(lvk, so):
    sg = lvk.StabilizationFilterSettings()
    _c.memmove(_c.byref(sg), _c.byref(so), _c.sizeof(so))
    return sg


@pytest.mark.parametrize("sink", ["direct", "copy", "auto"])
@pytest.mark.parametrize("nv12", [False, True])
def test_host_push_matches_the_oracle(ctx, oracle, monkeypatch, sink, nv12):
    import livevisionkit_amd as lvk
    if sink != "auto":
        monkeypatch.setenv("LVK_HIP_HOST_SINK", sink)
    rows, cols, n = 540, 960, 14
    frames, _ = synth.make_clip(rows, cols, n, seed=31, jitter=1.0)
    so = oracle_lib.preset("homography", predictive_samples=3, min_scene_quality=0.3, min_tracking_quality=0.2)      # (the trust factor leaves zero: real homographies)
    ost = oracle_lib.OracleStabilizer(oracle, so)
    gst = lvk.StabilizationFilter(_settings(lvk, so), context=ctx)
    gst.set_overlap(True)
    ins = [gst.host_planes(rows, cols, nv12) for _ in range(2)]
    outs = [gst.host_planes(rows, cols, nv12) for _ in range(3)]
    ins_a = [gst.prepare_yuv420_host(p) for p in ins]; outs_a = [gst.prepare_yuv420_host(p) for p in outs]
    emitted = 0
    for i, f in enumerate(frames):
        planes = oracle.egress_yuv420(f, nv12=nv12)
        want, _ = ost.push(oracle.ingest_yuv420(*planes), ts=i)
        for dst, src in zip(ins[i % 2], planes):
            dst[...] = src
        got, ts = gst.apply_yuv420_host_prepared(ins_a[i % 2], i, outs_a[i % 3])
        for dst in ins[i % 2]:
            dst[...] = 99                                # consumed on return: scribbling over the input must not matter
        if sink == "auto" and i % 3:
            ctx.sync()                                   # a caller that sometimes waits, sometimes runs free: both routes in one stream
        assert (want is None) == (got is None), i
        if want is not None:
            ctx.sync()
            assert ts == i - 3
            for a, b in zip(got, oracle.egress_yuv420(want, nv12=nv12)):
                assert np.array_equal(np.asarray(a), b), (i, sink, nv12)
            emitted += 1
    assert emitted == n - 3
    ost.close(); gst.close()


@pytest.mark.parametrize("preset", ["homography", "field"])
def test_host_push_equals_device_push_at_4k(ctx, preset, monkeypatch):
    """3840 x 2160, free-running: the host entry point (pitched input planes: 2-D copies; contiguous output: one download) emits exactly
    the planes of lvk_hip_stab_push_yuv420 on the same clip; then the direct route on the same stream."""
    import torch
    import livevisionkit_amd as lvk
    rows, cols, n = 2160, 3840, 26
    clip = clipgen.Clip(rows, cols, n, device="cuda")
    planes = [clip.render_i420(i) for i in range(n)]
    torch.cuda.synchronize()
    s = lvk.StabilizationFilterSettings.obs_preset(preset, strict=False, predictive_samples=4)       # relaxed QA: the outputs depend on the tracker

    def device_run():
        f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); f.configure(s); f.set_overlap(True)
        outs = []
        for i in range(n):
            got, _ = f.apply_yuv420(planes[i], timestamp=i)
            if got is not None:
                ctx.sync(); outs.append([p.cpu().numpy().copy() for p in got])
        f.close()
        return outs

    def host_run(sink):
        if sink:
            monkeypatch.setenv("LVK_HIP_HOST_SINK", sink)
        else:
            monkeypatch.delenv("LVK_HIP_HOST_SINK", raising=False)
        f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); f.configure(s); f.set_overlap(True)
        pad = 64
        hy = torch.empty((rows, cols + pad), dtype=torch.uint8).pin_memory(); hu = torch.empty((rows // 2, cols // 2 + pad), dtype=torch.uint8).pin_memory()
        hv = torch.empty((rows // 2, cols // 2 + pad), dtype=torch.uint8).pin_memory()
        src = (hy.numpy()[:, :cols], hu.numpy()[:, :cols // 2], hv.numpy()[:, :cols // 2])
        ring = [f.host_planes(rows, cols) for _ in range(n)]
        ring_a = [f.prepare_yuv420_host(p) for p in ring]
        # a second input set (contiguous, lvk_hip_host_malloc) alternates with the pitched one; from frame 8 on the NEXT frame is announced
        # (lvk_hip_stab_prefetch_yuv420_host: one whole-frame copy on alternating upload streams) before the current one is pushed
        srcs = [src, f.host_planes(rows, cols), f.host_planes(rows, cols)]
        srcs_a = [f.prepare_yuv420_host(p) for p in srcs]
        emitted = []

        def fill(i):
            for dst, p in zip(srcs[i % 3], planes[i]):
                dst[...] = p.cpu().numpy()
        fill(0)
        for i in range(n):
            if i == 8:
                f.prefetch_yuv420_host_prepared(srcs_a[i % 3])  # announced frames are pushed in the order announced: this one first
            if i >= 8 and i + 1 < n:
                fill(i + 1)                                    # (its buffer was consumed when push i - 2 returned)
                f.prefetch_yuv420_host_prepared(srcs_a[(i + 1) % 3])
            got, _ = f.apply_yuv420_host_prepared(srcs_a[i % 3], i, ring_a[i])          # no synchronisation between the pushes
            if i < 8 and i + 1 < n:
                fill(i + 1)
            if got is not None:
                emitted.append(i)
        ctx.sync()
        outs = [[np.array(p) for p in ring[i]] for i in emitted]
        f.close()
        return outs

    want = device_run()
    assert len(want) == n - 4
    for sink in ("copy", "direct", None):
        got = host_run(sink)
        assert len(got) == len(want)
        for k, (a, b) in enumerate(zip(got, want)):
            for pa, pb in zip(a, b):
                assert np.array_equal(pa, pb), (preset, sink, k)


def test_out_of_order_push_is_refused_with_a_message(ctx):
    """Look-ahead frames are pushed in the order announced; pushing another frame first is an argument error that says so, and the
    stream continues once the announced frame is pushed."""
    import livevisionkit_amd as lvk
    rows, cols = 360, 640
    f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
    a, b, out = (f.host_planes(rows, cols) for _ in range(3))
    for p in a + b:
        p[...] = 90
    pa, pb, po = (f.prepare_yuv420_host(p) for p in (a, b, out))
    f.prefetch_yuv420_host_prepared(pb)
    with pytest.raises(Exception, match="order announced"):
        f.apply_yuv420_host_prepared(pa, 0, po)
    f.apply_yuv420_host_prepared(pb, 0, po)
    f.apply_yuv420_host_prepared(pa, 1, po)
    ctx.sync()
    f.close()


def test_announced_frames_are_forgotten_on_restart_and_cancel(ctx):
    """A caller that announced frame n + 1 and then restarts, seeks or switches buffers: restart() and lvk_hip_stab_prefetch_cancel
    forget the announcement -- later pushes with other pointers go through (they used to be refused until the stale pointers were pushed), and
    a REUSED buffer pushed after the restart carries its new content, not the upload started before it."""
    import livevisionkit_amd as lvk
    rows, cols = 360, 640
    f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings.obs_preset("homography", predictive_samples=1, apply_crop=False), context=ctx)
    a, b, o0, o1 = (f.host_planes(rows, cols) for _ in range(4))
    yy, xx = np.mgrid[0:rows, 0:cols]
    tex = np.rint(128 + 60 * np.sin(xx / 37.0) + 40 * np.cos(yy / 23.0)).astype(np.uint8)      # smooth: an identity warp returns it (edge-adaptive filter)
    for p, val in ((a, 60), (b, 200)):
        p[0][...] = val; p[1][...] = 110; p[2][...] = 140
    pa, pb, p0, p1 = (f.prepare_yuv420_host(p) for p in (a, b, o0, o1))
    f.apply_yuv420_host_prepared(pa, 0, p0)
    f.prefetch_yuv420_host_prepared(pb)                     # announced, never pushed
    f.restart()
    f.apply_yuv420_host_prepared(pa, 1, p0)                 # other pointers than the announced ones: fine after a restart
    f.prefetch_yuv420_host_prepared(pb)
    f.prefetch_cancel()
    f.apply_yuv420_host_prepared(pa, 2, p0)                 # ... and after a cancel
    ctx.sync()
    # the reused-buffer case: announce b, restart, rewrite b, push b -- the emitted frame (delay 1, identity motion on flat frames) is the NEW b
    f.restart()
    f.prefetch_yuv420_host_prepared(pb)
    f.restart()
    b[0][...] = tex
    out, ts = f.apply_yuv420_host_prepared(pb, 10, p0)
    assert out is None
    out, ts = f.apply_yuv420_host_prepared(pa, 11, p1)
    ctx.sync()
    assert out is not None and ts == 10
    inner = (slice(40, rows - 40), slice(40, cols - 40))
    got = o1[0][inner].astype(int)
    # (identity warp: the resampler returns the centre pixel up to its x 255 truncation -- a stale upload would be the flat 200 of before)
    assert np.abs(got - tex[inner].astype(int)).max() <= 3, "the frame pushed after the restart carries the pre-restart upload"
    assert np.abs(got - 200).mean() > 30
    f.close()


def test_pageable_planes_are_refused_with_a_message(ctx):
    """The host entry points hand their plane pointers to copy engines and to a kernel: pageable memory is an argument error, not a GPU fault."""
    import livevisionkit_amd as lvk
    rows, cols = 360, 640
    f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
    good, out = f.host_planes(rows, cols), f.host_planes(rows, cols)
    bad = tuple(np.full(p.shape, 90, np.uint8) for p in good)               # plain numpy arrays: pageable
    for p in good:
        p[...] = 90
    pg, pb, po = (f.prepare_yuv420_host(p) for p in (good, bad, out))
    with pytest.raises(Exception, match="PINNED"):
        f.apply_yuv420_host_prepared(pb, 0, po)
    with pytest.raises(Exception, match="PINNED"):
        f.prefetch_yuv420_host_prepared(pb)
    with pytest.raises(Exception, match="PINNED"):
        f.apply_yuv420_host_prepared(pg, 0, pb)
    f.apply_yuv420_host_prepared(pg, 0, po)
    ctx.sync()
    f.close()


def test_host_and_device_pushes_mixed_in_one_stream(ctx):
    """lvk_hip.h: the host entry point shares the frame queue with lvk_hip_stab_push_yuv420, 'the two may be mixed'.  Even frames through the
    host entry (pinned planes in and out), odd frames through the device entry, free running; every emitted frame equals the all-device run."""
    import torch
    import livevisionkit_amd as lvk
    rows, cols, n = 720, 1280, 18
    clip = clipgen.Clip(rows, cols, n, device="cuda")
    planes = [clip.render_i420(i) for i in range(n)]
    torch.cuda.synchronize()
    s = lvk.StabilizationFilterSettings.obs_preset("homography", strict=False, predictive_samples=3)

    def run(mixed):
        f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); f.configure(s); f.set_overlap(True)
        hin = [f.host_planes(rows, cols) for _ in range(n)]; hout = [f.host_planes(rows, cols) for _ in range(n)]
        for i in range(n):
            for d, p in zip(hin[i], planes[i]):
                d[...] = p.cpu().numpy()
        dout = [tuple(torch.empty_like(p) for p in planes[0]) for _ in range(n)]
        emitted = []
        for i in range(n):
            if mixed and i % 2 == 0:
                got, _ = f.apply_yuv420_host_prepared(f.prepare_yuv420_host(hin[i]), i, f.prepare_yuv420_host(hout[i]))
                emitted.append(None if got is None else ("host", i))
            else:
                got, _ = f.apply_yuv420(planes[i], timestamp=i, out=dout[i])
                emitted.append(None if got is None else ("dev", i))
        ctx.sync()
        outs = [None if e is None else ([np.array(p) for p in hout[e[1]]] if e[0] == "host" else [p.cpu().numpy() for p in dout[e[1]]]) for e in emitted]
        f.close()
        return outs

    a, b = run(False), run(True)
    assert sum(o is not None for o in a) == n - 3
    for i, (x, y) in enumerate(zip(a, b)):
        assert (x is None) == (y is None), i
        if x is not None:
            for p, q in zip(x, y):
                assert np.array_equal(p, q), i


def test_pinned_planes_sit_on_the_gpus_numa_node(ctx):
    """lvk_hip_host_malloc makes its context's device current, and the runtime takes pinned host memory from the memory pool of the CPU agent
    nearest to that device: the planes of a host-fed stream land on the socket the GPU hangs off without the caller binding anything
    (eight host-fed 4K streams move ~80 GB/s per GPU through host DRAM).  Checked against /proc/self/numa_maps where the
    host has more than one node; reported either way."""
    import os
    import livevisionkit_amd as lvk
    from livevisionkit_amd import shard
    f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
    planes = f.host_planes(2160, 3840)
    for p in planes:
        p[...] = 7                                                          # touched
    node_gpu = shard.gpu_numa_node(0)
    node_mem = shard.numa_node_of_address(planes[0].ctypes.data)
    nodes = [d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()] if os.path.isdir("/sys/devices/system/node") else []
    print(f"\n[numa] GPU 0 on node {node_gpu}, pinned 4K I420 frame on node {node_mem}, host has {len(nodes)} node(s)")
    if len(nodes) > 1 and node_gpu >= 0 and node_mem >= 0:
        assert node_mem == node_gpu, (node_mem, node_gpu)
    f.close()
