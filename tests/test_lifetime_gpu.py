"""Object-lifetime contracts of the C-ABI (round-2 ADVICE): the memory pool refuses a double free and routes a block freed through another
context back to its owner; a caller-owned bulk context may be destroyed once the overlap has been switched off; a 4:2:0 push that
ends before the tracker's synchronisation (first frame) has consumed the caller's planes on return."""
import ctypes as _c

import numpy as np
import pytest

from tests import oracle_lib, synth

pytestmark = pytest.mark.gpu


def test_pool_double_free_and_cross_context_free(ctx):
    import livevisionkit_amd as lvk
    L = ctx.lib
    other = lvk.Context(0)
    p = _c.c_void_p(); q = _c.c_void_p()
    assert L.lvk_hip_malloc(ctx.handle, 1 << 20, _c.byref(p)) == 0
    assert L.lvk_hip_free(ctx.handle, p) == 0
    assert L.lvk_hip_free(ctx.handle, p) != 0                      # second free of a pooled block: refused, the pool keeps ONE copy
    assert b"twice" in L.lvk_hip_last_error(ctx.handle)
    a = _c.c_void_p(); b = _c.c_void_p()
    assert L.lvk_hip_malloc(ctx.handle, 1 << 20, _c.byref(a)) == 0 and a.value == p.value      # the pooled block comes back once ...
    assert L.lvk_hip_malloc(ctx.handle, 1 << 20, _c.byref(b)) == 0 and b.value != a.value       # ... and only once
    # freed through ANOTHER context: goes back to the pool of the context that allocated it
    assert L.lvk_hip_free(other.handle, a) == 0
    assert L.lvk_hip_malloc(ctx.handle, 1 << 20, _c.byref(q)) == 0 and q.value == a.value
    assert L.lvk_hip_free(ctx.handle, q) == 0 and L.lvk_hip_free(ctx.handle, b) == 0
    assert L.lvk_hip_trim(ctx.handle) == 0
    other.close()


def test_bulk_context_may_be_destroyed_after_overlap_off(ctx, oracle):
    """lvk_hip.h: lvk_hip_stab_set_bulk_context(st, NULL) = overlap off.  The stabilizer then lets go of the caller's stream: the bulk
    context is destroyed and the filter keeps running (lvk_hip_sync, configure with a stabilize_output toggle, destroy) bit-exactly."""
    import torch
    import livevisionkit_amd as lvk
    L = ctx.lib
    frames, _ = synth.make_clip(360, 640, 14, seed=5, jitter=1.0)
    so = oracle_lib.preset("homography", predictive_samples=2)
    sg = lvk.StabilizationFilterSettings(); _c.memmove(_c.byref(sg), _c.byref(so), _c.sizeof(so))
    ost = oracle_lib.OracleStabilizer(oracle, so)
    gst = lvk.StabilizationFilter(sg, context=ctx)
    bulk = lvk.Context(0, stream=torch.cuda.Stream())
    ctx._check(L.lvk_hip_stab_set_bulk_context(gst.handle, bulk.handle))
    for i, f in enumerate(frames):
        if i == 6:
            ctx._check(L.lvk_hip_stab_set_bulk_context(gst.handle, None))
            bulk.close()                                            # the caller's context (and with it nothing the filter still refers to)
            bulk = None
        if i == 9:
            off = so.__class__(); _c.memmove(_c.byref(off), _c.byref(so), _c.sizeof(so)); off.stabilize_output = 0
            ost.configure(off); sgo = lvk.StabilizationFilterSettings(); _c.memmove(_c.byref(sgo), _c.byref(off), _c.sizeof(off)); gst.configure(sgo)
        want, _ = ost.push(f, ts=i)
        got, _ = gst.apply(torch.from_numpy(f).cuda(), timestamp=i)
        ctx.sync()
        torch.cuda.synchronize()
        assert (want is None) == (got is None), i
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want), i
    ost.close(); gst.close()


def test_first_yuv420_push_has_consumed_its_planes(ctx, oracle):
    """The planes of lvk_hip_stab_push_yuv420 are the caller's again when the call returns -- also for a push that leaves track() before
    its synchronisation (the first frame; a frame with too few features): overwriting them right away must not change what was tracked."""
    import torch
    import livevisionkit_amd as lvk
    frames, _ = synth.make_clip(1080, 1920, 6, seed=21, jitter=1.0)
    so = oracle_lib.preset("homography", predictive_samples=2)
    sg = lvk.StabilizationFilterSettings(); _c.memmove(_c.byref(sg), _c.byref(so), _c.sizeof(so))
    ost = oracle_lib.OracleStabilizer(oracle, so)
    gst = lvk.StabilizationFilter(sg, context=ctx)
    gst.set_overlap(True)
    planes_d = None
    for i, f in enumerate(frames):
        planes = oracle.egress_yuv420(f)
        packed = oracle.ingest_yuv420(*planes)
        want, _ = ost.push(packed, ts=i)
        if planes_d is None:
            planes_d = tuple(torch.from_numpy(p).cuda() for p in planes)
        else:
            for d, p in zip(planes_d, planes):
                d.copy_(torch.from_numpy(p))
        torch.cuda.synchronize()
        got, _ = gst.apply_yuv420(planes_d, timestamp=i)
        # the call has returned: scribble over the planes on another stream at once
        with torch.cuda.stream(torch.cuda.Stream()):
            for d in planes_d:
                d.fill_(17)
        ctx.sync(); torch.cuda.synchronize()
        so_, sg_ = ost.stats(), gst.stats()
        assert (so_.n_detected, so_.n_matched, so_.n_tracked) == (sg_.n_detected, sg_.n_matched, sg_.n_tracked), i
        assert (want is None) == (got is None), i
        if want is not None:
            for a, b in zip(got, oracle.egress_yuv420(want)):
                assert np.array_equal(a.cpu().numpy(), b), i
    ost.close(); gst.close()


def test_staging_slots_survive_the_bulk_stream_they_were_used_on(ctx, oracle):
    """The context's staging slots remember the stream of the kernel that last read them.  A vector-field stabilizer in overlap mode stages
    its meshes for remaps on ITS bulk stream; once it is destroyed, the next users of those slots (here: 20 plain mesh remaps, more than
    the ring has slots) must still be able to wait for them.  (Found in round 3: hipEventSynchronize fails on an event whose stream is gone.)"""
    import torch
    import livevisionkit_amd as lvk
    frames, _ = synth.make_clip(360, 640, 8, seed=3, jitter=1.0)
    s = lvk.StabilizationFilterSettings.obs_preset("field", predictive_samples=2)
    f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); f.configure(s); f.set_overlap(True)
    for i, fr in enumerate(frames):
        f.apply(torch.from_numpy(fr).cuda(), timestamp=i)
    ctx.sync()
    f.close()
    rng = np.random.default_rng(0)
    src = synth.textured_frame(135, 240, seed=1)
    d = torch.from_numpy(src).cuda()
    for _ in range(20):
        mesh = synth.random_mesh(5, 7, rng, amp=0.02)
        got = ctx.remap_mesh(d, mesh, bg=(0, 128, 128), yuv=True)
        ctx.sync()
        assert np.array_equal(got.cpu().numpy(), oracle.remap_mesh(src, mesh, bg=(0, 128, 128), yuv=True))


def test_host_trace_prints_and_changes_nothing(tmp_path):
    """LVK_HIP_HOST_TRACE=1: the per-phase host times of a push are printed when the stabilizer is destroyed,
    and the emitted planes are those of a run without it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import sys, zlib, numpy as np, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "import livevisionkit_amd as lvk\n"
        "from tests import synth\n"
        "frames, _ = synth.make_clip(270, 480, 12, seed=5, jitter=1.0)\n"
        "ctx = lvk.Context(0)\n"
        "f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)\n"
        "f.configure(lvk.StabilizationFilterSettings.obs_preset('homography', strict=False, predictive_samples=2)); f.set_overlap(True)\n"
        "crc = 0\n"
        "for i, fr in enumerate(frames):\n"
        "    y = torch.from_numpy(np.ascontiguousarray(fr[..., 0])).cuda(); u = torch.from_numpy(np.ascontiguousarray(fr[::2, ::2, 1])).cuda(); v = torch.from_numpy(np.ascontiguousarray(fr[::2, ::2, 2])).cuda()\n"
        "    out, _ = f.apply_yuv420((y, u, v), timestamp=i)\n"
        "    ctx.sync()\n"
        "    if out is not None:\n"
        "        for p in out: crc = zlib.crc32(p.cpu().numpy().tobytes(), crc)\n"
        "f.close(); ctx.close(); print('crc', crc)\n")
    outs = {}
    for trace in (False, True):
        env = dict(os.environ)
        env.pop("LVK_HIP_HOST_TRACE", None)
        if trace:
            env["LVK_HIP_HOST_TRACE"] = "1"
        p = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[trace] = (p.stdout.strip().splitlines()[-1], p.stderr)
    assert outs[False][0] == outs[True][0] and outs[True][0].startswith("crc ")
    assert "[lvk host trace]" in outs[True][1] and "frames" in outs[True][1]
    assert "[lvk host trace]" not in outs[False][1]


def test_schedule_mode_follows_the_caller_with_one_push_of_grace(ctx):
    """lvk_hip_stab_schedule_counters makes the per-push schedule choice visible (round-5 VERDICT weak #6): back-to-back pushes are taken as a
    free-running caller's (persistent remap grid, event wait), pushes with a synchronisation in front of each as a waiting caller's (full grid, host
    signal word) -- and after a free-running streak ONE synchronisation does not switch the mode, the second push in a row that looks synchronous does."""
    import torch
    import livevisionkit_amd as lvk
    small, _ = synth.make_clip(270, 480, 16, seed=5, jitter=1.0)
    frames = small.repeat(4, axis=1).repeat(4, axis=2)                     # 1080p: the remap (~25 us) is still running when the next free-running push begins
    f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
    f.configure(lvk.StabilizationFilterSettings.obs_preset("homography", strict=False, predictive_samples=2)); f.set_overlap(True)
    planes = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (fr[..., 0], fr[::2, ::2, 1], fr[::2, ::2, 2])) for fr in frames]
    args = [f.prepare_yuv420(p) for p in planes]
    outs = [f.prepare_yuv420(tuple(torch.empty_like(p) for p in planes[0])) for _ in range(4)]
    torch.cuda.synchronize()
    k = [0]

    def push():
        f.apply_yuv420_prepared(args[k[0] % 16 if (k[0] // 16) % 2 == 0 else 15 - k[0] % 16], k[0], outs[k[0] & 3]); k[0] += 1
    # (the very first pushes may find the freshly made bulk stream still busy with its own set-up and count as free-running, and the push after a
    #  free-running one keeps that mode by design -- and until the frame delay has filled a push emits nothing, so the synchronisation in front of the
    #  next one returns at once and that push begins within the 15 us that mark a free-running caller: eight pushes settle it before anything is counted)
    for _ in range(8):
        ctx.sync(); push()
    ctx.sync()
    f.schedule_counters(reset=True)
    for _ in range(6):                                                    # a caller that waits for every frame
        ctx.sync(); push()
    ctx.sync()
    c = f.schedule_counters(reset=True)
    assert c["push_synchronised"] == 6 and c["push_free_running"] == 0 and c["remap_persistent"] == 0, c
    for _ in range(60):                                                   # free-running
        push()
    c = f.schedule_counters(reset=True)
    assert c["push_free_running"] >= 50 and c["remap_persistent"] >= 48, c
    for _ in range(12):                                                   # (a solid streak right in front of the synchronisation)
        push()
    c = f.schedule_counters(reset=True)
    assert c["push_free_running"] == 12, c
    ctx.sync(); push()                                                    # ONE synchronisation: still a free-running caller's push
    c = f.schedule_counters(reset=True)
    assert c["push_free_running"] == 1 and c["push_synchronised"] == 0, c
    ctx.sync(); push()                                                    # the second in a row: a waiting caller
    ctx.sync(); push()
    c = f.schedule_counters(reset=True)
    assert c["push_synchronised"] == 2 and c["push_free_running"] == 0 and c["wait_signal_word"] + c["wait_word_timeout"] >= 1, c
    ctx.sync(); f.close()


def test_signal_word_timeout_falls_back_to_the_stream_wait():
    """A caller that waits for every frame takes the chain's completion from a word in host memory (csrc/stabilizer.hip); the spin on it is
    bounded (LVK_HIP_SIGNAL_SPIN_US) and a word that does not arrive falls back to the wait that blocks in the runtime (round-5 ADVICE).  Three
    runs of the same synchronised stream -- default, LVK_HIP_SIGNAL_TEST_LOSE=1 (the word never arrives: EVERY push times out after 50 us and
    takes the fall-back), LVK_HIP_SIGNAL_SPIN_US=0 (never spin) -- must emit the same bytes, and the schedule counters must say which wait
    each run's pushes took."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import sys, json, zlib, numpy as np, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "import livevisionkit_amd as lvk\n"
        "from tests import synth\n"
        "frames, _ = synth.make_clip(270, 480, 16, seed=5, jitter=1.0)\n"
        "ctx = lvk.Context(0)\n"
        "f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)\n"
        "f.configure(lvk.StabilizationFilterSettings.obs_preset('homography', strict=False, predictive_samples=2)); f.set_overlap(True)\n"
        "crc = 0\n"
        "for i, fr in enumerate(frames):\n"
        "    y = torch.from_numpy(np.ascontiguousarray(fr[..., 0])).cuda(); u = torch.from_numpy(np.ascontiguousarray(fr[::2, ::2, 1])).cuda(); v = torch.from_numpy(np.ascontiguousarray(fr[::2, ::2, 2])).cuda()\n"
        "    torch.cuda.synchronize()\n"
        "    out, _ = f.apply_yuv420((y, u, v), timestamp=i)\n"
        "    ctx.sync()\n"
        "    if out is not None:\n"
        "        for p in out: crc = zlib.crc32(p.cpu().numpy().tobytes(), crc)\n"
        "st = f.stats(); c = f.schedule_counters()\n"
        "f.close(); ctx.close(); print(json.dumps({'crc': crc, 'trust': st.trust, 'matched': st.n_matched, 'sched': c}))\n")
    res = {}
    for name, over in (("default", {}), ("lost", {"LVK_HIP_SIGNAL_TEST_LOSE": "1", "LVK_HIP_SIGNAL_SPIN_US": "50"}), ("nospin", {"LVK_HIP_SIGNAL_SPIN_US": "0"})):
        env = dict(os.environ)
        for k in ("LVK_HIP_SIGNAL_TEST_LOSE", "LVK_HIP_SIGNAL_SPIN_US"):
            env.pop(k, None)
        env.update(over)
        p = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["default"]["crc"] == res["lost"]["crc"] == res["nospin"]["crc"] and res["default"]["matched"] > 100
    d, lost, nospin = (res[k]["sched"] for k in ("default", "lost", "nospin"))
    assert d["push_synchronised"] >= 10 and d["wait_signal_word"] >= 10 and d["wait_word_timeout"] == 0, d       # the plugin's pattern: the word
    assert lost["wait_signal_word"] == 0 and lost["wait_word_timeout"] >= 10 and lost["wait_event"] >= lost["wait_word_timeout"], lost
    assert nospin["wait_event"] >= 10, nospin


def test_cross_context_free_fences_the_callers_side_streams(ctx):
    """A block freed through ANOTHER context goes back to its owner's pool only when nothing the caller has in flight can still touch it -- on the
    caller's own stream and on the bulk / transfer streams of its stabilizers (csrc/ctx.hip lvk_hip_free; round-4 VERDICT weak #10): the output
    planes of a free-running overlap-mode filter on context B live in a block of context A, are freed through B right after the last push, and
    must hold the last frame's bytes when A hands the block out again."""
    import torch
    import livevisionkit_amd as lvk
    L = ctx.lib
    rows, cols, n = 2160, 3840, 12                                # 4K: the last remap is still running (~85 us) when the free arrives (~10 us later)
    frames, _ = synth.make_clip(rows // 4, cols // 4, n, seed=91, jitter=1.0)
    frames = np.ascontiguousarray(frames.repeat(4, axis=1).repeat(4, axis=2))
    other = lvk.Context(0, stream=torch.cuda.Stream())
    f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=other)
    f.configure(lvk.StabilizationFilterSettings.obs_preset("homography", strict=False, predictive_samples=2)); f.set_overlap(True)
    # reference: the same stream with ordinary torch outputs
    g = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=other)
    g.configure(lvk.StabilizationFilterSettings.obs_preset("homography", strict=False, predictive_samples=2)); g.set_overlap(True)
    nbytes = rows * cols * 3 // 2
    blk = _c.c_void_p()
    assert L.lvk_hip_malloc(ctx.handle, nbytes, _c.byref(blk)) == 0            # owned by `ctx` (context A)
    out_t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    def planes_at(base):
        # (y, u, v) views of one block: a tensor over foreign memory through the cuda array interface
        class _Wrap:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": "|u1", "data": (ptr, False), "version": 2}
        y = torch.as_tensor(_Wrap(base, (rows, cols)), device="cuda")
        u = torch.as_tensor(_Wrap(base + rows * cols, (rows // 2, cols // 2)), device="cuda")
        v = torch.as_tensor(_Wrap(base + rows * cols + rows * cols // 4, (rows // 2, cols // 2)), device="cuda")
        return y, u, v
    got_planes = planes_at(blk.value)
    want_planes = planes_at(out_t.data_ptr())
    ins = []
    for fr in frames:
        ins.append((torch.from_numpy(np.ascontiguousarray(fr[..., 0])).cuda(), torch.from_numpy(np.ascontiguousarray(fr[::2, ::2, 1])).cuda(),
                    torch.from_numpy(np.ascontiguousarray(fr[::2, ::2, 2])).cuda()))
    torch.cuda.synchronize()
    for i in range(n):
        g.apply_yuv420(ins[i], timestamp=i, out=want_planes)
    other.sync()
    want = out_t.cpu().numpy().copy()
    for i in range(n):
        f.apply_yuv420(ins[i], timestamp=i, out=got_planes)                     # free running; the last remap is still in flight on B's bulk stream
    assert L.lvk_hip_free(other.handle, blk) == 0                               # freed through B: must wait for B's bulk stream
    again = _c.c_void_p()
    assert L.lvk_hip_malloc(ctx.handle, nbytes, _c.byref(again)) == 0 and again.value == blk.value      # back in A's pool
    host = np.empty(nbytes, np.uint8)
    assert L.lvk_hip_download(ctx.handle, host.ctypes.data_as(_c.c_void_p), again, nbytes) == 0
    ctx.sync()
    assert np.array_equal(host, want), "the block went back to its owner's pool while the caller's bulk stream was still writing it"
    assert L.lvk_hip_free(ctx.handle, again) == 0
    f.close(); g.close(); other.close()
