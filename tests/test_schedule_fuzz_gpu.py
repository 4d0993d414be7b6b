"""Seeded schedule fuzz + memory soak of ONE stabilizer instance, under the driver's eyes (distilled from scripts/fuzz_overlap.py and
scripts/soak_probe.py, whose evidence used to live in commit messages only).

The fuzz: one 4:2:0 stream whose pushes are drawn at random from everything a host may do between two frames of the same filter --
  * device-resident planes (lvk_hip_stab_push_yuv420) or pinned host planes (lvk_hip_stab_push_yuv420_host), interleaved;
  * the next frame announced (lvk_hip_stab_prefetch_yuv420 / _yuv420_host), announced WRONGLY (another frame's planes), or announced and
    cancelled (lvk_hip_stab_prefetch_cancel); a wrong host announcement must be refused by the push and leave the filter usable;
  * overlap mode switched on / off, restart(), reconfigure (frame delay up and down, homography <-> vector-field preset, stabilize_output off
    and on again = the delay-only passthrough), the fused lens pre-warp switched on / off (restarts the filter);
  * one resolution change in the middle of the stream;
the pushes free-running (no synchronisation between them beyond what the calls do themselves), every output in a buffer of its own.
After the last push every emitted plane is compared, BY TIMESTAMP, with the oracle chain ingest_yuv420 -> OracleStabilizer ->
egress_yuv420 driven through the same restarts / reconfigurations; the set of emitted timestamps must be the oracle's -- the frames
that were still queued at the old size when the resolution changed included (they leave at their own size, stabilizer.hip ensure_pool).  Reference behaviour this pins: Filters/StabilizationFilter.cpp:42-65,69-135,139-144 (configure / filter / restart on a
live stream), Modules/OBS-Plugin/Interop/VisionFilter.cpp:151-212 (frames arriving from whatever thread and memory the host has).

The soak: 3 000 free-running pushes through the device entry point with restarts, reconfigurations and announcements in between, then the
device memory in use must be what it was before the filter existed (every pool slot, staging plane, event and stream given back)."""
import ctypes

import numpy as np
import pytest

from tests import oracle_lib, synth

pytestmark = pytest.mark.gpu

SIZE_A, SIZE_B = (432, 768), (360, 640)


def _conv(o):
    import livevisionkit_amd as lvk
    s = lvk.StabilizationFilterSettings()
    ctypes.memmove(ctypes.byref(s), ctypes.byref(o), ctypes.sizeof(o))
    return s


def _settings(preset, delay, stabilize=True):
    # relaxed quality assurance: the trust factor leaves zero a few frames after every (re)start, so the emitted planes carry the warp
    return oracle_lib.preset(preset, predictive_samples=delay, min_scene_quality=0.3, min_tracking_quality=0.2, stabilize_output=1 if stabilize else 0)


# LVK_FUZZ_SEEDS="5-40" (or "7,9,11"): more seeds for a one-off sweep on a GPU box; the suite itself runs six
def _seeds():
    import os
    spec = os.environ.get("LVK_FUZZ_SEEDS")
    if not spec:
        return [1, 2, 3, 4, 5, 6]
    out = []
    for part in spec.split(","):
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


@pytest.mark.parametrize("seed", _seeds())
def test_schedule_fuzz_one_stream_against_the_oracle(ctx, oracle, seed, monkeypatch):
    import torch
    import livevisionkit_amd as lvk
    if seed & 4:
        # frames this small never look free-running to the library (the host's turn outlasts their remap): the seeds with bit 2 set pin the free-running
        # schedule -- persistent remap grids, conversions behind the chain on the tracking stream -- (the variable is read when the filter is created)
        monkeypatch.setenv("LVK_HIP_ASSUME_CALLER", "free")
    rng = np.random.default_rng(9000 + seed)
    n = 44
    change_at = int(rng.integers(18, 26))                        # first push of the second frame size
    clip_a, _ = synth.make_clip(*SIZE_A, n, seed=40 + seed, jitter=1.0)
    clip_b, _ = synth.make_clip(*SIZE_B, n, seed=40 + seed, jitter=1.0)
    frames = [clip_a[i] if i < change_at else clip_b[i] for i in range(n)]
    nv12 = bool(seed & 1)
    planes_h = [oracle.egress_yuv420(f, nv12=nv12) for f in frames]
    planes_d = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pl) for pl in planes_h]
    torch.cuda.synchronize()

    preset, delay = ("homography", "field")[seed % 2], 2
    s = _settings(preset, delay)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(_conv(s))
    overlap, stabilize, lens_on, stab_back_at = True, True, False, 0
    gst.set_overlap(overlap)
    dev_args = [gst.prepare_yuv420(p) for p in planes_d]

    def host_copy(i):
        hp = gst.host_planes(*frames[i].shape[:2], nv12)
        for d, p in zip(hp, planes_h[i]):
            d[...] = p
        return hp

    entries = [("device", "host")[int(rng.integers(0, 3)) == 0] for _ in range(n)]      # one push in three from host memory
    host_in = {i: host_copy(i) for i in range(n) if entries[i] == "host"}
    host_in_args = {i: gst.prepare_yuv420_host(p) for i, p in host_in.items()}
    want, want_push, got, log = {}, {}, {}, []
    live = 0                                                       # emitted frames whose warp carried the tracker's estimate (trust > 0)
    host_announced = None                                          # index of a host frame whose upload is under way
    refused = 0
    big = np.zeros((max(SIZE_A[0], SIZE_B[0]), max(SIZE_A[1], SIZE_B[1]), 3), np.uint8)
    for i in range(n):
        # ---- what happens BETWEEN two pushes
        ev = int(rng.integers(0, 20))
        if not stabilize and i >= stab_back_at:
            ev = 4                                                   # the passthrough lasts four pushes, then the stream is stabilized again
        if ev == 0 and i > 4:
            ost.restart(); gst.restart(); host_announced = None; log.append((i, "restart"))
        elif ev == 1:
            overlap = not overlap; gst.set_overlap(overlap); log.append((i, "overlap %d" % overlap))
        elif ev == 2:
            delay = 3 if delay == 2 else 2
            s = _settings(preset, delay, stabilize); ost.configure(s); gst.configure(_conv(s)); log.append((i, "delay %d" % delay))
        elif ev == 3 and i > 6:
            preset = "field" if preset == "homography" else "homography"
            s = _settings(preset, delay, stabilize); ost.configure(s); gst.configure(_conv(s)); log.append((i, preset))
        elif ev == 4 and i > 3:
            # stabilize_output off = the delay-only passthrough (StabilizationFilter.cpp:77-95; switching it off resets tracker and smoother, :49-52)
            stabilize = not stabilize; stab_back_at = i + 4
            s = _settings(preset, delay, stabilize); ost.configure(s); gst.configure(_conv(s)); log.append((i, "stabilize %d" % stabilize))
        elif ev == 5 and i > 8:
            # the fused lens pre-warp switched on / off in the middle of the stream: both sides restart (lvk_hip_stab_set_lens)
            lens_on = not lens_on
            r, c = frames[i].shape[:2]
            prof = np.array([0.8 * c, 0.8 * c, c / 2.0, r / 2.0, -0.12, 0.03, 0.0, 0.0, 0.0]) if lens_on else None
            ost.set_lens(prof); gst.set_lens(prof); host_announced = None; log.append((i, "lens %d" % lens_on))
        # ---- the oracle's push
        buf = big.copy()
        w, wts = ost.push(oracle.ingest_yuv420(*planes_h[i]), ts=i, nthreads=32, out=buf)
        if w is not None:
            r, c = frames[wts].shape[:2]
            want[wts] = oracle.egress_yuv420(np.ascontiguousarray(buf[:r, :c]), nv12=nv12); want_push[wts] = i
            live += 1 if ost.stats().trust > 0.05 else 0
        # ---- announcements for THIS push's successor, then the push
        nxt = i + 1 if i + 1 < n else None
        a = int(rng.integers(0, 6))
        if entries[i] == "device":
            if host_announced is not None:                        # a host frame was announced but this push goes through the device entry
                gst.prefetch_cancel(); host_announced = None
            if nxt is not None and a == 0:
                gst.prefetch_yuv420_prepared(dev_args[nxt])                                   # right
            elif nxt is not None and a == 1:
                gst.prefetch_yuv420_prepared(dev_args[(i + 3) % n])                           # wrong planes (maybe a wrong size as well)
            elif nxt is not None and a == 2:
                gst.prefetch_yuv420_prepared(dev_args[nxt]); gst.prefetch_cancel()            # announced, then cancelled
            g, gts = gst.apply_yuv420(planes_d[i], timestamp=i)
            if g is not None:
                got[gts] = ("device", g)
        else:
            due = gst.next_output(*frames[i].shape[:2])                   # the emitted frame has the DELAYED frame's size
            out = gst.host_planes(*(due[:2] if due else frames[i].shape[:2]), nv12)
            out_args = gst.prepare_yuv420_host(out)
            if host_announced is not None and host_announced != i:
                # a WRONG announcement is outstanding: the push must be refused, and a cancel makes the filter usable again
                with pytest.raises(lvk.LvkHipError):
                    gst.apply_yuv420_host_prepared(host_in_args[i], i, out_args)
                refused += 1
                gst.prefetch_cancel(); host_announced = None
            g, _ = gst.apply_yuv420_host_prepared(host_in_args[i], i, out_args)
            host_announced = None
            if g is not None:
                got[gst._ots.value] = ("host", out)
            if nxt is not None and entries[nxt] == "host" and a <= 2:
                later = [k for k in host_in if k > nxt]
                if a == 2 and later and frames[later[0]].shape == frames[nxt].shape:
                    gst.prefetch_yuv420_host_prepared(host_in_args[later[0]]); host_announced = later[0]      # wrong frame announced
                elif a == 1:
                    gst.prefetch_yuv420_host_prepared(host_in_args[nxt]); gst.prefetch_cancel()
                else:
                    gst.prefetch_yuv420_host_prepared(host_in_args[nxt]); host_announced = nxt
    ctx.sync()
    so, sg = ost.stats(), gst.stats()
    assert (so.n_detected, so.n_matched, so.n_tracked, so.tracking_stability, so.trust) == (sg.n_detected, sg.n_matched, sg.n_tracked, sg.tracking_stability, sg.trust), log
    # frames of the old size still queued when the size changed leave at their own size, as in the oracle and the reference (rounds 2-5 dropped them)
    late = {ts for ts, at in want_push.items() if ts < change_at <= at}
    assert sorted(got) == sorted(want), (log, sorted(late))
    if seed & 4:
        c = gst.schedule_counters()
        assert c["push_free_running"] > 0 and c["push_synchronised"] == 0, c
    assert len(got) >= 10, (len(got), log)                           # (schedules that restart every few pushes emit little: the suite's seeds emit 25-35)
    assert live >= 4, f"only {live} emitted frames had a trust factor above zero: the schedule restarts too often to test the warp ({log})"
    for ts, (kind, planes) in sorted(got.items()):
        for k, (p, q) in enumerate(zip(planes, want[ts])):
            p = p.cpu().numpy() if kind == "device" else np.asarray(p)
            assert p.shape == q.shape, (ts, k)
            if not np.array_equal(p, q):
                d = np.abs(p.astype(np.int32) - q.astype(np.int32))
                raise AssertionError(f"seed {seed}: frame ts {ts} ({kind} push) plane {k}: {int((d > 0).sum())} bytes differ, max |d| {d.max()}; schedule {log}")
    print(f"\n[schedule fuzz seed {seed}] {len(got)} frames compared ({live} of the oracle's with trust > 0), {len(late)} of the old size emitted after the size change (push {change_at}), "
          f"{refused} wrong host announcements refused, {gst.lookahead_frames()} pushes found their pyramid built ahead; events {log}")
    ost.close(); gst.close()


PACKED_SIZES = [(432, 768), (300, 768), (360, 640)]               # same width / fewer rows, then narrower: both ways round in a run


@pytest.mark.parametrize("seed", _seeds())
def test_schedule_fuzz_packed_stream_with_resizes(ctx, oracle, seed):
    """The same fuzz over the PACKED entry (lvk_hip_stab_push: what lvk::StabilizationFilter::filter and with it the plugin's VSFilter call):
    frame size AND format change several times in the middle of the stream, overlap toggles, restarts, the frame delay goes up and down, the
    delay-only passthrough, the fused lens pre-warp, device look-ahead right / wrong / cancelled, and now and then an output buffer sized from
    the INCOMING frame -- refused when the delayed frame is larger, before anything changes.  Free-running; every output in a guarded buffer of
    its own.  The reference's behaviour is the bar (StabilizationFilter.cpp:118-131, WarpMesh.cpp:183-223, Image.cpp:53,116): EVERY frame
    leaves, at its own size and format, equal to the oracle's frame of the same timestamp, and no byte outside rows x cols x 3 is written."""
    import torch
    import livevisionkit_amd as lvk
    rng = np.random.default_rng(7000 + seed)
    n = 40
    clips = [synth.make_clip(r, c, n, seed=60 + seed, jitter=1.0)[0] for r, c in PACKED_SIZES]
    size_k, fmt = 0, 4
    preset, delay, overlap, stabilize, lens_on, stab_back_at = "homography", 2, bool(seed & 1), True, False, 0
    s = _settings(preset, delay)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(_conv(s))
    gst.set_overlap(overlap)
    GUARD = 0x5A

    def guarded(rows, cols):
        pitch = cols * 3 + 64
        buf = torch.full((rows + 8, pitch), GUARD, dtype=torch.uint8, device="cuda")
        return buf, buf.as_strided((rows, cols, 3), (pitch, 3, 1), 4 * pitch)

    # the schedule is drawn first (the look-ahead needs to know the NEXT frame): events[i] happens before push i
    events, frames, fmts = [], [], []
    for i in range(n):
        ev = int(rng.integers(0, 16))
        if ev in (6, 7, 8) and i > 2:
            size_k = (size_k + 1 + int(rng.integers(0, 2))) % 3
        elif ev == 9:
            fmt = (4, 0, 2)[int(rng.integers(0, 3))]
        f = clips[size_k][i]
        if fmt != 4:
            f = f[..., [1, 0, 2]]                                     # the textured channel where the grey value weighs most
        events.append(ev); frames.append(np.ascontiguousarray(f)); fmts.append(fmt)
    if sum(frames[i].shape != frames[i - 1].shape for i in range(1, n)) < 2:
        # a draw with fewer than two size changes: two are put in (pushes 12-23 at the second size, 30.. at the third), whatever else the schedule does
        for i in range(12, n):
            k = 1 if i < 24 else (2 if i >= 30 else 0)
            f = clips[k][i]
            frames[i] = np.ascontiguousarray(f if fmts[i] == 4 else f[..., [1, 0, 2]])
    dev = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    want, got, log = {}, {}, []
    refused = live = resizes = ahead_right = 0
    big = np.zeros((max(r for r, _ in PACKED_SIZES), max(c for _, c in PACKED_SIZES), 3), np.uint8)
    for i in range(n):
        ev = events[i]
        if not stabilize and i >= stab_back_at:
            ev = 4
        if ev == 0 and i > 4:
            ost.restart(); gst.restart(); log.append((i, "restart"))
        elif ev == 1:
            overlap = not overlap; gst.set_overlap(overlap); log.append((i, "overlap %d" % overlap))
        elif ev == 2:
            delay = 3 if delay == 2 else 2
            s = _settings(preset, delay, stabilize); ost.configure(s); gst.configure(_conv(s)); log.append((i, "delay %d" % delay))
        elif ev == 3 and i > 6:
            preset = "field" if preset == "homography" else "homography"
            s = _settings(preset, delay, stabilize); ost.configure(s); gst.configure(_conv(s)); log.append((i, preset))
        elif ev == 4 and i > 3:
            stabilize = not stabilize; stab_back_at = i + 4
            s = _settings(preset, delay, stabilize); ost.configure(s); gst.configure(_conv(s)); log.append((i, "stabilize %d" % stabilize))
        elif ev == 5 and i > 8:
            lens_on = not lens_on
            r, c = PACKED_SIZES[0]
            prof = np.array([0.8 * c, 0.8 * c, c / 2.0, r / 2.0, -0.12, 0.03, 0.0, 0.0, 0.0]) if lens_on else None
            ost.set_lens(prof); gst.set_lens(prof); log.append((i, "lens %d" % lens_on))
        f, fmt, d = frames[i], fmts[i], dev[i]
        if i > 0 and (f.shape != frames[i - 1].shape or fmt != fmts[i - 1]):
            resizes += f.shape != frames[i - 1].shape; log.append((i, "%dx%d format %d" % (f.shape[1], f.shape[0], fmt)))
        buf = big.copy()
        w, wts = ost.push(f, ts=i, fmt=fmt, nthreads=32, out=buf)
        if w is not None:
            r, c = frames[wts].shape[:2]
            want[wts] = buf[:r, :c].copy()
            live += 1 if ost.stats().trust > 0.05 else 0
        # ---- the look-ahead of the packed entry (lvk_hip_stab_prefetch): the next frame announced rightly, wrongly, or announced and cancelled
        a = int(rng.integers(0, 8))
        nxt = i + 1 if i + 1 < n else None
        if nxt is not None and a == 0:
            gst.prefetch(dev[nxt], fmt=fmts[nxt]); ahead_right += 1
        elif nxt is not None and a == 1:
            gst.prefetch(dev[(i + 3) % n], fmt=fmts[(i + 3) % n])
        elif nxt is not None and a == 2:
            gst.prefetch(dev[nxt], fmt=fmts[nxt]); gst.prefetch_cancel()
        due = gst.next_output(f.shape[0], f.shape[1], fmt)
        assert (due is None) == (w is None), (i, log)
        if due is not None and (due[0] > f.shape[0] or due[1] > f.shape[1]) and a < 5:
            with pytest.raises(lvk.LvkHipError):                      # an output sized from the incoming frame: refused, nothing changes
                gst.apply(d, timestamp=i, out=guarded(*f.shape[:2])[1], fmt=fmt)
            refused += 1
        if due is not None:
            gb, view = guarded(due[0], due[1])
            g, gts = gst.apply(d, timestamp=i, out=view, fmt=fmt)
            assert g is not None and tuple(g.shape) == (due[0], due[1], 3) and gst.last_format == fmts[gts] == due[2], (i, log)
            got[gts] = (gb, g)
        else:
            g, _ = gst.apply(d, timestamp=i, fmt=fmt)
            assert g is None, (i, log)
    ctx.sync()
    so, sg = ost.stats(), gst.stats()
    assert (so.n_detected, so.n_matched, so.n_tracked, so.tracking_stability, so.trust) == (sg.n_detected, sg.n_matched, sg.n_tracked, sg.tracking_stability, sg.trust), log
    assert sorted(got) == sorted(want), log                            # every frame the oracle emits, old sizes included: nothing is dropped
    assert len(got) >= 10 and resizes >= 2, (len(got), resizes, log)
    for ts, (gb, g) in sorted(got.items()):
        r, c = frames[ts].shape[:2]
        p, q = g.cpu().numpy(), want[ts]
        assert p.shape == q.shape, (ts, p.shape, q.shape)
        if not np.array_equal(p, q):
            dd = np.abs(p.astype(np.int32) - q.astype(np.int32))
            raise AssertionError(f"seed {seed}: frame ts {ts} ({c}x{r}, format {fmts[ts]}): {int((dd > 0).sum())} bytes differ, max |d| {dd.max()}; schedule {log}")
        b = gb.cpu().numpy()
        assert (b[:4] == GUARD).all() and (b[4 + r:] == GUARD).all() and (b[4:4 + r, c * 3:] == GUARD).all(), f"frame ts {ts}: bytes outside the {c}x{r} output were written; {log}"
    print(f"\n[packed fuzz seed {seed}] {len(got)} frames compared ({live} with trust > 0), {resizes} size changes, {refused} undersized outputs refused, {gst.lookahead_frames()} of {ahead_right} right announcements found their pyramid built; events {log}")
    ost.close(); gst.close()


def test_soak_3000_pushes_give_every_byte_back():
    """3 000 free-running pushes (1080p I420, device entry, overlap mode) with restarts, reconfigurations and right / wrong / missing
    announcements in between.  When the filter and its context are gone the device has its memory back (pool slots, staging planes,
    pyramids, events, streams: lvk_hip_trim accounting), the tracker was live at the end (trust), and a second run of the same schedule
    emits the same last frame bit for bit (nothing in the free-running schedule -- which stream converts, whether a pyramid was built
    ahead -- reaches a pixel; the pixels themselves are held to the oracle by the fuzz above and tests/test_long_run_gpu.py)."""
    import torch
    import livevisionkit_amd as lvk
    from tests import clipgen
    rows, cols, n, m = 1080, 1920, 3000, 32
    clip = clipgen.Clip(rows, cols, m, device="cuda", cut_at=None)
    planes = [clip.render_i420(k) for k in range(m)]
    outs = [tuple(torch.empty_like(p) for p in planes[0]) for _ in range(4)]
    torch.cuda.synchronize()

    def idx(i):                                       # the clip played forwards and backwards: no scene cut where it wraps
        k = i % (2 * m - 2)
        return k if k < m else 2 * m - 2 - k

    def run(count):
        c = lvk.Context(0, stream=torch.cuda.Stream())
        f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=c)
        f.configure(lvk.StabilizationFilterSettings.obs_preset("homography", strict=False, predictive_samples=4)); f.set_overlap(True)
        pa = [f.prepare_yuv420(p) for p in planes]; oa = [f.prepare_yuv420(o) for o in outs]
        for i in range(count):
            if i == count // 3:
                f.configure(lvk.StabilizationFilterSettings.obs_preset("field", strict=False, predictive_samples=4))
                f.configure(lvk.StabilizationFilterSettings.obs_preset("homography", strict=False, predictive_samples=6))
                f.configure(lvk.StabilizationFilterSettings.obs_preset("homography", strict=False, predictive_samples=4))
            if i in (count // 2, count - 200):
                f.restart()
            if i % 97 != 96:                          # announced one push ahead; now and then not, now and then wrongly
                f.prefetch_yuv420_prepared(pa[idx(i + (5 if i % 89 == 88 else 1))])
            f.apply_yuv420_prepared(pa[idx(i)], i, oa[i & 3])
        c.sync()
        last = [p.cpu().numpy().copy() for p in outs[(count - 1) & 3]]
        trust, ahead = f.stats().trust, f.lookahead_frames()
        f.close()
        assert c.lib.lvk_hip_trim(c.handle) == 0
        c.close()
        return last, trust, ahead

    run(60)                                           # the runtime's own pools (streams, events, code objects) warm before measuring
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    last, trust, ahead = run(n)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    again, _, _ = run(n)
    delta = (free0 - free1) / 1e6
    print(f"\n[soak] {n} pushes twice, device memory delta {delta:+.1f} MB, final trust {trust:.2f}, {ahead} pushes found their pyramid built ahead")
    assert trust > 0.5, trust
    assert ahead > n // 2, ahead
    for a, b in zip(last, again):
        assert np.array_equal(a, b), "two runs of the same free-running schedule emitted different last frames"
    assert abs(delta) < 16.0, f"device memory not returned: {delta:+.1f} MB"


OBS_YUV_FORMATS = ["I420", "NV12", "I422", "I42A", "I444", "YUVA", "YUY2", "YVYU", "UYVY", "AYUV", "I40A"]


@pytest.mark.parametrize("seed", _seeds())
def test_schedule_fuzz_obs_formats_against_the_oracle(ctx, oracle, seed):
    """lvk_hip_stab_push_obs, fuzzed: one source whose video format changes from frame to frame (every YUV format FrameIngest::Select accepts -- they
    all become the same packed frame), whose size changes twice, that is restarted, that turns stabilize_output off and on, in either schedule.  Every
    emitted frame -- the DELAYED one, at its own size, converted to the format of the push that emits it -- against the oracle's
    ingest -> filter -> egress; planes that cannot hold it are refused before anything changes."""
    import torch
    import livevisionkit_amd as lvk
    rng = np.random.default_rng(7000 + seed)
    n = 44
    base, _ = synth.make_clip(SIZE_B[0], SIZE_B[1], n, seed=300 + seed, jitter=1.0)
    cuts = sorted(rng.choice(np.arange(8, n - 8), 2, replace=False))
    sizes = [SIZE_B, (288, 512), (240, 400)]
    frames = []
    for i in range(n):
        r, c = sizes[int(i >= cuts[0]) + int(i >= cuts[1])]
        y0, x0 = (SIZE_B[0] - r) // 2, (SIZE_B[1] - c) // 2
        frames.append(np.ascontiguousarray(base[i][y0:y0 + r, x0:x0 + c]))
    delay = int(rng.integers(1, 5))
    s = _settings("homography", delay)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_conv(s), context=ctx)
    gst.set_overlap(bool(seed & 1))
    size_of = {}
    log, emitted, refused = [], 0, 0
    keep = []
    for i, f in enumerate(frames):
        a = int(rng.integers(0, 16))
        if a == 0 and i > 4:
            ost.restart(); gst.restart(); log.append((i, "restart"))
        elif a == 1:
            s.stabilize_output = 1 - s.stabilize_output
            ost.configure(s); gst.configure(_conv(s)); log.append((i, "stabilize_output=%d" % s.stabilize_output))
        fmt = OBS_YUV_FORMATS[int(rng.integers(0, len(OBS_YUV_FORMATS)))]
        size_of[i] = f.shape[:2]
        planes = oracle.egress_obs(fmt, f)
        packed = oracle.ingest_obs(fmt, planes)
        big = np.zeros((SIZE_B[0], SIZE_B[1], 3), np.uint8)
        w, wts = ost.push(packed, ts=i, fmt=4, out=big)
        dev = [torch.from_numpy(p).cuda() for p in planes]; keep.append(dev)
        due = gst.next_output(f.shape[0], f.shape[1], 4)
        assert (due is None) == (w is None), (seed, i, log)
        if due is not None and (due[0] > f.shape[0] or due[1] > f.shape[1]) and int(rng.integers(0, 2)):
            small = [torch.empty_like(p) for p in dev]                        # planes sized from the INCOMING frame: too small for the delayed one
            with pytest.raises(lvk.LvkHipError, match="DELAYED"):
                gst.apply_obs(fmt, dev, timestamp=i, out=small)
            assert gst.next_output(f.shape[0], f.shape[1], 4) == due
            refused += 1
        out, ots = gst.apply_obs(fmt, dev, timestamp=i)
        if int(rng.integers(0, 3)) == 0:
            ctx.sync()                                                        # a caller that sometimes waits for its frame
        assert (out is None) == (w is None), (seed, i, log)
        if out is not None:
            ctx.sync()
            emitted += 1
            r, c = size_of[wts]
            assert ots == wts and tuple(out[0].shape[:2]) == (r, c), (seed, i, fmt, log)
            for g, want in zip(out, oracle.egress_obs(fmt, np.ascontiguousarray(w[:r, :c]))):
                assert np.array_equal(g.cpu().numpy(), want), (seed, i, fmt, log)
    ctx.sync(); gst.close()
    assert emitted >= n - 6 * (delay + 1), (emitted, log)
    print(f"[obs fuzz {seed}] {emitted} frames, delay {delay}, cuts {cuts}, {refused} refusals, events {log}")
