"""GPU parity of lvk_hip_ingest_obs / lvk_hip_egress_obs -- FrameIngest::to_ocl / to_obs for every OBS video format FrameIngest::Select knows
(Modules/OBS-Plugin/Interop/FrameIngest.cpp:36-75,476-753) -- against the oracle, through the C-ABI.  Integer arithmetic: bit-exact.  Sizes from
ragged to 4K; planes with row padding; guard bytes around every output; the stabilizer fed through a non-4:2:0 format end to end."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORMATS = ["I420", "NV12", "YVYU", "YUY2", "UYVY", "RGBA", "BGRA", "BGRX", "Y800", "I444", "BGR3", "I422", "I40A", "I42A", "YUVA", "AYUV"]
SIZES = [(6, 8), (34, 50), (270, 480), (2, 2), (1080, 1920), (66, 258)]


def _gpu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _planes(oracle, fmt, rows, cols, seed=0):
    rng = np.random.default_rng(seed + rows * 7 + cols)
    return [rng.integers(0, 256, sh, dtype=np.uint8) for sh in oracle.obs_plane_shapes(fmt, rows, cols)]


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("fmt", FORMATS)
def test_ingest_bit_exact(ctx, oracle, fmt, size):
    rows, cols = size
    planes = _planes(oracle, fmt, rows, cols)
    want = oracle.ingest_obs(fmt, planes)
    got = ctx.ingest_obs(fmt, [_gpu(p) for p in planes])
    ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("fmt", FORMATS)
def test_egress_bit_exact_and_leaves_the_other_bytes_alone(ctx, oracle, fmt, size):
    import torch
    rows, cols = size
    rng = np.random.default_rng(rows * 3 + cols)
    frame = rng.integers(0, 256, (rows, cols) if fmt == "Y800" else (rows, cols, 3), dtype=np.uint8)
    before = _planes(oracle, fmt, rows, cols, seed=9)                       # what the OBS frame held: the reference overwrites only what it converts
    want = oracle.egress_obs(fmt, frame, planes=[p.copy() for p in before])
    # every plane sits inside a guard of 0xA5 bytes
    guards = []
    dev = []
    for p in before:
        flat = torch.full((p.size + 512,), 0xA5, dtype=torch.uint8, device="cuda")
        flat[256:256 + p.size] = _gpu(p).reshape(-1)
        guards.append(flat)
        dev.append(flat[256:256 + p.size].view(*p.shape))
    ctx.egress_obs(fmt, _gpu(frame), dev)
    ctx.sync()
    for g, d, w in zip(guards, dev, want):
        assert np.array_equal(d.cpu().numpy(), w)
        gg = g.cpu().numpy()
        assert (gg[:256] == 0xA5).all() and (gg[256 + w.size:] == 0xA5).all()


@pytest.mark.parametrize("fmt", ["I422", "YUY2", "UYVY", "I444", "AYUV", "I420", "NV12"])
def test_planes_with_row_padding_and_unaligned_output(ctx, oracle, fmt):
    """linesize > width on the way in (OBS aligns its planes) and an output frame whose rows start at odd addresses (the byte-store path)."""
    import torch
    rows, cols = 38, 52
    planes = _planes(oracle, fmt, rows, cols, seed=4)
    want = oracle.ingest_obs(fmt, planes)
    padded = []
    for p in planes:
        wide = torch.zeros((p.shape[0], p.shape[1] + 13) + p.shape[2:], dtype=torch.uint8, device="cuda")
        wide[:, :p.shape[1]] = _gpu(p)
        padded.append(wide[:, :p.shape[1]])
    # a frame at byte offset 1 of a wider buffer: rows start at odd addresses
    raw = torch.zeros((rows * (cols * 3 + 7) + 1,), dtype=torch.uint8, device="cuda")
    frame = torch.as_strided(raw, (rows, cols, 3), (cols * 3 + 7, 3, 1), 1)
    got = ctx.ingest_obs(fmt, padded, out=frame)
    ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)
    # and back into padded planes
    pad = 4 if fmt == "AYUV" else 13                                          # (the AYUV sink writes dwords: its pitch is a multiple of 4)
    back = [torch.zeros((p.shape[0], p.shape[1] + pad) + p.shape[2:], dtype=torch.uint8, device="cuda") for p in planes]
    views = [b[:, :p.shape[1]] for b, p in zip(back, planes)]
    ctx.egress_obs(fmt, frame, views)
    ctx.sync()
    for v, w in zip(views, oracle.egress_obs(fmt, want)):
        assert np.array_equal(v.cpu().numpy(), w)
    for b, p in zip(back, planes):
        assert (b[:, p.shape[1]:] == 0).all()                                # the padding is not written


def test_4k_formats_and_round_trips(ctx, oracle):
    """3840 x 2160: the lossless formats come back byte for byte; 4:2:2 comes back within the resampling pair's error (<= 1 after the first trip is
    NOT promised by the reference; what is checked is the oracle's bytes)."""
    rows, cols = 2160, 3840
    for fmt in ("I444", "AYUV", "BGR3"):
        planes = _planes(oracle, fmt, rows, cols, seed=2)
        dev = [_gpu(p) for p in planes]
        frame = ctx.ingest_obs(fmt, dev)
        back = [d.clone().zero_() for d in dev]
        ctx.egress_obs(fmt, frame, back)
        ctx.sync()
        for b, p in zip(back, planes):
            got = b.cpu().numpy()
            if fmt == "AYUV":
                assert (got[..., 0] == 255).all() and np.array_equal(got[..., 1:], p[..., 1:])
            else:
                assert np.array_equal(got, p)
    for fmt in ("I422", "YUY2"):
        planes = _planes(oracle, fmt, rows, cols, seed=3)
        frame = ctx.ingest_obs(fmt, [_gpu(p) for p in planes])
        ctx.sync()
        assert np.array_equal(frame.cpu().numpy(), oracle.ingest_obs(fmt, planes))


def test_refusals(ctx):
    import torch
    y = torch.zeros((4, 5), dtype=torch.uint8, device="cuda"); c = torch.zeros((4, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        ctx.ingest_obs("I422", [y, c, c])                                    # odd width
    with pytest.raises(Exception):
        ctx.ingest_obs("I444", [y])                                          # planes missing
    rgba = torch.zeros((4, 6, 4), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        ctx.ingest_obs("RGBA", [rgba[:, :5]])                                # DirectIngest reads a tight byte stream
    assert ctx.lib.lvk_hip_obs_frame_format(0) < 0 and ctx.lib.lvk_hip_obs_frame_format(17) < 0
    assert ctx.obs_frame_format("UYVY") == 4 and ctx.obs_frame_format("Y800") == 5 and ctx.obs_frame_format("RGBA") == 2 and ctx.obs_frame_format("BGRX") == 0


def test_stabilizer_fed_from_packed_422(ctx, oracle):
    """A UYVY stream (what a capture card delivers) through ingest -> lvk_hip_stab_push -> egress equals the oracle's stabilizer fed the oracle's frames."""
    import livevisionkit_amd as lvk
    from tests import oracle_lib, synth
    from tests.test_stabilizer_gpu import _to_settings
    rows, cols, n = 270, 480, 12
    clip, _ = synth.make_clip(rows, cols, n, seed=31, jitter=1.0)
    s = oracle_lib.preset("homography", predictive_samples=3, min_scene_quality=0.3, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    emitted = 0
    for i, f in enumerate(clip):
        raw = oracle.egress_obs("UYVY", f)[0]                                # the capture card's view of the frame
        want_in = oracle.ingest_obs("UYVY", [raw])
        got_in = ctx.ingest_obs("UYVY", [_gpu(raw)])
        assert ctx.obs_frame_format("UYVY") == 4
        w, wts = ost.push(want_in, ts=i, fmt=4)
        frame, ots = gst.apply(got_in, timestamp=i, fmt=4)
        ctx.sync()
        assert (frame is None) == (w is None)
        if frame is not None:
            emitted += 1
            assert ots == wts
            assert np.array_equal(frame.cpu().numpy(), w)
            back = [_gpu(np.zeros_like(raw))]
            ctx.egress_obs("UYVY", frame, back)
            ctx.sync()
            assert np.array_equal(back[0].cpu().numpy(), oracle.egress_obs("UYVY", w)[0])
    assert emitted >= n - 4


# ---- lvk_hip_stab_push_obs: the plugin's asynchronous path in one call, for every format ---------------------------------------------------------
def _obs_stream(oracle, fmt, clip):
    """the clip as an OBS source of `fmt` would deliver it: (planes, the packed frame FrameIngest::to_ocl makes of them) per frame"""
    out = []
    for f in clip:
        planes = oracle.egress_obs(fmt, f)
        out.append((planes, oracle.ingest_obs(fmt, planes)))
    return out


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("fmt,size,preset", [(f, (270, 480), "homography") for f in ("I422", "I444", "YUY2", "YVYU", "UYVY", "AYUV", "BGR3", "I42A", "NV12", "I420")] +
                         [(f, (146, 258), "homography") for f in ("I422", "I444", "UYVY", "YUY2", "AYUV")] +                 # ragged: tails of 2 pixels, rows at odd alignments
                         [(f, (270, 480), "field") for f in ("UYVY", "I444", "I422")])                                       # the mesh kernels' fused sinks
def test_push_obs_equals_oracle_ingest_filter_egress(ctx, oracle, fmt, size, preset, overlap):
    """(the formats with a fused remap + egress kernel -- 4:2:2, 4:4:4, AYUV -- leave through it whenever the remap runs, and through the packed buffer +
    egress kernel on the frames the filter passes through: both routes are in every one of these streams)"""
    import livevisionkit_amd as lvk
    from tests import oracle_lib, synth
    from tests.test_stabilizer_gpu import _to_settings
    (rows, cols), n = size, 12
    clip, _ = synth.make_clip(rows, cols, n, seed=41, jitter=1.0)
    s = oracle_lib.preset(preset, predictive_samples=3, min_scene_quality=0.3, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    gst.set_overlap(overlap)
    ffmt = ctx.obs_frame_format(fmt)
    emitted = 0
    for i, (planes, packed) in enumerate(_obs_stream(oracle, fmt, clip)):
        w, wts = ost.push(packed, ts=i, fmt=ffmt)
        dev = [_gpu(p) for p in planes]
        out, ots = gst.apply_obs(fmt, dev, timestamp=i)
        for d in dev:
            d.fill_(0x33)                                                   # the planes are consumed when the call returns
        ctx.sync()
        assert (out is None) == (w is None), (fmt, i)
        if out is not None:
            emitted += 1
            assert ots == wts
            for g, want in zip(out, oracle.egress_obs(fmt, w)):
                assert np.array_equal(g.cpu().numpy(), want), (fmt, i)
    assert emitted == n - 3
    gst.close()


def test_push_obs_420_is_push_yuv420(ctx, oracle):
    """the 4:2:0 formats through lvk_hip_stab_push_obs take lvk_hip_stab_push_yuv420's route: same bytes, fused remap + egress (no ingest / egress stage of their own)"""
    import livevisionkit_amd as lvk
    from tests import synth
    rows, cols, n = 270, 480, 10
    clip, _ = synth.make_clip(rows, cols, n, seed=43, jitter=1.0)
    a = lvk.StabilizationFilter(lvk.StabilizationFilterSettings.obs_preset("homography", strict=False, predictive_samples=2), context=ctx)
    b = lvk.StabilizationFilter(lvk.StabilizationFilterSettings.obs_preset("homography", strict=False, predictive_samples=2), context=ctx)
    a.set_overlap(True); b.set_overlap(True)
    for i, f in enumerate(clip):
        y, u, v = oracle.egress_yuv420(f)
        pa = [_gpu(p) for p in (y, u, v)]; pb = [_gpu(p) for p in (y, u, v)]
        oa, _ = a.apply_obs("I420", pa, timestamp=i)
        ob, _ = b.apply_yuv420(tuple(pb), timestamp=i)
        ctx.sync()
        assert (oa is None) == (ob is None)
        if oa is not None:
            for g, h in zip(oa, ob):
                assert np.array_equal(g.cpu().numpy(), h.cpu().numpy())
    a.close(); b.close()


def test_push_obs_frame_size_change_and_refusal(ctx, oracle):
    """a UYVY source that is resized mid-stream: the queued frames leave at their OWN size; planes sized from the incoming frame are refused before
    anything changes (lvk_hip_stab_next_output), and the same push with planes that hold the delayed frame carries on bit-exactly; Y800 is refused."""
    import torch
    import livevisionkit_amd as lvk
    from tests import oracle_lib, synth
    from tests.test_stabilizer_gpu import _to_settings
    base, _ = synth.make_clip(360, 640, 12, seed=47, jitter=1.0)
    frames = [np.ascontiguousarray(f) for f in base[:6]] + [np.ascontiguousarray(f[60:300, 100:500]) for f in base[6:]]      # 640 x 360, then 400 x 240
    s = oracle_lib.preset("homography", predictive_samples=3, min_scene_quality=0.3, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    size_of = {i: f.shape[:2] for i, f in enumerate(frames)}
    refused = 0
    for i, f in enumerate(frames):
        planes = oracle.egress_obs("UYVY", f)
        packed = oracle.ingest_obs("UYVY", planes)
        big = np.zeros((360, 640, 3), np.uint8)
        w, wts = ost.push(packed, ts=i, fmt=4, out=big)
        dev = [_gpu(p) for p in planes]
        due = gst.next_output(f.shape[0], f.shape[1], 4)
        assert (due is None) == (w is None)
        if due is not None and (due[0], due[1]) != f.shape[:2]:
            small = [torch.empty((f.shape[0], f.shape[1], 2), dtype=torch.uint8, device="cuda")]
            before = gst.features()
            with pytest.raises(lvk.LvkHipError, match="DELAYED"):
                gst.apply_obs("UYVY", dev, timestamp=i, out=small)
            assert np.array_equal(gst.features(), before) and gst.next_output(f.shape[0], f.shape[1], 4) == due
            refused += 1
        out, ots = gst.apply_obs("UYVY", dev, timestamp=i)
        ctx.sync()
        assert (out is None) == (w is None)
        if out is not None:
            r, c = size_of[wts]
            assert ots == wts and tuple(out[0].shape[:2]) == (r, c)
            assert np.array_equal(out[0].cpu().numpy(), oracle.egress_obs("UYVY", w[:r, :c])[0])
    assert refused == 3
    y = torch.zeros((240, 400), dtype=torch.uint8, device="cuda")
    with pytest.raises(lvk.LvkHipError):
        gst.apply_obs("Y800", [y], timestamp=99, out=[y.clone()])
    gst.close()
