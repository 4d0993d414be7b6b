"""The PACKED boundary (lvk_hip_stab_push, the entry lvk::StabilizationFilter::filter and with it the OBS plugin's VSFilter use) when the frame
size changes in the middle of a stream.  Reference: the queue holds whole frames (Filters/StabilizationFilter.cpp:118-131), WarpMesh::apply
creates dst from the DELAYED source (Math/WarpMesh.cpp:183-223 -> Functions/Image.cpp:53,116), and VSFilter.cpp:352-364 does not restart its
filter on a resize: the next frame_delay pushes emit frames of the OLD size and format.  Every emitted frame is held to the oracle's frame of
the same timestamp; the output buffers carry guard rows / columns that must stay untouched; a push whose output would not fit is refused before
anything changes, and the same push with a large enough buffer then carries on bit-exactly."""
import numpy as np
import pytest

from tests import oracle_lib, synth
from tests.test_stabilizer_gpu import _to_settings

pytestmark = pytest.mark.gpu

GUARD = 0xA5
SIZES = [(1080, 1920), (800, 1920), (720, 1280), (1080, 1920)]      # fewer rows at the same width, then narrower, then larger again
FORMATS = [4, 4, 0, 4]                                               # the 720p segment is BGR: the emitted frame carries the DELAYED frame's format
SEG = 6


def _segments():
    """One shaky 1080p clip; the segments are crops of it (the tracker sees a jump at every change and carries on)."""
    base, _ = synth.make_clip(1080, 1920, SEG * len(SIZES), seed=77, jitter=1.0)
    out = []
    for k, ((r, c), fmt) in enumerate(zip(SIZES, FORMATS)):
        y0, x0 = (1080 - r) // 2, (1920 - c) // 2
        for f in base[k * SEG:(k + 1) * SEG]:
            f = f[y0:y0 + r, x0:x0 + c]
            if fmt == 0:
                f = f[..., [1, 0, 2]]                                     # the textured channel where the BGR grey value weighs most
            out.append((np.ascontiguousarray(f), fmt))
    return out


@pytest.fixture(scope="module")
def segments():
    return _segments()


def _guarded(rows, cols, device="cuda"):
    """A buffer with 8 guard rows above and below and 32 guard bytes per row behind the frame: (whole buffer, the rows x cols x 3 view)."""
    import torch
    pitch = cols * 3 + 32
    buf = torch.full((rows + 16, pitch), GUARD, dtype=torch.uint8, device=device)
    return buf, buf.as_strided((rows, cols, 3), (pitch, 3, 1), 8 * pitch)


def _guards_intact(buf, rows, cols):
    b = buf.cpu().numpy()
    return (b[:8] == GUARD).all() and (b[8 + rows:] == GUARD).all() and (b[8:8 + rows, cols * 3:] == GUARD).all()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("stabilize", [True, False])
def test_packed_frame_size_changes_mid_stream(ctx, oracle, segments, overlap, stabilize):
    import torch
    import livevisionkit_amd as lvk
    s = oracle_lib.preset("homography", predictive_samples=3, min_scene_quality=0.3, min_tracking_quality=0.2, stabilize_output=1 if stabilize else 0)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    gst.set_overlap(overlap)
    size_of = {i: f.shape[:2] for i, (f, _) in enumerate(segments)}
    fmt_of = {i: fmt for i, (_, fmt) in enumerate(segments)}
    emitted, refused, live = 0, 0, 0
    keep = []                                                          # borrowed inputs / in-flight outputs stay alive
    for i, (f, fmt) in enumerate(segments):
        big = np.zeros((1080, 1920, 3), np.uint8)
        w, wts = ost.push(f, ts=i, fmt=fmt, out=big, nthreads=32)
        d = torch.from_numpy(f).cuda(); keep.append(d)
        due = gst.next_output(f.shape[0], f.shape[1], fmt)
        assert (due is None) == (w is None), i
        if due is not None:
            assert due == (*size_of[wts], fmt_of[wts]), (i, due)
            if due[0] > f.shape[0] or due[1] > f.shape[1]:
                # what the round-5 facade did: an output sized from the INCOMING frame.  Refused, and nothing has changed.
                _, small = _guarded(f.shape[0], f.shape[1])
                before = gst.features()
                with pytest.raises(lvk.LvkHipError, match="DELAYED"):
                    gst.apply(d, timestamp=i, out=small, fmt=fmt)
                assert gst.next_output(f.shape[0], f.shape[1], fmt) == due and np.array_equal(gst.features(), before)
                refused += 1
            buf, view = _guarded(due[0], due[1])
            keep.append(buf)
            g, gts = gst.apply(d, timestamp=i, out=view, fmt=fmt)
        else:
            g, gts = gst.apply(d, timestamp=i, fmt=fmt)
        ctx.sync()
        assert (g is None) == (w is None), i
        if g is not None:
            r, c = size_of[wts]
            assert gts == wts and tuple(g.shape) == (r, c, 3) and gst.last_format == fmt_of[wts], (i, gts, wts, tuple(g.shape))
            assert np.array_equal(g.cpu().numpy(), big[:r, :c]), f"push {i}: emitted frame {gts} ({c}x{r}) differs from the oracle's"
            assert _guards_intact(buf, r, c), f"push {i}: the remap wrote outside the {c}x{r} output"
            emitted += 1
        if stabilize:
            assert np.array_equal(gst.features(), ost.features()), i
            so, sg = ost.stats(), gst.stats()
            assert (so.trust, so.n_matched, list(so.homography)) == (sg.trust, sg.n_matched, list(sg.homography)), i
            live += so.trust > 0.1
    assert emitted == len(segments) - 3                                 # every frame leaves, the old sizes included
    assert refused >= 2                                                 # 1080p -> 1920x800 (fewer rows: was an out-of-bounds write), 1920x800 -> 720p (narrower)
    if stabilize:
        assert live >= 6, "the clip must be tracked for part of the run (a warp, not only the crop)"
    ost.close(); gst.close()


def test_packed_resize_with_a_lens_profile(ctx, oracle, segments):
    """Fused lens mode across a resize: the pre-warp of every emitted frame is built for THAT frame's size (round 5 refused such a stream)."""
    import torch
    import livevisionkit_amd as lvk
    s = oracle_lib.preset("homography", predictive_samples=2, min_scene_quality=0.3, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, s); gst = lvk.StabilizationFilter(_to_settings(s), context=ctx)
    # one profile for the stream (the plugin's LCFilter holds one, LCFilter.cpp:133-171): centred for 1080p
    p = (0.8 * 1920, 0.8 * 1920, 1920 / 2, 1080 / 2, -0.12, 0.03, 0, 0, 0)
    ost.set_lens(p); gst.set_lens(p)
    frames = [segments[k] for k in (0, 1, 2, 3, SEG, SEG + 1, SEG + 2, SEG + 3)]
    size_of = {i: f.shape[:2] for i, (f, _) in enumerate(frames)}
    keep, n = [], 0
    for i, (f, fmt) in enumerate(frames):
        big = np.zeros((1080, 1920, 3), np.uint8)
        w, wts = ost.push(f, ts=i, out=big, nthreads=32)
        d = torch.from_numpy(f).cuda(); keep.append(d)
        g, gts = gst.apply(d, timestamp=i); ctx.sync()
        assert (g is None) == (w is None), i
        if g is not None:
            r, c = size_of[wts]
            assert gts == wts and tuple(g.shape) == (r, c, 3) and np.array_equal(g.cpu().numpy(), big[:r, :c]), (i, gts)
            n += 1
    assert n == len(frames) - 2
    ost.close(); gst.close()
