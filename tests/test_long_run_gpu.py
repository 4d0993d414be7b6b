"""SURVEY.md section 8d's clip at length: 600 frames of the synthetic 1080p stream (tests/clipgen.py: smooth pan + AR(1) jitter in
translation, rotation and zoom, scene cut at frame 300) as I420 planes through lvk_hip_stab_push_yuv420 in overlap mode, every emitted
plane against the oracle chain -- the QA state machine (trust falls at the cut and recovers), the smoother's adaptive factor, 600
consecutive tracker states, the persistent remap grid with real rotations / zooms."""
import ctypes

import numpy as np
import pytest

from tests import clipgen, oracle_lib

pytestmark = pytest.mark.gpu


def _conv(o):
    import livevisionkit_amd as lvk
    s = lvk.StabilizationFilterSettings()
    ctypes.memmove(ctypes.byref(s), ctypes.byref(o), ctypes.sizeof(o))
    return s


@pytest.mark.parametrize("preset,rows,cols,n", [("homography", 1080, 1920, 600), ("field", 540, 960, 240)])
def test_600_frames_overlap_yuv420_bit_exact(ctx, oracle, preset, rows, cols, n):
    import torch
    import livevisionkit_amd as lvk
    clip = clipgen.Clip(rows, cols, n, device="cuda", cut_at=n // 2)
    s = oracle_lib.preset(preset)                                       # the OBS preset as shipped: strict QA, predictive_samples 10, crop on
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(_conv(s))
    gst.set_overlap(True)
    delay = s.predictive_samples
    trust, emitted, pending = [], 0, []
    for i in range(n):
        planes = clip.render_i420(i)
        torch.cuda.synchronize()                                        # rendered on torch's stream, consumed on the filter's
        host = [p.cpu().numpy() for p in planes]
        want, wts = ost.push(oracle.ingest_yuv420(*host), ts=i, nthreads=32)
        got, gts = gst.apply_yuv420(planes, timestamp=i)
        so, sg = ost.stats(), gst.stats()
        assert (so.n_detected, so.n_matched, so.n_tracked, so.tracking_stability, so.trust) == \
               (sg.n_detected, sg.n_matched, sg.n_tracked, sg.tracking_stability, sg.trust), i
        trust.append(sg.trust)
        assert (want is None) == (got is None), i
        if want is not None:
            assert wts == gts == i - delay
            pending.append((i, got, oracle.egress_yuv420(want)))
            emitted += 1
        if len(pending) >= 8 or i == n - 1:                             # compare in batches: the pushes stay free-running in between
            ctx.sync()
            for k, g, w in pending:
                for a, b in zip(g, w):
                    assert np.array_equal(a.cpu().numpy(), b), f"frame {k}"
            pending = []
    assert emitted == n - delay
    trust = np.array(trust)
    cut = n // 2
    assert trust[cut - 5:cut].min() == 1.0, "trust had recovered before the cut"
    assert trust[cut:cut + 12].min() == 0.0, "the scene cut must drop the trust factor"
    assert trust[-1] == 1.0, "and it recovers afterwards"
    ost.close(); gst.close()


@pytest.mark.parametrize("placement", ["tracker", "bulk", None])
def test_free_running_pushes_bit_exact(oracle, placement, monkeypatch):
    """40 pushes of a 4K I420 stream back to back, no synchronisation in between (what bench.py times): the 4:2:0 conversion of a push then
    finds the bulk stream still busy with the previous remap and goes behind the tracker chain on the tracking stream (per-slot events,
    the push waiting for the chain through an event).  Both placements pinned, and the per-push decision, against the oracle."""
    import torch
    import livevisionkit_amd as lvk
    if placement: monkeypatch.setenv("LVK_HIP_INGEST_PLACEMENT", placement)
    else: monkeypatch.delenv("LVK_HIP_INGEST_PLACEMENT", raising=False)
    rows, cols, n = 2160, 3840, 40
    clip = clipgen.Clip(rows, cols, n, device="cuda", cut_at=None)
    s = oracle_lib.preset("homography")
    s.predictive_samples = 4
    ws = torch.cuda.Stream()
    ctx = lvk.Context(0, stream=ws)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(_conv(s))
    gst.set_overlap(True)
    planes = [clip.render_i420(i) for i in range(n)]
    torch.cuda.synchronize()
    got = [gst.apply_yuv420(planes[i], timestamp=i) for i in range(n)]          # free running
    ctx.sync()
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    emitted = 0
    for i in range(n):
        want, wts = ost.push(oracle.ingest_yuv420(*[p.cpu().numpy() for p in planes[i]]), ts=i, nthreads=32)
        g, gts = got[i]
        assert (want is None) == (g is None), i
        if want is not None:
            assert wts == gts
            for a, b in zip(g, oracle.egress_yuv420(want)):
                assert np.array_equal(a.cpu().numpy(), b), f"frame {i} ({placement})"
            emitted += 1
    assert emitted == n - s.predictive_samples
    oracle_lib.require_live_warp(ost, "4K free-running pushes")
    ost.close(); gst.close(); ctx.close()
