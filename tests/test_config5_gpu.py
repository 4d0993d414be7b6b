"""BASELINE config 5 and the full-size field path, through lvk_hip_stab_push_yuv420 exactly as bench.py drives it:
  * fused lens pre-warp + overlap mode + 4:2:0 planes in / out at 1080p and 3840x2160, both presets, I420 and NV12
    (kernels k_remap_homography_lens_420 / k_remap_mesh_lens_420 on the persistent grid, k_lens_undistort in the tracker chain);
  * the vector-field preset without a lens at 1080p / 4K (k_remap_mesh_420 on the persistent grid);
  * random ragged even sizes (the former scripts/fuzz_overlap.py), with and without a lens.
Every emitted plane is compared bit for bit with the oracle chain OracleStabilizer(.set_lens) + egress_yuv420.
Reference chain this replaces: LCFilter::filter ahead of VSFilter's apply (Modules/OBS-Plugin/Sources/Enhancement/LCFilter.cpp:133-192,
Sources/Stabilisation/VSFilter.cpp:352-364) between I4XXIngest / NV12Ingest::to_ocl and ::to_obs (Interop/FrameIngest.cpp:494-602)."""
import ctypes

import numpy as np
import pytest

from tests import oracle_lib, synth

pytestmark = pytest.mark.gpu


def survey_profile(rows, cols):
    """fx = fy = 0.8 W, cx = W / 2, cy = H / 2, k1 = -0.12, k2 = 0.03, p1 = p2 = k3 = 0."""
    return np.array([0.8 * cols, 0.8 * cols, cols / 2.0, rows / 2.0, -0.12, 0.03, 0.0, 0.0, 0.0])


def _conv(o):
    import livevisionkit_amd as lvk
    s = lvk.StabilizationFilterSettings()
    ctypes.memmove(ctypes.byref(s), ctypes.byref(o), ctypes.sizeof(o))
    return s


def _clip(rows, cols, n, seed, up):
    """Cheap full-size frames with corners: a small shaky clip, pixel-replicated `up` times."""
    small, _ = synth.make_clip(rows // up, cols // up, n, seed=seed, jitter=1.0)
    return np.ascontiguousarray(small.repeat(up, axis=1).repeat(up, axis=2))


def _run(ctx, oracle, frames, settings, nv12, lens, n_delay, check_stats=True):
    """Free-running overlap-mode pushes (no sync between them), compared after the last one."""
    import torch
    import livevisionkit_amd as lvk
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(settings)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(_conv(settings))
    gst.set_overlap(True)
    if lens is not None:
        ost.set_lens(lens); gst.set_lens(lens)
    wants, gots = [], []
    for i, f in enumerate(frames):
        planes = oracle.egress_yuv420(f, nv12=nv12)
        want, wts = ost.push(oracle.ingest_yuv420(*planes), ts=i, nthreads=32)
        got, gts = gst.apply_yuv420(tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in planes), timestamp=i)
        assert (want is None) == (got is None), i
        if check_stats:
            so, sg = ost.stats(), gst.stats()
            assert (so.n_detected, so.n_matched, so.n_tracked) == (sg.n_detected, sg.n_matched, sg.n_tracked), i
        if want is not None:
            assert wts == gts
            wants.append(oracle.egress_yuv420(want, nv12=nv12)); gots.append(got)
    ctx.sync()
    assert len(wants) == len(frames) - n_delay
    oracle_lib.require_live_warp(ost, f"{frames.shape[1]}x{frames.shape[2]} nv12={nv12} lens={lens is not None}")
    assert ost.stats().trust == gst.stats().trust
    for i, (w, g) in enumerate(zip(wants, gots)):
        for k, (a, b) in enumerate(zip(g, w)):
            a = a.cpu().numpy()
            if not np.array_equal(a, b):
                d = np.abs(a.astype(np.int32) - b.astype(np.int32))
                raise AssertionError(f"emitted frame {i} plane {k}: {int((d > 0).sum())} bytes differ, max |d| = {d.max()}")
    ost.close(); gst.close()
    return len(wants)


@pytest.mark.parametrize("preset", ["homography", "field"])
@pytest.mark.parametrize("size,nv12", [((1080, 1920), False), ((1080, 1920), True), ((2160, 3840), False), ((2160, 3840), True)])
def test_config5_fused_lens_overlap_yuv420(ctx, oracle, size, nv12, preset):
    rows, cols = size
    n = 9
    frames = _clip(rows, cols, n, seed=rows + 11 * int(nv12) + (5 if preset == "field" else 0), up=4)
    # relaxed quality assurance (both presets): the trust factor leaves zero within the clip, so the compared frames carry the lens, the
    # crop AND the stabilizing warp the tracker estimated (with the presets' strict 0.95 the scene-quality EMA keeps trust at 0 for the
    # whole clip and the comparison would be lens + crop only); _run asserts it through oracle_lib.require_live_warp
    s = oracle_lib.preset(preset, predictive_samples=2, min_scene_quality=0.4, min_tracking_quality=0.2)
    assert _run(ctx, oracle, frames, s, nv12, survey_profile(rows, cols), 2) == n - 2


@pytest.mark.parametrize("size,nv12", [((1080, 1920), False), ((2160, 3840), False), ((2160, 3840), True)])
def test_field_preset_overlap_yuv420_full_size(ctx, oracle, size, nv12):
    """Vector-field preset, no lens: k_remap_mesh_420 (in-kernel 16 x 16 mesh interpolation + 4:2:0 egress) on the persistent grid."""
    rows, cols = size
    n = 10
    frames = _clip(rows, cols, n, seed=rows + 3, up=4)
    s = oracle_lib.preset("field", predictive_samples=2, min_scene_quality=0.3, min_tracking_quality=0.2)
    assert _run(ctx, oracle, frames, s, nv12, None, 2) == n - 2


@pytest.mark.parametrize("trial", range(8))
def test_overlap_yuv420_random_ragged_sizes(ctx, oracle, trial):
    """Random even frame sizes (ragged right / bottom strips, persistent and full remap grids), both presets, I420 / NV12, every other
    trial with the fused lens."""
    rng = np.random.default_rng(700 + trial)
    rows = int(rng.integers(150, 560)) * 2; cols = int(rng.integers(250, 980)) * 2
    nv12 = bool(trial & 1)
    preset = "field" if trial % 3 == 0 else "homography"
    lens = survey_profile(rows, cols) if trial % 2 == 0 else None
    if trial == 5:
        lens = np.array([0.7 * cols, 0.75 * cols, 0.52 * cols, 0.47 * rows, -0.2, 0.05, 1e-3, -2e-3, 0.01])     # decentred, tangential terms
    frames = _clip(rows, cols, 9, seed=trial + 100, up=2)
    s = oracle_lib.preset(preset, predictive_samples=2, min_scene_quality=0.3, min_tracking_quality=0.2)
    assert _run(ctx, oracle, frames, s, nv12, lens, 2) == 7
