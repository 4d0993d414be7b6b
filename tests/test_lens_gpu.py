"""GPU parity of the lens-correction row against the CPU oracle, through the C-ABI:
the host-built undistortion offset map, the map-driven EASU remap, and the LC -> VS filter chain.  Bar: bit-exact."""
import numpy as np
import pytest

from tests import oracle_lib, synth

pytestmark = pytest.mark.gpu

PROFILES = [
    lambda r, c: (0.8 * c, 0.8 * c, c / 2, r / 2, -0.12, 0.03, 0, 0, 0),
    lambda r, c: (0.9 * c, 0.85 * c, c / 2 + 3, r / 2 - 2, -0.2, 0.05, 1e-3, -2e-3, 0.01),
    lambda r, c: (1.1 * c, 1.1 * c, c / 2, r / 2, 0.08, -0.01, 0, 0, 0),
    lambda r, c: (1.0 * c, 1.0 * c, c / 2, r / 2, 0, 0, 0, 0, 0),
]


def _gpu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("size", [(135, 240), (270, 480), (67, 131)])
@pytest.mark.parametrize("profile", range(len(PROFILES)))
def test_lens_map_bit_exact(ctx, oracle, size, profile):
    rows, cols = size
    params = PROFILES[profile](rows, cols)
    want, wview = oracle.lens_offset_map(params, rows, cols)
    got, gview = ctx.lens_map(params, rows, cols)
    assert tuple(gview) == tuple(wview)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("yuv", [True, False])
@pytest.mark.parametrize("size", [(72, 96), (67, 131), (270, 480), (5, 9)])
def test_remap_map_bit_exact(ctx, oracle, yuv, size):
    rows, cols = size
    rng = np.random.default_rng(rows * 7 + cols)
    src = synth.textured_frame(rows, cols, seed=rows) if rows >= 32 else rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    dsrc = _gpu(src)
    maps = [rng.uniform(-3, 3, (rows, cols, 2)).astype(np.float32),
            rng.uniform(-1.5 * cols, 1.5 * cols, (rows, cols, 2)).astype(np.float32),        # mostly out of bounds
            oracle.lens_offset_map(PROFILES[0](rows, cols), rows, cols)[0]]
    for k, m in enumerate(maps):
        want = oracle.remap_map(src, m, bg=(9, 99, 199), yuv=yuv)
        got = ctx.remap_map(dsrc, _gpu(m), bg=(9, 99, 199), yuv=yuv)
        ctx.sync()
        assert np.array_equal(got.cpu().numpy(), want), f"map {k} {size} yuv={yuv}"


def test_remap_map_non_finite_offsets(ctx, oracle):
    rows, cols = 40, 56
    src = synth.textured_frame(rows, cols, seed=2)
    m = np.zeros((rows, cols, 2), np.float32)
    m[5, 5] = (np.nan, 0); m[6, 6] = (np.inf, 1); m[7, 7] = (-np.inf, -np.inf); m[8, 8] = (1e30, -1e30)
    want = oracle.remap_map(src, m)
    got = ctx.remap_map(_gpu(src), _gpu(m)); ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)


def test_remap_map_1080p_full_size(ctx, oracle):
    rows, cols = 1080, 1920
    src = synth.textured_frame(rows, cols, seed=11)
    params = PROFILES[0](rows, cols)
    want_map, _ = oracle.lens_offset_map(params, rows, cols)
    got_map, _ = ctx.lens_map(params, rows, cols)
    assert np.array_equal(got_map.cpu().numpy(), want_map)
    want = oracle.remap_map(src, want_map, nthreads=32)
    got = ctx.remap_map(_gpu(src), got_map); ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)


def test_lens_then_stabilize_chain(ctx, oracle):
    """The OBS filter chain LC -> VS: every stabilized frame of the lens-corrected clip matches the oracle's."""
    import ctypes
    import livevisionkit_amd as lvk
    rows, cols = 270, 480
    frames, _ = synth.make_clip(rows, cols, 14, seed=21)
    params = PROFILES[0](rows, cols)
    omap, _ = oracle.lens_offset_map(params, rows, cols)
    gmap, _ = ctx.lens_map(params, rows, cols)
    so = oracle_lib.preset("homography", predictive_samples=4)
    sg = lvk.StabilizationFilterSettings()
    ctypes.memmove(ctypes.byref(sg), ctypes.byref(so), ctypes.sizeof(so))
    ost = oracle_lib.OracleStabilizer(oracle, so)
    gst = lvk.StabilizationFilter(sg, context=ctx)
    produced = 0
    for i, f in enumerate(frames):
        co = oracle.remap_map(f, omap)
        cg = ctx.remap_map(_gpu(f), gmap)
        want, _ts = ost.push(co, ts=i)
        got, _gts = gst.apply(cg, timestamp=i)
        ctx.sync()
        assert (want is None) == (got is None), i
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want), f"frame {i}"
            produced += 1
    ost.close(); gst.close()
    assert produced == len(frames) - 4


# ---- fused lens mode -----------------------------------------------------------------------------------------------------
def _grid(rows, cols):
    jj, ii = np.meshgrid(np.arange(cols, dtype=np.float32), np.arange(rows, dtype=np.float32))
    return np.stack([jj, ii], axis=2)


@pytest.mark.parametrize("profile", range(len(PROFILES)))
def test_undistort_points_bit_exact(ctx, oracle, profile):
    rows, cols = 2160, 3840
    params = PROFILES[profile](rows, cols)
    rng = np.random.default_rng(profile)
    pts = np.c_[rng.uniform(-20, 500, 1856), rng.uniform(-20, 290, 1856)].astype(np.float32)
    pts[:4] = [(0, 0), (479, 269), (1e6, -1e6), (np.nan, 3)]
    want = oracle.lens_undistort_points(params, rows, cols, 8.0, 8.0, pts)
    got = ctx.lens_undistort_points(params, rows, cols, 8.0, 8.0, pts)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("yuv", [True, False])
@pytest.mark.parametrize("size", [(72, 96), (67, 131), (270, 480)])
def test_warpmesh_apply_lens_bit_exact(ctx, oracle, yuv, size):
    rows, cols = size
    rng = np.random.default_rng(rows + cols)
    src = synth.textured_frame(rows, cols, seed=cols)
    dsrc = _gpu(src)
    for profile in (0, 1):
        params = PROFILES[profile](rows, cols)
        for mesh in (np.zeros((2, 2, 2), np.float32), rng.uniform(-0.03, 0.03, (2, 2, 2)).astype(np.float32),
                     synth.random_mesh(16, 16, rng, amp=0.02), synth.random_mesh(3, 5, rng, amp=0.3)):
            want = oracle.warpmesh_apply_lens(src, mesh, params, bg=(7, 70, 170), yuv=yuv)
            got = ctx.warpmesh_apply_lens(dsrc, mesh, params, bg=(7, 70, 170), yuv=yuv)
            ctx.sync()
            assert np.array_equal(got.cpu().numpy(), want), (size, yuv, profile, mesh.shape)


def _fused_pair(ctx, oracle, frames, params, settings, then_configure=None, overlap=False):
    import ctypes
    import livevisionkit_amd as lvk

    def conv(o):
        s = lvk.StabilizationFilterSettings()
        ctypes.memmove(ctypes.byref(s), ctypes.byref(o), ctypes.sizeof(o))
        return s
    ost = oracle_lib.OracleStabilizer(oracle, settings)
    gst = lvk.StabilizationFilter(conv(settings), context=ctx)
    if overlap:
        gst.set_overlap(True)
    if then_configure is not None:
        ost.configure(then_configure); gst.configure(conv(then_configure))
    ost.set_lens(params); gst.set_lens(params)
    produced = 0
    for i, f in enumerate(frames):
        want, _ = ost.push(f, ts=i)
        got, _ = gst.apply(_gpu(f), timestamp=i)
        ctx.sync()
        so, sg = ost.stats(), gst.stats()
        for k in ("n_detected", "n_matched", "n_tracked", "tracking_stability", "trust"):
            assert getattr(so, k) == getattr(sg, k), (i, k, getattr(so, k), getattr(sg, k))
        mo, _ = ost.meshes(); mg, _ = gst.meshes()
        assert np.array_equal(mo.view(np.uint32), mg.view(np.uint32)), i
        assert (want is None) == (got is None), i
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want), f"frame {i}"
            produced += 1
    ost.close(); gst.close()
    return produced


@pytest.fixture(scope="module")
def lens_clip(oracle):
    rows, cols = 360, 640
    frames, _ = synth.make_clip(rows, cols, 18, seed=31)
    params = PROFILES[0](rows, cols)
    cr = oracle.lens_undistort_points(params, rows, cols, 1.0, 1.0, _grid(rows, cols).reshape(-1, 2)).reshape(rows, cols, 2)
    return synth.lens_distort(frames, cr), params


def test_fused_stabilizer_homography_bit_exact(ctx, oracle, lens_clip):
    raw, params = lens_clip
    s = oracle_lib.preset("homography", predictive_samples=4)
    assert _fused_pair(ctx, oracle, raw, params, s) == len(raw) - 4
    assert _fused_pair(ctx, oracle, raw, params, s, overlap=True) == len(raw) - 4


def test_fused_stabilizer_field_bit_exact(ctx, oracle, lens_clip):
    raw, params = lens_clip
    field = oracle_lib.preset("field", predictive_samples=3, min_scene_quality=0.4, min_tracking_quality=0.2)
    assert _fused_pair(ctx, oracle, raw[:14], params, oracle_lib.preset("default"), then_configure=field) == 11


def test_fused_passthrough_and_switch_off(ctx, oracle, lens_clip):
    """stabilize_output = false still applies the lens map (identity stabilizing warp); set_lens(None) returns to the plain filter."""
    import ctypes
    import livevisionkit_amd as lvk
    raw, params = lens_clip
    so = oracle_lib.preset("homography", predictive_samples=2, stabilize_output=0, crop_to_stable_region=0)
    sg = lvk.StabilizationFilterSettings(); ctypes.memmove(ctypes.byref(sg), ctypes.byref(so), ctypes.sizeof(so))
    ost = oracle_lib.OracleStabilizer(oracle, so); gst = lvk.StabilizationFilter(sg, context=ctx)
    ost.set_lens(params); gst.set_lens(params)
    n = 0
    for i, f in enumerate(raw[:6]):
        want, _ = ost.push(f, ts=i); got, _ = gst.apply(_gpu(f), timestamp=i); ctx.sync()
        assert (want is None) == (got is None)
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want), i
            assert not np.array_equal(want, raw[i - 2])
            n += 1
    assert n == 4
    ost.set_lens(None); gst.set_lens(None)
    for i, f in enumerate(raw[:4]):
        want, _ = ost.push(f, ts=i); got, _ = gst.apply(_gpu(f), timestamp=i); ctx.sync()
        assert (want is None) == (got is None)
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want) and np.array_equal(want, raw[i - 2])
    ost.close(); gst.close()
