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
