"""GPU parity of the debug overlays against the CPU oracle, through the C-ABI.  Bar: bit-exact."""
import numpy as np
import pytest

from tests import oracle_lib, synth

pytestmark = pytest.mark.gpu


def _gpu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("size,grid,thickness", [((270, 480), (15, 15), 1), ((2160, 3840), (15, 15), 1), ((1080, 1920), (1, 1), 1),
                                                  ((67, 131), (7, 3), 2), ((50, 50), (64, 64), 1), ((33, 47), (5, 9), 3)])
def test_draw_grid_bit_exact(ctx, oracle, size, grid, thickness):
    rows, cols = size
    src = np.random.default_rng(rows).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    want = oracle.draw_grid(src, grid, (29, 255, 107), thickness)
    got = ctx.draw_grid(_gpu(src), grid, (29, 255, 107), thickness); ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("size", [(270, 480), (2160, 3840), (40, 56)])
def test_draw_crosses_bit_exact(ctx, oracle, size):
    rows, cols = size
    rng = np.random.default_rng(cols)
    src = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    pts = np.c_[rng.uniform(-10, 490, 1500), rng.uniform(-10, 280, 1500)].astype(np.float32)
    pts[:6] = [(0, 0), (479.5, 269.5), (0.5, 1.5), (1e9, 3), (-1e9, -1e9), (np.nan, 5)]
    sc = (cols / 480.0, rows / 270.0)
    for cs, th in ((7, 4), (1, 1), (12, 2)):
        want = oracle.draw_crosses(src, pts, (76, 84, 255), cs, th, scaling=sc)
        got = ctx.draw_crosses(_gpu(src), pts, (76, 84, 255), cs, th, scaling=sc); ctx.sync()
        assert np.array_equal(got.cpu().numpy(), want), (size, cs, th)
    got = ctx.draw_crosses(_gpu(src), np.zeros((0, 2), np.float32), (1, 2, 3), 7, 4); ctx.sync()
    assert np.array_equal(got.cpu().numpy(), src)


@pytest.mark.parametrize("preset,fmt", [("homography", 4), ("field", 4), ("homography", 0)])
def test_stabilizer_test_mode_overlays_bit_exact(ctx, oracle, preset, fmt):
    """VSFilter test mode (VSFilter.cpp:356-361): apply, draw_motion_mesh, draw_trackers every frame -- the overlays are drawn
    into the queued frame and come out stabilized N frames later."""
    import ctypes
    import livevisionkit_amd as lvk
    frames, _ = synth.make_clip(360, 640, 14, seed=13)
    s = oracle_lib.preset(preset, predictive_samples=3)
    g = lvk.StabilizationFilterSettings(); ctypes.memmove(ctypes.byref(g), ctypes.byref(s), ctypes.sizeof(s))
    d = oracle_lib.preset("default")
    gd = lvk.StabilizationFilterSettings(); ctypes.memmove(ctypes.byref(gd), ctypes.byref(d), ctypes.sizeof(d))
    ost = oracle_lib.OracleStabilizer(oracle, d); ost.configure(s)
    gst = lvk.StabilizationFilter(gd, context=ctx); gst.configure(g)
    produced = 0
    for i, f in enumerate(frames):
        want, _ = ost.push(f, ts=i, fmt=fmt); ost.draw_motion_mesh(); ost.draw_trackers()
        gf = _gpu(f)
        got, _ = gst.apply(gf, timestamp=i, fmt=fmt); gst.draw_motion_mesh(); gst.draw_trackers()
        ctx.sync()
        assert (want is None) == (got is None)
        if want is not None:
            assert np.array_equal(got.cpu().numpy(), want), i
            produced += 1
    assert produced == len(frames) - 3
    assert not np.array_equal(gf.cpu().numpy(), f)                     # drawn in place into the caller's (borrowed) frame
    ost.close(); gst.close()
