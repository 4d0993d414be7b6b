"""GPU parity of the tracker image operations against the CPU oracle, through
the C-ABI.  Bar: bit-exact (integer arithmetic; the float steps use the same binary32 op sequence)."""
import numpy as np
import pytest

from tests import synth
from tests.test_oracle_imgproc import _smooth_scene

pytestmark = pytest.mark.gpu


def _gpu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("src_size,dst_size", [((1080, 1920), (270, 480)), ((2160, 3840), (270, 480)), ((720, 1280), (270, 480)),
                                               ((540, 960), (270, 480)), ((270, 480), (270, 480)), ((300, 500), (256, 256)),
                                               ((777, 1033), (100, 333))])
def test_luma_area_resize_packed(ctx, oracle, src_size, dst_size):
    rng = np.random.default_rng(src_size[0])
    frame = rng.integers(0, 256, src_size + (3,), dtype=np.uint8)
    want = oracle.luma_area_resize(frame, *dst_size)
    got = ctx.luma_area_resize(_gpu(frame), *dst_size)
    ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)


def test_luma_area_resize_planar(ctx, oracle):
    plane = synth.textured_frame(1080, 1920, seed=2, channels=1)
    want = oracle.luma_area_resize(plane, 270, 480)
    got = ctx.luma_area_resize(_gpu(plane), 270, 480)
    ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("src_size,dst_size", [((1440, 2560), (270, 480)), ((1200, 1920), (270, 480)), ((720, 1280), (270, 480)), ((1600, 2560), (270, 480)),
                                               ((2162, 3846), (270, 480)), ((271, 481), (270, 480)), ((600, 700), (64, 97)), ((2400, 4000), (270, 480))])
def test_luma_area_resize_fractional_scales_planar(ctx, oracle, src_size, dst_size):
    """Non-integer scales (resizeArea_ with its decimate-alpha tables): 1440p / 1200p / 720p canvases, barely-above-one and ragged scales, up to 8
    taps per axis through the batched-load kernel, beyond that (2400 -> 270: 9-10 taps) through the loop kernel -- same sums in the same order."""
    rng = np.random.default_rng(src_size[1])
    plane = rng.integers(0, 256, src_size, dtype=np.uint8)
    want = oracle.luma_area_resize(plane, *dst_size)
    got = ctx.luma_area_resize(_gpu(plane), *dst_size)
    ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("src_size,dst_size", [((180, 320), (270, 480)), ((135, 240), (270, 480)), ((200, 300), (270, 480)), ((400, 300), (270, 480)),
                                               ((100, 640), (270, 480)), ((269, 479), (270, 480)), ((7, 5), (33, 47))])
@pytest.mark.parametrize("packed", [False, True])
def test_luma_area_resize_enlargement(ctx, oracle, src_size, dst_size, packed):
    """INTER_AREA towards a larger image on one or both axes (cv::resize's bilinear emulation): k_area_enlarge against the oracle."""
    rng = np.random.default_rng(src_size[0] + 3)
    frame = rng.integers(0, 256, src_size + ((3,) if packed else ()), dtype=np.uint8)
    want = oracle.luma_area_resize(frame, *dst_size)
    got = ctx.luma_area_resize(_gpu(frame), *dst_size)
    ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("shape", [(270, 480), (135, 240), (68, 120), (33, 47), (256, 256)])
def test_pyr_down_and_scharr(ctx, oracle, shape):
    img = np.random.default_rng(shape[1]).integers(0, 256, shape, dtype=np.uint8)
    d = _gpu(img)
    got_p = ctx.pyr_down(d)
    got_s = ctx.scharr(d)
    ctx.sync()
    assert np.array_equal(got_p.cpu().numpy(), oracle.pyr_down(img))
    assert np.array_equal(got_s.cpu().numpy(), oracle.scharr_deriv(img))


@pytest.mark.parametrize("shape", [(270, 480), (256, 256), (143, 211), (97, 100), (40, 40)])
def test_whole_pyramid_matches_oracle(ctx, oracle, shape):
    """The fused pyramid launch (levels 1-3 from one LDS window) and the generic per-level path: every level and every
    derivative image identical to the pyrDown / Scharr chain of the oracle."""
    img = np.random.default_rng(shape[0] * 7 + shape[1]).integers(0, 256, shape, dtype=np.uint8)
    got = ctx.build_pyramid(_gpu(img))
    sizes = oracle.pyramid_levels(*shape)
    assert [g[0].shape for g in got] == sizes
    cur = img
    for lvl, (g_img, g_der) in enumerate(got):
        if lvl > 0:
            cur = oracle.pyr_down(cur)
        assert np.array_equal(g_img, cur), (shape, lvl)
        assert np.array_equal(g_der, oracle.scharr_deriv(cur)), (shape, lvl)


@pytest.mark.parametrize("layout", ["2x1", "2x2", "1x1", "odd"])
def test_fast_detect_regions(ctx, oracle, layout):
    rows, cols = 270, 480
    img = synth.textured_frame(rows, cols, seed=17, channels=1)
    if layout == "2x1":
        regions = [(0, 0, 240, 270, 10, 1), (240, 0, 240, 270, 35, 1)]
    elif layout == "2x2":
        regions = [(0, 0, 240, 135, 10, 1), (240, 0, 240, 135, 20, 0), (0, 135, 240, 135, 60, 1), (240, 135, 240, 135, 15, 1)]
    elif layout == "1x1":
        regions = [(0, 0, 480, 270, 12, 1)]
    else:
        regions = [(5, 7, 131, 77, 10, 1), (200, 100, 65, 9, 10, 1), (300, 3, 6, 200, 10, 1), (100, 150, 201, 119, 250, 1)]
    got, counts = ctx.fast_detect(_gpu(img), regions)
    total = 0
    for i, (x, y, w, h, t, active) in enumerate(regions):
        want = oracle.fast(img, t, roi=(x, y, w, h)) if active else np.zeros((0, 3), np.int32)
        assert counts[i] == len(want), (layout, i, counts[i], len(want))
        assert np.array_equal(got[i], want), (layout, i)
        total += len(want)
    assert total > 50 or layout == "odd"


def test_fast_detect_capacity_truncates_but_counts_all(ctx, oracle):
    img = synth.textured_frame(270, 480, seed=3, channels=1)
    want = oracle.fast(img, 10)
    got, counts = ctx.fast_detect(_gpu(img), [(0, 0, 480, 270, 10, 1)], cap=40)
    assert counts[0] == len(want) and len(want) > 40
    assert np.array_equal(got[0], want[:40])


@pytest.mark.parametrize("shift", [(0.0, 0.0), (1.3, -0.7), (-3.6, 2.2), (6.5, 4.25)])
def test_pyrlk_bit_exact_smooth_scene(ctx, oracle, shift):
    rows, cols = 270, 480
    prev = _smooth_scene(rows, cols, 0, 0)
    nxt = _smooth_scene(rows, cols, -shift[0], -shift[1])
    rng = np.random.default_rng(4)
    pts = np.c_[rng.uniform(-5, cols + 5, 600), rng.uniform(-5, rows + 5, 600)].astype(np.float32)
    want_p, want_s = oracle.pyrlk(prev, nxt, pts)
    got_p, got_s = ctx.pyrlk(_gpu(prev), _gpu(nxt), pts)
    assert np.array_equal(got_s, want_s)
    assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32)), np.abs(got_p - want_p).max()


def test_pyrlk_bit_exact_textured_and_corners(ctx, oracle):
    """Real feature positions (FAST corners) on a noisy textured frame warped by a homography-like shift."""
    rows, cols = 270, 480
    big = synth.textured_frame(rows + 20, cols + 20, seed=23, channels=1)
    prev = np.ascontiguousarray(big[10:-10, 10:-10])
    nxt = np.ascontiguousarray(big[8:-12, 13:-7])                    # content moves by (-3, +2)
    kp = oracle.fast(prev, 20)
    pts = kp[:, :2].astype(np.float32)
    assert len(pts) > 200
    want_p, want_s = oracle.pyrlk(prev, nxt, pts)
    got_p, got_s = ctx.pyrlk(_gpu(prev), _gpu(nxt), pts)
    assert np.array_equal(got_s, want_s)
    assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32))
    ok = want_s == 1
    assert np.median(np.abs(want_p[ok] - (pts[ok] + np.array([-3, 2], np.float32)))) < 0.1


def test_pyrlk_other_geometry(ctx, oracle):
    """Library-default 256x256 tracking frame, small frames with fewer pyramid levels, other windows."""
    for (rows, cols, win, lv) in [(256, 256, (11, 11), 3), (40, 40, (11, 11), 3), (90, 120, (7, 9), 2), (135, 240, (15, 15), 3)]:
        prev = _smooth_scene(rows, cols, 0, 0)
        nxt = _smooth_scene(rows, cols, 0.8, -1.1)
        rng = np.random.default_rng(rows)
        pts = np.c_[rng.uniform(0, cols, 150), rng.uniform(0, rows, 150)].astype(np.float32)
        want_p, want_s = oracle.pyrlk(prev, nxt, pts, win=win, max_level=lv)
        got_p, got_s = ctx.pyrlk(_gpu(prev), _gpu(nxt), pts, win=win, max_level=lv)
        assert np.array_equal(got_s, want_s), (rows, cols, win)
        assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32)), (rows, cols, win)


def test_pyrlk_large_windows_and_large_flow(ctx, oracle):
    """Windows up to the 31 x 31 limit (the per-level LDS of a block then exceeds the default dynamic-LDS cap and the window staging
    takes its generic chunk loops), deep pyramids, and flows beyond the staged search margin (the next-frame window is re-centred)."""
    for (rows, cols, win, lv, shift) in [(540, 960, (31, 31), 4, (1.3, -0.7)), (270, 480, (21, 25), 3, (0.4, 0.9)),
                                          (270, 480, (11, 11), 3, (14.0, -11.0)), (400, 400, (31, 17), 5, (-9.5, 6.25))]:
        prev = _smooth_scene(rows, cols, 0, 0)
        nxt = _smooth_scene(rows, cols, shift[0], shift[1])
        rng = np.random.default_rng(rows + win[0])
        pts = np.c_[rng.uniform(-3, cols + 3, 200), rng.uniform(-3, rows + 3, 200)].astype(np.float32)
        want_p, want_s = oracle.pyrlk(prev, nxt, pts, win=win, max_level=lv)
        got_p, got_s = ctx.pyrlk(_gpu(prev), _gpu(nxt), pts, win=win, max_level=lv)
        assert np.array_equal(got_s, want_s), (rows, cols, win)
        assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32)), (rows, cols, win)


@pytest.mark.parametrize("channel", [-1, -2])
@pytest.mark.parametrize("src_size,dst_size", [((2160, 3840), (270, 480)), ((1080, 1920), (270, 480)), ((720, 1280), (270, 480)),
                                                 ((270, 480), (270, 480)), ((90, 130), (45, 65)), ((61, 97), (20, 31))])
def test_luma_area_resize_bgr_rgb(ctx, oracle, src_size, dst_size, channel):
    """a3 second half: cvtColor(BGR2GRAY / RGB2GRAY) fused with the INTER_AREA downscale."""
    rng = np.random.default_rng(src_size[0] + channel)
    frame = rng.integers(0, 256, (*src_size, 3), dtype=np.uint8)
    want = oracle.luma_area_resize(frame, *dst_size, channel=channel)
    got = ctx.luma_area_resize(_gpu(frame), *dst_size, channel=channel)
    ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)


def _swap_erase(arrays, keep):
    """fast_filter (Functions/Container.tpp:97-121): back to front, swap a dropped element with the last kept one."""
    arrays = [a.copy() for a in arrays]
    m = len(keep)
    for k in range(len(keep) - 1, -1, -1):
        if not keep[k]:
            m -= 1
            for a in arrays:
                a[[k, m]] = a[[m, k]]
    return [a[:m] for a in arrays]


@pytest.mark.parametrize("n", [1, 2, 7, 64, 257, 1856, 3728, 4096])
def test_gpu_fast_filter_matches_swap_erase_order(ctx, n):
    rng = np.random.default_rng(n)
    prev = rng.uniform(0, 480, (n, 2)).astype(np.float32); matched = rng.uniform(0, 480, (n, 2)).astype(np.float32)
    patterns = [np.ones(n, np.uint8), np.zeros(n, np.uint8), (np.arange(n) % 2).astype(np.uint8), ((np.arange(n) + 1) % 2).astype(np.uint8),
                (rng.random(n) < 0.9).astype(np.uint8), (rng.random(n) < 0.3).astype(np.uint8),
                (np.arange(n) < n // 3).astype(np.uint8), (np.arange(n) >= n // 3).astype(np.uint8),      # all holes in the tail / in the front
                (rng.random(n) < 0.5).astype(np.uint8) * 255]
    for keep in patterns:
        want_p, want_m = _swap_erase([prev, matched], keep)
        got_p, got_m = ctx.fast_filter(prev, matched, keep)
        assert len(got_p) == len(want_p)
        assert np.array_equal(got_p, want_p) and np.array_equal(got_m, want_m)
