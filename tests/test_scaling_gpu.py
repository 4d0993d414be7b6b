"""GPU parity of ScalingFilter's two kernels: EASU upscale and RCAS sharpening, against the CPU oracle
through the C-ABI.  Bar: bit-exact."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _to_gpu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _assert_same(got, want, what):
    got = got.cpu().numpy()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not np.array_equal(got, want):
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        ys, xs = np.nonzero(d.max(axis=2))
        raise AssertionError(f"{what}: {len(ys)} pixels differ, max |d| = {d.max()}, first at (x={xs[0]}, y={ys[0]}): "
                             f"gpu={got[ys[0], xs[0]]} oracle={want[ys[0], xs[0]]}")


@pytest.mark.parametrize("yuv", [True, False])
@pytest.mark.parametrize("src_size,dst_size", [((72, 96), (192, 144)), ((67, 131), (200, 101)), ((270, 480), (1280, 720)),
                                               ((48, 64), (64, 49)), ((5, 9), (31, 17)), ((8, 8), (9, 8))])
def test_upscale_bit_exact(ctx, oracle, yuv, src_size, dst_size):
    rows, cols = src_size
    src = synth.textured_frame(rows, cols, seed=rows + cols) if rows >= 32 else \
        np.random.default_rng(rows).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    want = oracle.upscale(src, dst_size, yuv=yuv)
    got = ctx.upscale(_to_gpu(src), dst_size, yuv=yuv)
    ctx.sync()
    _assert_same(got, want, f"upscale {src_size} -> {dst_size} yuv={yuv}")


def test_upscale_same_size_copies_and_downscale_is_rejected(ctx):
    import torch
    from livevisionkit_amd import LvkHipError
    src = _to_gpu(synth.textured_frame(40, 56, seed=3))
    out = ctx.upscale(src, (56, 40))
    ctx.sync()
    assert torch.equal(out, src)
    with pytest.raises(LvkHipError):
        ctx.upscale(src, (55, 40))                                     # Image.cpp:157
    with pytest.raises(LvkHipError):
        ctx.upscale(src, (56, 40), out=src)                            # aliasing


def test_upscale_padded_pitches(ctx, oracle):
    import torch
    src = synth.textured_frame(60, 70, seed=8)
    dsrc_full = torch.zeros((60, 80, 3), dtype=torch.uint8, device="cuda")
    dsrc_full[:, :70] = _to_gpu(src)
    ddst_full = torch.zeros((90, 120, 3), dtype=torch.uint8, device="cuda")
    got = ctx.upscale(dsrc_full[:, :70], (105, 90), out=ddst_full[:, :105])
    ctx.sync()
    _assert_same(got, oracle.upscale(src, (105, 90)), "padded pitches")
    assert int(ddst_full[:, 105:].max()) == 0


@pytest.mark.parametrize("sharpness", [0.0, 0.7, 0.8, 1.0])
@pytest.mark.parametrize("size", [(72, 96), (67, 131), (270, 480), (3, 3), (2, 9), (9, 2), (1, 1), (17, 257), (33, 1030)])
def test_sharpen_bit_exact(ctx, oracle, sharpness, size):
    rows, cols = size
    src = synth.textured_frame(rows, cols, seed=rows * 3 + cols) if rows >= 32 else \
        np.random.default_rng(rows * 7 + cols).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    if rows > 40:
        src[5:9, 5:30] = 0; src[20:24, 40:70] = 255                    # saturated rings: the 0 * inf limiters
    want = oracle.sharpen(src, sharpness)
    got = ctx.sharpen(_to_gpu(src), sharpness)
    ctx.sync()
    _assert_same(got, want, f"sharpen {size} s={sharpness}")


def test_sharpen_all_byte_values_and_extremes(ctx, oracle):
    """Every (ring extremum, centre) byte combination along one axis: pins the two reciprocal tables."""
    k = np.arange(256, dtype=np.uint8)
    src = np.zeros((3 * 256, 3 * 256, 3), np.uint8)
    ring = np.repeat(np.repeat(k[:, None], 256, axis=1), 3, axis=0).repeat(3, axis=1)      # ring value varies with y block
    src[...] = ring[..., None]
    src[1::3, 1::3, :] = k[None, :, None]                                                    # centre value varies with x block
    src[1::3, 1::3, 1] = 255 - src[1::3, 1::3, 1]
    for s in (0.3, 1.0):
        got = ctx.sharpen(_to_gpu(src), s)
        ctx.sync()
        _assert_same(got, oracle.sharpen(src, s), f"byte sweep s={s}")


def test_sharpen_padded_pitch_and_argument_checks(ctx, oracle):
    import torch
    from livevisionkit_amd import LvkHipError
    src = synth.textured_frame(50, 61, seed=9)
    dsrc_full = torch.zeros((50, 70, 3), dtype=torch.uint8, device="cuda")
    dsrc_full[:, :61] = _to_gpu(src)
    ddst_full = torch.zeros((50, 64, 3), dtype=torch.uint8, device="cuda")
    got = ctx.sharpen(dsrc_full[:, :61], 0.6, out=ddst_full[:, :61])
    ctx.sync()
    _assert_same(got, oracle.sharpen(src, 0.6), "padded pitch")
    assert int(ddst_full[:, 61:].max()) == 0
    with pytest.raises(LvkHipError):
        ctx.sharpen(dsrc_full, 1.01)                                   # LVK_ASSERT_01, Image.cpp:210
    with pytest.raises(LvkHipError):
        ctx.sharpen(dsrc_full, 0.5, out=dsrc_full)                     # in place is a race in the reference; rejected here


def test_scaling_filter_chain_1080p_to_4k(ctx, oracle):
    """ScalingFilter::filter (ScalingFilter.cpp:52-59) at its headline use: 1080p -> 4K upscale + sharpen."""
    src = synth.textured_frame(1080, 1920, seed=31)
    up = ctx.upscale(_to_gpu(src), (3840, 2160), yuv=True)
    out = ctx.sharpen(up, 0.8)
    ctx.sync()
    want_up = oracle.upscale(src, (3840, 2160), yuv=True, nthreads=32)
    _assert_same(up, want_up, "upscale 1080p -> 4K")
    _assert_same(out, oracle.sharpen(want_up, 0.8, nthreads=32), "sharpen 4K")
