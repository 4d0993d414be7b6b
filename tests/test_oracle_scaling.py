"""CPU tests that pin the oracle's EASU upscale and RCAS sharpening (ScalingFilter half; no GPU):
known-answer cases and a second, independent numpy restatement (tests/np_easu.py)."""
import numpy as np
import pytest

from tests import np_easu, synth


def test_upscale_same_size_is_a_copy(oracle):
    src = synth.textured_frame(36, 52, seed=11)
    assert np.array_equal(oracle.upscale(src, (52, 36)), src)          # Image.cpp:162-166


def test_upscale_integer_factor_hits_source_pixels(oracle):
    """x2: even destination pixels land on pp = 0 of a source pixel.  In a flat patch only the centre tap has weight, so the value is
    the source pushed through the * norm * 255 truncation; borders are nearest copies of trunc(dst * rscale)."""
    src = synth.textured_frame(40, 48, seed=12)
    src[8:32, 8:40] = (60, 140, 220)
    out = oracle.upscale(src, (96, 80), yuv=True)
    v = (np.array([60, 140, 220], np.float32) * np.float32(0.00392156862) * np.float32(255.0)).astype(np.int32).astype(np.uint8)
    assert (out[24:56, 24:72] == v).all()
    assert np.array_equal(out[0, ::2], src[0]) and np.array_equal(out[1, ::2], src[0])      # sy == 0 -> nearest
    assert np.array_equal(out[::2, 0], src[:, 0])
    assert np.array_equal(out[-8:, 20], src[-4:, 10].repeat(2, axis=0))                       # sy >= rows - 4 -> nearest


def test_upscale_rejects_downscale(oracle):
    src = synth.textured_frame(32, 32, seed=13)
    with pytest.raises(AssertionError):
        oracle.upscale(src, (31, 32))                                  # Image.cpp:157


@pytest.mark.parametrize("yuv", [True, False])
@pytest.mark.parametrize("size", [(96, 72), (100, 61), (65, 49), (200, 48)])
def test_upscale_matches_independent_numpy_restatement(oracle, yuv, size):
    src = synth.textured_frame(48, 64, seed=14)
    a = oracle.upscale(src, size, yuv=yuv)
    b = np_easu.upscale(src, size, yuv)
    diff = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3


def test_sharpen_border_is_copied_and_flat_is_fixed_point(oracle):
    src = synth.textured_frame(40, 56, seed=15)
    src[10:30, 10:40] = (90, 91, 92)
    out = oracle.sharpen(src, 0.8)
    assert np.array_equal(out[0], src[0]) and np.array_equal(out[-1], src[-1])
    assert np.array_equal(out[:, 0], src[:, 0]) and np.array_equal(out[:, -1], src[:, -1])
    inner = out[12:28, 12:38].astype(np.int32)
    assert (np.abs(inner - np.array([90, 91, 92])) <= 1).all() and (inner <= np.array([90, 91, 92])).all()   # medium rcp under-estimates


def test_sharpen_black_and_white_rings_do_not_poison(oracle):
    """A ring of all 0 (or all 1) makes 0 * inf = NaN in one limiter; fmax drops it (FSR.cl:513-521)."""
    src = np.zeros((9, 9, 3), np.uint8)
    src[4, 4] = 200
    out = oracle.sharpen(src, 1.0)
    assert out[4, 4].min() >= 199 and out[2, 2].max() == 0
    src = np.full((9, 9, 3), 255, np.uint8)
    src[4, 4] = 20
    out = oracle.sharpen(src, 1.0)
    assert out[4, 4].max() <= 20 and out[2, 2].min() >= 254


def test_sharpen_increases_local_contrast(oracle):
    src = np.full((16, 16, 3), 100, np.uint8)
    src[:, 8:] = 160
    soft, hard = oracle.sharpen(src, 0.0), oracle.sharpen(src, 1.0)
    # under/over-shoot at the two sides of the step grows with sharpness, stays inside [0, 255]
    assert hard[8, 7, 0] <= soft[8, 7, 0] <= 100 and hard[8, 8, 0] >= soft[8, 8, 0] >= 159
    assert hard[8, 7, 0] < 100


@pytest.mark.parametrize("sharpness", [0.0, 0.7, 1.0])
def test_sharpen_matches_independent_numpy_restatement(oracle, sharpness):
    src = synth.textured_frame(64, 80, seed=16)
    src[5:9, 5:30] = 0; src[20:24, 40:70] = 255                       # saturated rings
    a = oracle.sharpen(src, sharpness)
    b = np_easu.sharpen(src, sharpness)
    diff = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3


def test_sharpen_rejects_out_of_range_sharpness(oracle):
    src = synth.textured_frame(16, 16, seed=17)
    with pytest.raises(AssertionError):
        oracle.sharpen(src, 1.5)                                       # LVK_ASSERT_01, Image.cpp:210
