"""The oracle's FrameIngest restatement for every OBS video format (oracle/ingest.cpp::lvko_ingest_obs / lvko_egress_obs; reference
Modules/OBS-Plugin/Interop/FrameIngest.cpp:36-75,476-753) against an independent numpy restatement, scipy's resampling geometry and the
round trips the formats allow.  CPU only."""
import numpy as np
import pytest
from scipy import ndimage

FORMATS = ["I420", "NV12", "YVYU", "YUY2", "UYVY", "RGBA", "BGRA", "BGRX", "Y800", "I444", "BGR3", "I422", "I40A", "I42A", "YUVA", "AYUV"]
SIZES = [(6, 8), (34, 50), (270, 480), (2, 2)]


def _planes(oracle, fmt, rows, cols, seed=0):
    rng = np.random.default_rng(seed + rows * 7 + cols)
    return [rng.integers(0, 256, sh, dtype=np.uint8) for sh in oracle.obs_plane_shapes(fmt, rows, cols)]


def np_upsample_x2_cols(c):
    """cv::resize(8U, (2 * w, h), INTER_LINEAR) of a [h, w] plane as OpenCV's CPU path computes it: 11-bit coefficients, horizontal pass in int,
    vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2 with (b0, b1) = (2048, 0)."""
    h, w = c.shape
    c = c.astype(np.int64)
    out = np.zeros((h, 2 * w), np.int64)
    k = np.arange(w)
    left = np.maximum(k - 1, 0); right = np.minimum(k + 1, w - 1)
    even = c[:, left] * 512 + c * 1536                 # x = 2k: 0.25 / 0.75 between columns k - 1, k
    even[:, 0] = c[:, 0] * 2048                        # left edge: the single sample
    odd = c * 1536 + c[:, right] * 512                 # x = 2k + 1: 0.75 / 0.25 between k, k + 1
    odd[:, w - 1] = c[:, w - 1] * 2048                 # right edge
    out[:, 0::2] = even; out[:, 1::2] = odd
    return ((((2048 * (out >> 4)) >> 16) + 2) >> 2).astype(np.uint8)


def np_ingest(fmt, planes):
    p = planes
    if fmt in ("I444", "YUVA"):
        return np.stack(p[:3], -1)
    if fmt in ("I422", "I42A"):
        return np.stack([p[0], np_upsample_x2_cols(p[1]), np_upsample_x2_cols(p[2])], -1)
    if fmt in ("YUY2", "YVYU", "UYVY"):
        raw = p[0]
        yb, cb = (1, 0) if fmt == "UYVY" else (0, 1)
        y = raw[..., yb]
        first, second = raw[:, 0::2, cb], raw[:, 1::2, cb]
        u, v = (first, second) if fmt != "YVYU" else (second, first)
        return np.stack([y, np_upsample_x2_cols(u), np_upsample_x2_cols(v)], -1)
    if fmt == "AYUV":
        return p[0][..., 1:4].copy()
    if fmt == "Y800":
        return p[0].copy()
    if fmt == "BGR3":
        return p[0].copy()
    if fmt in ("RGBA", "BGRA", "BGRX"):
        rows, cols = p[0].shape[:2]
        return p[0].reshape(-1)[: rows * cols * 3].reshape(rows, cols, 3).copy()
    raise ValueError(fmt)


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("fmt", [f for f in FORMATS if f not in ("I420", "NV12", "I40A")])
def test_ingest_equals_numpy_restatement(oracle, fmt, size):
    rows, cols = size
    planes = _planes(oracle, fmt, rows, cols)
    assert np.array_equal(oracle.ingest_obs(fmt, planes), np_ingest(fmt, planes))


@pytest.mark.parametrize("size", SIZES)
def test_420_formats_go_through_the_pinned_420_path(oracle, size):
    rows, cols = size
    planes = _planes(oracle, "I420", rows, cols)
    want = oracle.ingest_yuv420(*planes)
    assert np.array_equal(oracle.ingest_obs("I420", planes), want) and np.array_equal(oracle.ingest_obs("I40A", planes), want)
    uv = np.ascontiguousarray(np.stack(planes[1:], -1))
    assert np.array_equal(oracle.ingest_obs("NV12", [planes[0], uv]), want)
    y, u, v = oracle.egress_obs("I420", want)
    y2, u2, v2 = oracle.egress_yuv420(want)
    assert np.array_equal(y, y2) and np.array_equal(u, u2) and np.array_equal(v, v2)
    yn, uvn = oracle.egress_obs("NV12", want)
    assert np.array_equal(uvn, np.stack([u2, v2], -1))


@pytest.mark.parametrize("size", [(34, 50), (270, 480)])
def test_422_chroma_upsampling_geometry_equals_scipy_zoom(oracle, size):
    """the horizontal 2x of I422 against scipy.ndimage.zoom(order=1, grid_mode=True, mode="nearest"): exact values are multiples of 1/4; OpenCV's fixed point
    drops the low four bits of the row sum before its rounding constant, so the result is floor(z + 0.5) except that .5 may round down."""
    rows, cols = size
    planes = _planes(oracle, "I422", rows, cols, seed=3)
    got = oracle.ingest_obs("I422", planes)
    for ch in (1, 2):
        z = ndimage.zoom(planes[ch].astype(np.float64), (1, 2), order=1, mode="nearest", grid_mode=True)
        g = got[..., ch].astype(np.float64)
        frac = z - np.floor(z)
        tie = frac == 0.5
        assert np.array_equal(g[~tie], np.floor(z[~tie] + 0.5))
        assert np.isin(g[tie] - np.floor(z[tie]), (0.0, 1.0)).all()


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("fmt", ["I422", "I42A", "YUY2", "YVYU", "UYVY"])
def test_422_egress_is_the_pairwise_mean_rounded_half_to_even(oracle, fmt, size):
    """cv::resize(Size(), 0.5, 1.0, INTER_AREA) = resizeAreaFast_'s generic loop: saturate_cast<uchar>((a + b) * 0.5f), i.e. numpy's rint of the mean."""
    rows, cols = size
    rng = np.random.default_rng(rows + cols)
    frame = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    out = oracle.egress_obs(fmt, frame)
    u = np.rint((frame[:, 0::2, 1].astype(np.float64) + frame[:, 1::2, 1]) / 2).astype(np.uint8)
    v = np.rint((frame[:, 0::2, 2].astype(np.float64) + frame[:, 1::2, 2]) / 2).astype(np.uint8)
    if fmt in ("I422", "I42A"):
        assert np.array_equal(out[0], frame[..., 0]) and np.array_equal(out[1], u) and np.array_equal(out[2], v)
        return
    raw = out[0]
    yb, cb = (1, 0) if fmt == "UYVY" else (0, 1)
    assert np.array_equal(raw[..., yb], frame[..., 0])
    first, second = (u, v) if fmt != "YVYU" else (v, u)
    assert np.array_equal(raw[:, 0::2, cb], first) and np.array_equal(raw[:, 1::2, cb], second)
    # the ties exist and go to the even neighbour
    s = frame[:, 0::2, 1].astype(int) + frame[:, 1::2, 1]
    if s.size > 16:
        odd = (s & 1) == 1
        assert odd.any() and ((u[odd] & 1) == 0).all()


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("fmt", ["I444", "YUVA", "AYUV", "BGR3", "Y800", "RGBA", "BGRA", "BGRX"])
def test_lossless_formats_round_trip(oracle, fmt, size):
    rows, cols = size
    planes = _planes(oracle, fmt, rows, cols, seed=5)
    frame = oracle.ingest_obs(fmt, planes)
    # to_obs into the frame's own planes: bytes the reference does not write stay what they were
    back = oracle.egress_obs(fmt, frame, planes=[p.copy() for p in planes])
    if fmt == "AYUV":
        assert (back[0][..., 0] == 255).all() and np.array_equal(back[0][..., 1:], planes[0][..., 1:])      # P444Ingest::to_obs sets alpha to 255
    else:
        for a, b in zip(planes, back):
            assert np.array_equal(a, b)
    if fmt in ("RGBA", "BGRA", "BGRX"):
        # DirectIngest moves rows * cols * 3 BYTES of the 4-byte pixels (FrameIngest.cpp:743-753): the frame is the byte stream re-cut, not the colour planes
        assert not np.array_equal(frame, planes[0][..., :3]) or rows * cols < 4
        assert np.array_equal(frame.reshape(-1), planes[0].reshape(-1)[: rows * cols * 3])


def test_formats_that_subsample_reject_odd_widths(oracle):
    rng = np.random.default_rng(0)
    lib = oracle.lib
    import ctypes as c
    u8p = c.POINTER(c.c_uint8)
    y = rng.integers(0, 256, (4, 5), dtype=np.uint8); ch = rng.integers(0, 256, (4, 3), dtype=np.uint8); dst = np.zeros((4, 5, 3), np.uint8)
    ptrs = (u8p * 3)(y.ctypes.data_as(u8p), ch.ctypes.data_as(u8p), ch.ctypes.data_as(u8p)); steps = (c.c_int * 3)(5, 3, 3)
    lib.lvko_ingest_obs.restype = c.c_int
    lib.lvko_ingest_obs.argtypes = [c.c_int, u8p * 3, c.c_int * 3, c.c_int, c.c_int, u8p, c.c_int]
    assert lib.lvko_ingest_obs(12, ptrs, steps, 4, 5, dst.ctypes.data_as(u8p), 15) != 0      # I422, 5 columns
    assert lib.lvko_ingest_obs(99, ptrs, steps, 4, 4, dst.ctypes.data_as(u8p), 15) != 0      # unknown format
