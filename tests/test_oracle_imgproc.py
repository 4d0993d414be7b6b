"""CPU tests pinning the oracle's tracker image operations (INTER_AREA, pyrDown, Scharr, FAST-9/16, PyrLK) against
known answers and independent numpy restatements of the OpenCV 4.8 semantics."""
import numpy as np
import pytest

from tests import synth


def _round_half_even(x):
    return np.rint(x)


def test_area_resize_integer_scales_exact(oracle):
    rng = np.random.default_rng(0)
    for scale in (3, 4, 8):
        src = rng.integers(0, 256, (scale * 9, scale * 13), dtype=np.uint8)
        got = oracle.luma_area_resize(src, 9, 13)
        s = src.reshape(9, scale, 13, scale).astype(np.int64).sum(axis=(1, 3))
        want = _round_half_even((s.astype(np.float32) * np.float32(1.0 / (scale * scale))).astype(np.float32))
        assert np.array_equal(got, want.astype(np.uint8))


def test_area_resize_2x2_rounds_half_up(oracle):
    src = np.array([[1, 2], [0, 0]], np.uint8).repeat(1, 0)          # sum 3 -> (3+2)>>2 = 1 ; 0.75 rounds to 1
    assert oracle.luma_area_resize(src, 1, 1)[0, 0] == 1
    src = np.array([[1, 1], [0, 0]], np.uint8)                         # sum 2 -> 0.5: (2+2)>>2 = 1 (half-even would give 0)
    assert oracle.luma_area_resize(src, 1, 1)[0, 0] == 1


def test_area_resize_packed_channel0_and_planar_agree(oracle):
    frame = synth.textured_frame(64 * 4, 96 * 4, seed=3)
    a = oracle.luma_area_resize(frame, 64, 96)
    b = oracle.luma_area_resize(np.ascontiguousarray(frame[..., 0]), 64, 96)
    assert np.array_equal(a, b)


def test_area_resize_non_integer_matches_area_average(oracle):
    """720p -> 480x270 (scale 2.667): decimate-alpha tables == exact box-area average up to float rounding."""
    src = synth.textured_frame(720, 1280, seed=5, channels=1)
    got = oracle.luma_area_resize(src, 270, 480).astype(np.float64)
    # exact area average via integral image in float64
    ii = np.zeros((721, 1281)); ii[1:, 1:] = src.astype(np.float64).cumsum(0).cumsum(1)
    def integral_at(y, x):   # bilinear-free: exact for the piecewise constant image using fractional coverage
        y0 = np.floor(y).astype(int); x0 = np.floor(x).astype(int)
        fy = y - y0; fx = x - x0
        y1 = np.minimum(y0 + 1, 720); x1 = np.minimum(x0 + 1, 1280)
        return (ii[y0][:, x0] * np.outer(1 - fy, 1 - fx) + ii[y0][:, x1] * np.outer(1 - fy, fx)
                + ii[y1][:, x0] * np.outer(fy, 1 - fx) + ii[y1][:, x1] * np.outer(fy, fx))
    sy = np.arange(271) * (720 / 270); sx = np.arange(481) * (1280 / 480)
    I = integral_at(sy, sx)
    want = (I[1:, 1:] - I[:-1, 1:] - I[1:, :-1] + I[:-1, :-1]) / ((720 / 270) * (1280 / 480))
    assert np.abs(got - want).max() <= 0.51


def _np_pyr_down(img):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    p = np.pad(img.astype(np.int64), 2, mode="reflect")
    rows, cols = img.shape
    dr, dc = (rows + 1) // 2, (cols + 1) // 2
    h = sum(k[i] * p[:, i:i + 2 * dc:2][:, :dc] for i in range(5))
    v = sum(k[i] * h[i:i + 2 * dr:2][:dr] for i in range(5))
    return ((v + 128) >> 8).astype(np.uint8)


@pytest.mark.parametrize("shape", [(270, 480), (135, 240), (68, 120), (33, 47)])
def test_pyr_down_matches_numpy(oracle, shape):
    img = np.random.default_rng(shape[0]).integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(oracle.pyr_down(img), _np_pyr_down(img))


def test_pyramid_level_sizes(oracle):
    assert oracle.pyramid_levels(270, 480) == [(270, 480), (135, 240), (68, 120), (34, 60)]
    assert oracle.pyramid_levels(256, 256) == [(256, 256), (128, 128), (64, 64), (32, 32)]
    assert oracle.pyramid_levels(40, 40) == [(40, 40), (20, 20)]          # next level (10x10) <= window -> stop


def test_scharr_matches_numpy(oracle):
    img = np.random.default_rng(1).integers(0, 256, (37, 53), dtype=np.uint8)
    p = np.pad(img.astype(np.int64), 1, mode="reflect")
    t0 = 3 * (p[:-2] + p[2:]) + 10 * p[1:-1]            # vertical smooth   (rows x cols+2)
    t1 = p[2:] - p[:-2]                                  # vertical diff
    ix = t0[:, 2:] - t0[:, :-2]
    iy = 3 * (t1[:, 2:] + t1[:, :-2]) + 10 * t1[:, 1:-1]
    d = oracle.scharr_deriv(img)
    assert np.array_equal(d[..., 0], ix.astype(np.int16)) and np.array_equal(d[..., 1], iy.astype(np.int16))


def _brute_fast(img, t):
    off = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    rows, cols = img.shape
    score = np.zeros((rows, cols), np.int32)
    im = img.astype(np.int32)
    for y in range(3, rows - 3):
        for x in range(3, cols - 3):
            d = [im[y, x] - im[y + dy, x + dx] for dx, dy in off]
            best = 0
            for s in range(16):
                arc = [d[(s + k) % 16] for k in range(9)]
                best = max(best, min(arc), min(-a for a in arc))
            if best > t:
                score[y, x] = best - 1
    out = []
    for y in range(3, rows - 3):
        for x in range(3, cols - 3):
            s = score[y, x]
            if s > 0:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                if (s > nb).all():
                    out.append((x, y, s))
    return np.array(out, np.int32).reshape(-1, 3)


def test_fast_matches_bruteforce_definition(oracle):
    img = synth.textured_frame(48, 64, seed=9, channels=1)
    for t in (10, 25, 60):
        got = oracle.fast(img, t)
        want = _brute_fast(img, t)
        assert np.array_equal(got, want), (t, len(got), len(want))
    assert len(oracle.fast(img, 10)) > 5


def test_fast_roi_edge_is_image_edge(oracle):
    img = synth.textured_frame(60, 80, seed=10, channels=1)
    roi = (40, 0, 40, 60)
    got = oracle.fast(img, 10, roi=roi)
    want = _brute_fast(np.ascontiguousarray(img[:, 40:80]), 10)
    assert np.array_equal(got, want)
    assert got[:, 0].min() >= 3 and got[:, 0].max() <= 36


def _smooth_scene(rows, cols, dx, dy):
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    xx = xx + dx; yy = yy + dy
    v = (128 + 50 * np.sin(xx * 0.11 + 0.3) * np.cos(yy * 0.13) + 40 * np.sin((xx + 2 * yy) * 0.07)
         + 30 * np.cos((xx - yy) * 0.05 + 1.0) + 15 * np.sin(xx * 0.31) * np.sin(yy * 0.29))
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shift", [(0.0, 0.0), (1.3, -0.7), (-3.6, 2.2), (6.5, 4.25)])
def test_pyrlk_recovers_known_translation(oracle, shift):
    rows, cols = 270, 480
    prev = _smooth_scene(rows, cols, 0, 0)
    nxt = _smooth_scene(rows, cols, -shift[0], -shift[1])          # content moves by +shift
    rng = np.random.default_rng(2)
    pts = np.c_[rng.uniform(30, cols - 30, 200), rng.uniform(30, rows - 30, 200)].astype(np.float32)
    out, st = oracle.pyrlk(prev, nxt, pts)
    ok = st == 1
    assert ok.mean() > 0.95
    err = np.abs(out[ok] - (pts[ok] + np.array(shift, np.float32)))
    assert np.median(err) < 0.05 and np.percentile(err, 95) < 0.3


def test_pyrlk_status_rules(oracle):
    rows, cols = 135, 240
    img = _smooth_scene(rows, cols, 0, 0)
    flat = np.full((rows, cols), 77, np.uint8)
    pts = np.array([[50, 50], [-20.0, 40.0], [239.9, 134.9], [500.0, 500.0]], np.float32)
    out, st = oracle.pyrlk(img, img, pts)
    assert st[0] == 1 and np.abs(out[0] - pts[0]).max() < 1e-3
    assert st[1] == 0 and st[3] == 0                                # window origin left of -winSize / beyond the image
    out, st = oracle.pyrlk(flat, flat, pts[:1])
    assert st[0] == 0                                               # minEig below threshold on a flat patch


# ---- YUV420 <-> packed 444 ------------------------------------------------------------
def test_ingest_constant_chroma_and_luma_passthrough(oracle):
    rng = np.random.default_rng(0)
    y = rng.integers(0, 256, (36, 48), dtype=np.uint8)
    u = np.full((18, 24), 77, np.uint8); v = np.full((18, 24), 201, np.uint8)
    f = oracle.ingest_yuv420(y, u, v)
    assert np.array_equal(f[..., 0], y) and (f[..., 1] == 77).all() and (f[..., 2] == 201).all()


def test_ingest_bilinear_phase_and_roundtrip(oracle):
    """2x upsampling samples at phases .25/.75 (fx = dx/2 - 0.25); egress (2x2 area mean) of ingest(x) stays within 1 LSB of x
    away from the borders for a smooth chroma plane."""
    yy, xx = np.mgrid[0:20, 0:30]
    u = (40 + 5 * xx + 2 * yy).astype(np.uint8); v = (200 - 3 * xx).astype(np.uint8)
    y = np.zeros((40, 60), np.uint8)
    f = oracle.ingest_yuv420(y, u, v)
    # interior column 2k+1 = .75*u[k] + .25*u[k+1] (exact here: multiples of 5 * .25 ...), checked loosely
    exp = 0.75 * u[:, :-1].astype(float) + 0.25 * u[:, 1:].astype(float)
    assert np.abs(f[1::2, 1:-1:2, 1][:-1] - (0.75 * exp[:-1] + 0.25 * exp[1:])).max() <= 1.0
    y2, u2, v2 = oracle.egress_yuv420(f)
    assert np.array_equal(y2, y)
    assert np.abs(u2[1:-1, 1:-1].astype(int) - u[1:-1, 1:-1]).max() <= 1 and np.abs(v2[1:-1, 1:-1].astype(int) - v[1:-1, 1:-1]).max() <= 1


def test_nv12_equals_i420(oracle):
    rng = np.random.default_rng(5)
    y = rng.integers(0, 256, (32, 40), dtype=np.uint8)
    u = rng.integers(0, 256, (16, 20), dtype=np.uint8); v = rng.integers(0, 256, (16, 20), dtype=np.uint8)
    a = oracle.ingest_yuv420(y, u, v)
    b = oracle.ingest_yuv420(y, np.stack([u, v], -1))
    assert np.array_equal(a, b)
    ya, ua, va = oracle.egress_yuv420(a)
    yb, uvb = oracle.egress_yuv420(a, nv12=True)
    assert np.array_equal(ya, yb) and np.array_equal(ua, uvb[..., 0]) and np.array_equal(va, uvb[..., 1])


def test_bgr_and_rgb_gray_known_answers(oracle):
    """cv::cvtColor(BGR2GRAY / RGB2GRAY) on 8U (OpenCV 4.8 RGB2Gray<uchar>: 15-bit coefficients 3735 / 19235 / 9798, round to
    nearest): primaries, white, and agreement with the float luma 0.114 B + 0.587 G + 0.299 R to 1 LSB."""
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [10, 200, 30]]], np.uint8)
    bgr = oracle.luma_area_resize(px, 1, 6, channel=-1)[0]
    rgb = oracle.luma_area_resize(px, 1, 6, channel=-2)[0]
    assert list(bgr[:5]) == [29, 150, 76, 255, 0] and list(rgb[:5]) == [76, 150, 29, 255, 0]
    assert bgr[5] == (10 * 3735 + 200 * 19235 + 30 * 9798 + 16384) >> 15
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    g = oracle.luma_area_resize(img, 32, 48, channel=-1).astype(np.float64)
    ref = 0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]
    assert np.abs(g - ref).max() <= 1.0
    assert np.array_equal(oracle.luma_area_resize(img[..., ::-1], 32, 48, channel=-2), g.astype(np.uint8))
    # downscale of the gray image == gray then box average
    small = oracle.luma_area_resize(img, 8, 12, channel=-1)
    want = oracle.luma_area_resize(np.ascontiguousarray(g.astype(np.uint8)), 8, 12)
    assert np.array_equal(small, want)


# ---- row a7: a second, independent statement of the tracker (tests/np_pyrlk.py, written from SURVEY App. A.3 / A.4) ------------------
@pytest.mark.parametrize("case", ["shifted scene", "rotated texture", "small frame, 2 levels"])
def test_pyrlk_matches_independent_numpy_restatement(oracle, case):
    from tests import np_pyrlk
    rng = np.random.default_rng(len(case))
    if case == "shifted scene":
        rows, cols = 135, 240
        prev = _smooth_scene(rows, cols, 0, 0); nxt = _smooth_scene(rows, cols, -2.3, 1.4)
    elif case == "rotated texture":
        rows, cols = 120, 160
        yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
        tex = lambda x, y: 128 + 60 * np.sin(0.21 * x + 0.07 * y) * np.cos(0.17 * y - 0.05 * x) + 30 * np.sin(0.43 * x) * np.sin(0.39 * y)
        th = 0.02
        xr = np.cos(th) * (xx - 80) - np.sin(th) * (yy - 60) + 80 + 0.8; yr = np.sin(th) * (xx - 80) + np.cos(th) * (yy - 60) + 60 - 0.5
        prev = np.clip(np.rint(tex(xx, yy)), 0, 255).astype(np.uint8); nxt = np.clip(np.rint(tex(xr, yr)), 0, 255).astype(np.uint8)
        prev[40:60, 50:90] = 200; nxt[41:61, 52:92] = 200                       # a flat rectangle: corners to track, and a flat interior (minEig)
    else:
        rows, cols = 44, 52                                                      # the third level would be <= the window: two levels only
        prev = _smooth_scene(rows, cols, 0, 0); nxt = _smooth_scene(rows, cols, 0.6, -0.4)
    pts = np.c_[rng.uniform(-8, cols + 6, 70), rng.uniform(-8, rows + 6, 70)].astype(np.float32)      # incl. points whose window leaves the frame
    pts[:6] = [[0, 0], [cols - 1, rows - 1], [5.5, 5.5], [cols - 6.5, 5.25], [cols / 2, rows / 2], [70.25, 50.75]]
    want, wst = oracle.pyrlk(prev, nxt, pts)
    got, gst = np_pyrlk.calc(prev, nxt, pts)
    assert np.array_equal(wst, gst)
    assert 0.2 < wst.mean() < 1.0                                               # both outcomes are exercised
    ok = wst == 1
    assert np.array_equal(want[ok].view(np.uint32), got[ok].view(np.uint32)), np.abs(want[ok] - got[ok]).max()


def _np_area_tab(ssize, dsize):
    """computeResizeAreaTab, per destination index: [(source index, float32 weight)] in table order."""
    scale = ssize / dsize
    out = []
    for dx in range(dsize):
        fsx1 = dx * scale; fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = int(np.ceil(fsx1)); sx2 = min(int(np.floor(fsx2)), ssize - 1); sx1 = min(sx1, sx2)
        taps = []
        if sx1 - fsx1 > 1e-3:
            taps.append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            taps.append((sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            taps.append((sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        out.append(taps)
    return out


@pytest.mark.parametrize("src_shape,dst_shape", [((144, 256), (27, 48)), ((96, 171), (27, 48)), ((120, 192), (27, 48)), ((72, 128), (27, 48)), ((61, 97), (60, 96))])
def test_area_resize_fractional_matches_independent_numpy_restatement(oracle, src_shape, dst_shape):
    """resizeArea_ for non-integer scales (1440p-, 1200p-, 720p-like ratios and a barely-above-one scale), restated with whole-row float32
    arithmetic in the table's order: horizontal pass per source row (buf += S * alpha), vertical accumulation (sum += beta * buf), round half
    to even -- bit-identical to oracle/imgproc.cpp."""
    f32 = np.float32
    src = np.random.default_rng(src_shape[1]).integers(0, 256, src_shape, dtype=np.uint8)
    xt, yt = _np_area_tab(src_shape[1], dst_shape[1]), _np_area_tab(src_shape[0], dst_shape[0])
    kx = max(len(t) for t in xt)
    xi = np.array([[t[min(k, len(t) - 1)][0] for k in range(kx)] for t in xt])                       # [dcols, kx]
    xa = np.array([[t[k][1] if k < len(t) else f32(0) for k in range(kx)] for t in xt], f32)
    want = np.zeros(dst_shape, np.uint8)
    for dy, taps in enumerate(yt):
        total = np.zeros(dst_shape[1], f32)
        for sy, beta in taps:
            row = src[sy].astype(f32)
            buf = np.zeros(dst_shape[1], f32)
            for k in range(kx):
                buf = (buf + row[xi[:, k]] * xa[:, k]).astype(f32)                                   # a tap beyond the list adds v * 0
            total = (total + beta * buf).astype(f32)
        want[dy] = np.clip(np.rint(total), 0, 255).astype(np.uint8)
    assert np.array_equal(oracle.luma_area_resize(src, *dst_shape), want)


def _np_enlarge_axis(ssize, dsize, horizontal):
    """cv::hal::resize with area_mode and ksize = 2 (the INTER_AREA "enlargement" of imgproc/resize.cpp), one axis: (s0, s1, w0, w1) per index."""
    f32 = np.float32
    inv = dsize / ssize; scale = 1.0 / inv
    s0 = np.floor(np.arange(dsize) * scale).astype(np.int64)
    f = ((np.arange(dsize) + 1) - (s0 + 1) * inv).astype(f32)
    f = np.where(f <= 0, f32(0), f - np.floor(f)).astype(f32)
    w0 = np.clip(np.rint((f32(1) - f) * f32(2048)), -32768, 32767).astype(np.int64)
    w1 = np.clip(np.rint(f * f32(2048)), -32768, 32767).astype(np.int64)
    if horizontal:
        tail = np.maximum.accumulate(s0 + 1 >= ssize)                 # from the first index whose second tap leaves the source: S[last] * 2048
        w0 = np.where(tail, 2048, w0); w1 = np.where(tail, 0, w1)
    return np.clip(s0, 0, ssize - 1), np.clip(s0 + 1, 0, ssize - 1), w0, w1


@pytest.mark.parametrize("src_shape,dst_shape", [((180, 320), (270, 480)), ((135, 240), (270, 480)), ((200, 300), (270, 480)), ((400, 300), (270, 480)),
                                                 ((100, 640), (270, 480)), ((269, 479), (270, 480)), ((7, 5), (33, 47))])
def test_area_resize_enlargement_matches_independent_numpy_restatement(oracle, src_shape, dst_shape):
    """A frame smaller than the detection resolution on either axis (FrameTracker.cpp:117 resizes whatever it gets): cv::resize's bilinear
    emulation of INTER_AREA -- 2 taps per axis, 11-bit fixed point -- restated with whole-image integer arithmetic.  2x enlargement replicates."""
    src = np.random.default_rng(src_shape[0] * 7 + src_shape[1]).integers(0, 256, src_shape, dtype=np.uint8).astype(np.int64)
    x0, x1, a0, a1 = _np_enlarge_axis(src_shape[1], dst_shape[1], True)
    y0, y1, b0, b1 = _np_enlarge_axis(src_shape[0], dst_shape[0], False)
    h = src[:, x0] * a0[None, :] + src[:, x1] * a1[None, :]                          # HResizeLinear, every source row
    want = ((((b0[:, None] * (h[y0] >> 4)) >> 16) + ((b1[:, None] * (h[y1] >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    assert np.array_equal(oracle.luma_area_resize(src.astype(np.uint8), *dst_shape), want)
    if src_shape == (135, 240):
        assert np.array_equal(want, src.astype(np.uint8).repeat(2, 0).repeat(2, 1))


def _np_linear8_axis(ssize, dsize, vertical):
    """cv::resize 8U INTER_LINEAR, one axis: taps and 11-bit weights (imgproc/resize.cpp; the rows are clipped individually, the columns fold the
    tail into a single tap of weight 2048)."""
    f32 = np.float32
    scale = 1.0 / (dsize / ssize)
    f = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(f32)
    s = np.floor(f).astype(np.int64); f = (f - s.astype(f32)).astype(f32)
    if vertical:
        return np.clip(s, 0, ssize - 1), np.clip(s + 1, 0, ssize - 1), np.rint((f32(1) - f) * f32(2048)).astype(np.int64), np.rint(f * f32(2048)).astype(np.int64)
    lo = s < 0; f[lo] = 0; s[lo] = 0
    single = s + 1 >= ssize
    s = np.minimum(s, ssize - 1)
    return (s, np.where(single, s, s + 1), np.where(single, 2048, np.rint((f32(1) - f) * f32(2048)).astype(np.int64)),
            np.where(single, 0, np.rint(f * f32(2048)).astype(np.int64)))


@pytest.mark.parametrize("shape", [(36, 48), (270, 480), (38, 50), (2, 2)])
@pytest.mark.parametrize("nv12", [False, True])
def test_yuv420_conversions_match_independent_numpy_restatement(oracle, shape, nv12):
    """FrameIngest's conversions either side of the filter (Interop/FrameIngest.cpp:494-602): chroma planes enlarged 2x with cv::resize(8U,
    INTER_LINEAR) -- fixed point, phases .25 / .75 --, merged with luma; back: split, 2 x 2 area mean (a + b + c + d + 2) >> 2.  Whole-plane
    integer arithmetic, bit-identical to oracle/ingest.cpp, I420 and NV12."""
    rows, cols = shape
    rng = np.random.default_rng(rows + cols)
    y = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    u = rng.integers(0, 256, (rows // 2, cols // 2), dtype=np.uint8); v = rng.integers(0, 256, (rows // 2, cols // 2), dtype=np.uint8)
    x0, x1, a0, a1 = _np_linear8_axis(cols // 2, cols, False)
    y0, y1, b0, b1 = _np_linear8_axis(rows // 2, rows, True)

    def enlarge(p):
        p = p.astype(np.int64)
        h = p[:, x0] * a0[None, :] + p[:, x1] * a1[None, :]
        return ((((b0[:, None] * (h[y0] >> 4)) >> 16) + ((b1[:, None] * (h[y1] >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    want = np.stack([y, enlarge(u), enlarge(v)], -1)
    got = oracle.ingest_yuv420(y, np.stack([u, v], -1)) if nv12 else oracle.ingest_yuv420(y, u, v)
    assert np.array_equal(got, want)
    # and back
    frame = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    f = frame.astype(np.int64)
    mean = lambda c: ((f[0::2, 0::2, c] + f[0::2, 1::2, c] + f[1::2, 0::2, c] + f[1::2, 1::2, c] + 2) >> 2).astype(np.uint8)
    planes = oracle.egress_yuv420(frame, nv12=nv12)
    assert np.array_equal(planes[0], frame[..., 0])
    if nv12:
        assert np.array_equal(planes[1][..., 0], mean(1)) and np.array_equal(planes[1][..., 1], mean(2))
    else:
        assert np.array_equal(planes[1], mean(1)) and np.array_equal(planes[2], mean(2))
