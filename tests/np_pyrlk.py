"""Independent numpy restatement of the sparse pyramidal Lucas-Kanade tracker as SURVEY.md Appendix A.3 / A.4 specifies it
(cv::calcOpticalFlowPyrLK, OpenCV 4.8 CPU fixed-point path, as Vision/FrameTracker.cpp:33-35,42-48,140-146 configures it: 11 x 11 window,
maxLevel 3, <= 5 iterations, epsilon 0.01, minEigThreshold 1e-4, flags 0).  TEST INFRASTRUCTURE: a second statement of row a7 next to
oracle/pyrlk.cpp, written from the appendix -- whole-window array arithmetic instead of the oracle's pixel loops, numpy's own pyramid and
Scharr -- so that a slip in either shows up as a difference (tests/test_oracle_imgproc.py compares the two bit for bit).

Arithmetic: Q14 bilinear weights (the fourth by complement), patches DESCALEd to 5 fractional bits, derivative patches to integers,
covariance / mismatch sums exact in integers and converted to binary32 once (the oracle's stated choice where OpenCV's float
accumulation is SIMD-width dependent), every float operation a separately rounded binary32 operation."""
import numpy as np

f32 = np.float32
W_BITS = 14
FLT_SCALE = f32(1.0 / (1 << 20))
FLT_EPSILON = f32(1.1920928955078125e-07)


def pyr_down(img):
    """5 x 5 [1 4 6 4 1] / 256 with (sum + 128) >> 8, BORDER_REFLECT_101, size ((w + 1) / 2, (h + 1) / 2)  (A.3)."""
    k = np.array([1, 4, 6, 4, 1], np.int64)
    p = np.pad(img.astype(np.int64), 2, mode="reflect")
    rows, cols = img.shape
    dr, dc = (rows + 1) // 2, (cols + 1) // 2
    h = sum(k[i] * p[:, i:i + 2 * dc:2][:, :dc] for i in range(5))
    v = sum(k[i] * h[i:i + 2 * dr:2][:dr] for i in range(5))
    return ((v + 128) >> 8).astype(np.uint8)


def scharr(img):
    """(Ix, Iy) int16 planes: vertical [3 10 3] / [-1 0 1] pass, horizontal [-1 0 1] / [3 10 3] pass, reflect-101  (A.3)."""
    p = np.pad(img.astype(np.int64), 1, mode="reflect")
    t0 = 3 * (p[:-2] + p[2:]) + 10 * p[1:-1]
    t1 = p[2:] - p[:-2]
    return (t0[:, 2:] - t0[:, :-2]), (3 * (t1[:, 2:] + t1[:, :-2]) + 10 * t1[:, 1:-1])


def build_pyramid(img, max_level, win):
    levels = [np.ascontiguousarray(img, np.uint8)]
    for _ in range(max_level):
        nxt = pyr_down(levels[-1])
        if nxt.shape[1] <= win[0] or nxt.shape[0] <= win[1]:
            break
        levels.append(nxt)
    return levels


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _weights(a, b):
    one = f32(1.0)
    s = f32(1 << W_BITS)
    w00 = int(np.rint(f32(f32(one - a) * f32(one - b)) * s))
    w01 = int(np.rint(f32(a * f32(one - b)) * s))
    w10 = int(np.rint(f32(f32(one - a) * b) * s))
    return w00, w01, w10, (1 << W_BITS) - w00 - w01 - w10


def _window(padded, pad, iy, ix, win):
    """(win_h + 1) x (win_w + 1) block whose top-left pixel is image pixel (ix, iy); `padded` carries `pad` pixels of border."""
    return padded[iy + pad:iy + pad + win[1] + 1, ix + pad:ix + pad + win[0] + 1]


def _sample(block, w, shift):
    w00, w01, w10, w11 = w
    return _descale(block[:-1, :-1] * w00 + block[:-1, 1:] * w01 + block[1:, :-1] * w10 + block[1:, 1:] * w11, shift)


def calc(prev, nxt, pts, win=(11, 11), max_level=3, max_count=5, epsilon=0.01, min_eig=1e-4):
    pts = np.asarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    P, N = build_pyramid(prev, max_level, win), build_pyramid(nxt, max_level, win)
    out = np.zeros((n, 2), np.float32)
    status = np.ones(n, np.uint8)
    eps2 = min(max(float(epsilon), 0.0), 10.0) ** 2
    half = (f32((win[0] - 1) * 0.5), f32((win[1] - 1) * 0.5))
    pad = max(win) + 2
    for level in range(len(P) - 1, -1, -1):
        I = np.pad(P[level].astype(np.int64), pad, mode="reflect")                       # image border: reflect-101
        dx_, dy_ = scharr(P[level])
        Ix = np.pad(dx_, pad, mode="constant"); Iy = np.pad(dy_, pad, mode="constant")   # derivative border: zeros
        J = np.pad(N[level].astype(np.int64), pad, mode="reflect")
        rows, cols = P[level].shape
        inv = f32(1.0 / (1 << level))
        for i in range(n):
            px, py = f32(pts[i, 0] * inv), f32(pts[i, 1] * inv)
            if level == len(P) - 1:
                nx, ny = px, py
            else:
                nx, ny = f32(out[i, 0] * f32(2.0)), f32(out[i, 1] * f32(2.0))
            out[i] = (nx, ny)
            px, py = f32(px - half[0]), f32(py - half[1])
            ipx, ipy = int(np.floor(px)), int(np.floor(py))
            if ipx < -win[0] or ipx >= cols or ipy < -win[1] or ipy >= rows:
                if level == 0:
                    status[i] = 0
                continue
            w = _weights(f32(px - f32(ipx)), f32(py - f32(ipy)))
            Iw = _sample(_window(I, pad, ipy, ipx, win), w, W_BITS - 5)
            Ixw = _sample(_window(Ix, pad, ipy, ipx, win), w, W_BITS)
            Iyw = _sample(_window(Iy, pad, ipy, ipx, win), w, W_BITS)
            A11 = f32(f32(int((Ixw * Ixw).sum())) * FLT_SCALE); A12 = f32(f32(int((Ixw * Iyw).sum())) * FLT_SCALE); A22 = f32(f32(int((Iyw * Iyw).sum())) * FLT_SCALE)
            D = f32(f32(A11 * A22) - f32(A12 * A12))
            d = f32(A11 - A22)
            min_e = f32(f32(f32(A22 + A11) - np.sqrt(f32(f32(d * d) + f32(f32(f32(4.0) * A12) * A12)))) / f32(2 * win[0] * win[1]))
            if min_e < f32(min_eig) or D < FLT_EPSILON:
                if level == 0:
                    status[i] = 0
                continue
            D = f32(f32(1.0) / D)
            nx, ny = f32(nx - half[0]), f32(ny - half[1])
            pdx = pdy = f32(0.0)
            for j in range(max_count):
                inx, iny = int(np.floor(nx)), int(np.floor(ny))
                if inx < -win[0] or inx >= cols or iny < -win[1] or iny >= rows:
                    if level == 0:
                        status[i] = 0
                    break
                wj = _weights(f32(nx - f32(inx)), f32(ny - f32(iny)))
                diff = _sample(_window(J, pad, iny, inx, win), wj, W_BITS - 5) - Iw
                b1 = f32(f32(int((diff * Ixw).sum())) * FLT_SCALE); b2 = f32(f32(int((diff * Iyw).sum())) * FLT_SCALE)
                dx = f32(f32(f32(A12 * b2) - f32(A22 * b1)) * D)
                dy = f32(f32(f32(A12 * b1) - f32(A11 * b2)) * D)
                nx, ny = f32(nx + dx), f32(ny + dy)
                out[i] = (f32(nx + half[0]), f32(ny + half[1]))
                if float(dx) * float(dx) + float(dy) * float(dy) <= eps2:
                    break
                if j > 0 and abs(float(f32(dx + pdx))) < 0.01 and abs(float(f32(dy + pdy))) < 0.01:
                    out[i] = (f32(out[i, 0] - f32(dx * f32(0.5))), f32(out[i, 1] - f32(dy * f32(0.5))))
                    break
                pdx, pdy = dx, dy
    return out, status
