"""Second, independent restatement of FSR.cl's EASU remap in vectorised numpy (float32).

Written straight from LiveVisionKit/Functions/OpenCL/Sources/FSR.cl:98-318,407-452 to pin the C oracle
(oracle/easu.cpp) structurally: tap order, offsets, luma selection, border rules.  Fused multiply-adds are
emulated through float64 (exact product, one extra rounding), so agreement with the C oracle is checked
as "<= 1 LSB, almost everywhere equal", not bit-for-bit.
"""
import numpy as np

f32 = np.float32


def _fma(a, b, c):
    return (a.astype(np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def _rcp_lo(a):
    return (np.uint32(0x7ef07ebb) - a.view(np.uint32)).view(f32)


def _rsq_lo(a):
    return (np.uint32(0x5f347d74) - (a.view(np.uint32) >> np.uint32(1))).view(f32)


def _sat(x):
    return np.maximum(f32(0), np.minimum(f32(1), x))


def easu_points(src, sx, sy, ppx, ppy, yuv):
    """src [rows, cols, 3] uint8; sx, sy int arrays (valid EASU interior); ppx, ppy float32 arrays."""
    norm = f32(0.00392156862)

    def px(dx, dy):
        return src[sy + dy, sx + dx].astype(f32) * norm       # [..., 3]

    b, c = px(0, -1), px(1, -1)
    e, f, g, h = px(-1, 0), px(0, 0), px(1, 0), px(2, 0)
    i, j, k, l = px(-1, 1), px(0, 1), px(1, 1), px(2, 1)
    n, o = px(0, 2), px(1, 2)

    def luma(p):
        if yuv:
            return _fma(p[..., 2], f32(0.5), _fma(p[..., 0], f32(0.5), p[..., 1]))
        return p[..., 0]

    L = {name: luma(v) for name, v in dict(b=b, c=c, e=e, f=f, g=g, h=h, i=i, j=j, k=k, l=l, n=n, o=o).items()}
    one = f32(1)
    dirx = np.zeros_like(ppx); diry = np.zeros_like(ppx); ln = np.zeros_like(ppx)

    def acc(w, lA, lB, lC, lD, lE):
        nonlocal dirx, diry, ln
        dc = lD - lC; cb = lC - lB
        lenX = _rcp_lo(np.maximum(np.abs(dc), np.abs(cb)))
        dX = lD - lB
        dirx = _fma(dX, w, dirx)
        lenX = _sat(np.abs(dX) * lenX); lenX = lenX * lenX
        ln = _fma(lenX, w, ln)
        ec = lE - lC; ca = lC - lA
        lenY = _rcp_lo(np.maximum(np.abs(ec), np.abs(ca)))
        dY = lE - lA
        diry = _fma(dY, w, diry)
        lenY = _sat(np.abs(dY) * lenY); lenY = lenY * lenY
        ln = _fma(lenY, w, ln)

    omx, omy = one - ppx, one - ppy
    acc(omx * omy, L['b'], L['e'], L['f'], L['g'], L['j'])
    acc(ppx * omy, L['c'], L['f'], L['g'], L['h'], L['k'])
    acc(omx * ppy, L['f'], L['i'], L['j'], L['k'], L['n'])
    acc(ppx * ppy, L['g'], L['j'], L['k'], L['l'], L['o'])

    dirR = dirx * dirx + diry * diry          # two statements in FSR.cl:252-253: not contracted
    zro = dirR < f32(1.0 / 32768.0)
    dirR = _rsq_lo(dirR)
    dirR = np.where(zro, one, dirR)
    dirx = np.where(zro, one, dirx)
    dirx = dirx * dirR; diry = diry * dirR
    ln = ln * f32(0.5); ln = ln * ln
    stretch = _fma(dirx, dirx, diry * diry) * _rcp_lo(np.maximum(np.abs(dirx), np.abs(diry)))
    len2x = _fma(stretch - one, ln, one)
    len2y = _fma(f32(-0.5), ln, one)
    lob = _fma((f32(1.0) / f32(4.0) - f32(0.04)) - f32(0.5), ln, f32(0.5))
    clp = _rcp_lo(lob)

    mi4 = np.minimum(np.minimum(f, g), np.minimum(j, k))
    ma4 = np.maximum(np.maximum(f, g), np.maximum(j, k))

    aC = np.zeros(ppx.shape + (3,), f32); aW = np.zeros_like(ppx)

    def tap(ox, oy, col):
        nonlocal aC, aW
        offx = f32(ox) - ppx; offy = f32(oy) - ppy
        vx = _fma(offx, dirx, offy * diry)
        vy = _fma(offx, -diry, offy * dirx)
        vx = vx * len2x; vy = vy * len2y
        d2 = np.minimum(_fma(vx, vx, vy * vy), clp)
        wA = _fma(lob, d2, f32(-1)); wB = _fma(f32(2.0) / f32(5.0), d2, f32(-1))
        wA = wA * wA
        wB = _fma(f32(25.0) / f32(16.0), wB * wB, -(f32(25.0) / f32(16.0) - f32(1)))
        w = wB * wA
        aC = _fma(col, w[..., None], aC)
        aW = aW + w

    tap(0, -1, b); tap(1, -1, c); tap(-1, 1, i); tap(0, 1, j); tap(0, 0, f); tap(-1, 0, e)
    tap(1, 1, k); tap(2, 1, l); tap(2, 0, h); tap(1, 0, g); tap(0, 2, n); tap(1, 2, o)

    rW = one / aW
    fpx = np.minimum(ma4, np.maximum(mi4, aC * rW[..., None]))
    return (fpx * f32(255.0)).astype(np.int32).astype(np.uint8)


def remap_homography(src, H, bg, yuv):
    rows, cols = src.shape[:2]
    H = np.asarray(H, f32).reshape(9)
    yy, xx = np.mgrid[0:rows, 0:cols]
    fx = xx.astype(f32); fy = yy.astype(f32)
    # FSR.cl:423-427 as clang contracts it: ((r.x * fx) + (r.y * fy)) + r.z = fma(r.x, fx, r.y * fy) + r.z
    dz = f32(1) / (_fma(H[6], fx, H[7] * fy) + H[8])
    ox = (_fma(H[0], fx, H[1] * fy) + H[2]) * dz - fx
    oy = (_fma(H[3], fx, H[4] * fy) + H[5]) * dz - fy
    subx = fx + ox; suby = fy + oy
    return _remap_tail(src, subx, suby, bg, yuv)


def remap_map(src, offmap, bg, yuv):
    rows, cols = src.shape[:2]
    yy, xx = np.mgrid[0:rows, 0:cols]
    subx = xx.astype(f32) + offmap[..., 0]; suby = yy.astype(f32) + offmap[..., 1]
    return _remap_tail(src, subx, suby, bg, yuv)


def _remap_tail(src, subx, suby, bg, yuv):
    rows, cols = src.shape[:2]
    sx = np.trunc(np.clip(subx, -2e9, 2e9)).astype(np.int64)
    sy = np.trunc(np.clip(suby, -2e9, 2e9)).astype(np.int64)
    ppx = subx - np.floor(subx); ppy = suby - np.floor(suby)
    out = np.empty(src.shape, np.uint8)
    out[...] = np.asarray(bg, np.uint8)
    border = (sx < 1) | (sy < 1) | (sx >= cols - 4) | (sy >= rows - 4)
    inside = (sx >= 0) & (sx < cols) & (sy >= 0) & (sy < rows)
    nn = border & inside
    out[nn] = src[sy[nn], sx[nn]]
    ea = ~border
    if ea.any():
        out[ea] = easu_points(src, sx[ea], sy[ea], ppx[ea].astype(f32), ppy[ea].astype(f32), yuv)
    return out


def upscale(src, size, yuv):
    """lvk::upscale / easu_scale (Functions/Image.cpp:155-202, FSR.cl:324-358); size = (width, height)."""
    rows, cols = src.shape[:2]
    dw, dh = int(size[0]), int(size[1])
    if (dw, dh) == (cols, rows):
        return src.copy()
    rsx = f32(cols) / f32(dw); rsy = f32(rows) / f32(dh)
    yy, xx = np.mgrid[0:dh, 0:dw]
    subx = xx.astype(f32) * rsx; suby = yy.astype(f32) * rsy
    sx = np.trunc(subx).astype(np.int64); sy = np.trunc(suby).astype(np.int64)
    ppx = subx - np.floor(subx); ppy = suby - np.floor(suby)
    out = np.empty((dh, dw, 3), np.uint8)
    border = (sx == 0) | (sy == 0) | (sx >= cols - 4) | (sy >= rows - 4)
    out[border] = src[sy[border], sx[border]]
    ea = ~border
    if ea.any():
        out[ea] = easu_points(src, sx[ea], sy[ea], ppx[ea].astype(f32), ppy[ea].astype(f32), yuv)
    return out


def sharpen(src, sharpness):
    """lvk::sharpen / rcas (Functions/Image.cpp:206-233, FSR.cl:460-535), out of place."""
    rows, cols = src.shape[:2]
    sharp = f32(np.exp2(f32(-2.0) * (f32(1.0) - f32(sharpness))))
    out = src.copy()
    if rows < 3 or cols < 3:
        return out
    norm = f32(0.00392156862)
    p = src.astype(f32) * norm
    b = p[:-2, 1:-1]; h = p[2:, 1:-1]; d = p[1:-1, :-2]; e = p[1:-1, 1:-1]; f = p[1:-1, 2:]
    with np.errstate(divide='ignore', invalid='ignore'):
        mn4 = np.fmin(b, np.fmin(d, np.fmin(f, h))); mx4 = np.fmax(b, np.fmax(d, np.fmax(f, h)))
        hit_min = np.fmin(mn4, e) * (f32(1) / (f32(4) * mx4))
        hit_max = (f32(1) - np.fmax(mx4, e)) * (f32(1) / _fma(mn4, f32(4), f32(-4)))
        lobe_c = np.fmax(-hit_min, hit_max)
    lobe = np.fmax(lobe_c[..., 2], np.fmax(lobe_c[..., 1], lobe_c[..., 0]))
    lobe = np.fmin(np.fmax(lobe, f32(-0.1875)), f32(0)) * sharp
    a = _fma(lobe, f32(4), f32(1))
    r = (np.uint32(0x7ef19fff) - a.view(np.uint32)).view(f32)
    rcp = r * _fma(-r, a, f32(2))
    v = _fma(((b + d) + h) + f, lobe[..., None], e) * rcp[..., None]
    out[1:-1, 1:-1] = (v * f32(255)).astype(np.int32).astype(np.uint8)
    return out
