"""Deterministic synthetic frames for the tests and the bench (no external media)."""
import numpy as np


def textured_frame(rows, cols, seed=0x4C564B31, channels=3):
    """Gratings + rectangles + noise: has FAST corners, edges for EASU and smooth chroma."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    img = np.zeros((rows, cols), np.float32)
    for _ in range(12):
        th = rng.uniform(0, np.pi)
        f = rng.uniform(0.01, 0.15)
        ph = rng.uniform(0, 2 * np.pi)
        img += rng.uniform(4, 14) * np.sin((np.cos(th) * xx + np.sin(th) * yy) * f * 2 * np.pi + ph)
    img += 128
    nrect = max(20, rows * cols // 2500)
    for _ in range(nrect):
        h = int(rng.integers(4, max(5, rows // 6)))
        w = int(rng.integers(4, max(5, cols // 6)))
        y = int(rng.integers(0, rows - h))
        x = int(rng.integers(0, cols - w))
        img[y:y + h, x:x + w] = rng.uniform(10, 245)
    img += rng.normal(0, 1.5, img.shape)
    y8 = np.clip(img, 0, 255).astype(np.uint8)
    if channels == 1:
        return y8
    out = np.empty((rows, cols, 3), np.uint8)
    out[..., 0] = y8
    out[..., 1] = np.clip(128 + 60 * np.sin(xx / cols * 5.0 + 0.3) + 20 * np.cos(yy / rows * 7.0), 0, 255).astype(np.uint8)
    out[..., 2] = np.clip(128 + 50 * np.cos(xx / cols * 3.0 - yy / rows * 4.0), 0, 255).astype(np.uint8)
    return out


def random_homography(rows, cols, rng, strength=1.0):
    """dst->src homography: small rotation/scale/shift/perspective about the frame centre."""
    th = rng.uniform(-0.03, 0.03) * strength
    s = 1.0 + rng.uniform(-0.04, 0.04) * strength
    tx, ty = rng.uniform(-0.03, 0.03, 2) * strength * np.array([cols, rows])
    cx, cy = cols / 2.0, rows / 2.0
    c, si = np.cos(th) * s, np.sin(th) * s
    A = np.array([[c, -si, cx - c * cx + si * cy + tx], [si, c, cy - si * cx - c * cy + ty], [0, 0, 1]], np.float64)
    A[2, 0] = rng.uniform(-2e-5, 2e-5) * strength
    A[2, 1] = rng.uniform(-2e-5, 2e-5) * strength
    return A.astype(np.float32)


def random_mesh(mrows, mcols, rng, amp=0.02):
    return rng.uniform(-amp, amp, (mrows, mcols, 2)).astype(np.float32)


def camera_path(nframes, cols, rng, jitter=1.0, pan=0.0015):
    """Smooth pan + AR(1) jitter: per-frame (tx, ty, theta, zoom) of the camera in pixels / radians."""
    t = np.zeros((nframes, 4))
    ar = np.zeros(4)
    sig = np.array([0.004 * cols, 0.004 * cols, np.deg2rad(0.15), 0.002]) * jitter
    for i in range(nframes):
        ar = 0.6 * ar + np.sqrt(1 - 0.36) * rng.normal(0, 1, 4) * sig
        t[i] = ar
        t[i, 0] += pan * cols * i
        t[i, 1] += 0.3 * pan * cols * i
    return t


def make_clip(rows, cols, nframes, seed=0x4C564B31, jitter=1.0):
    """Synthetic shaky clip: packed YUV444 frames sampled (bilinear) from one textured canvas.
    Returns (frames [n, rows, cols, 3] uint8, path [n, 4]); path[i] = (tx, ty, theta, zoom-1) of frame i in the canvas."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    m = int(0.10 * cols) + 8
    canvas = textured_frame(rows + 2 * m, cols + 2 * m, seed=seed).astype(np.float32)
    path = camera_path(nframes, cols, rng, jitter=jitter)
    frames = np.empty((nframes, rows, cols, 3), np.uint8)
    cy, cx = (rows - 1) / 2.0, (cols - 1) / 2.0
    for i in range(nframes):
        tx, ty, th, z = path[i]
        s = 1.0 + z
        c, si = np.cos(th) * s, np.sin(th) * s
        # output (y, x) -> canvas (y, x): rotate/zoom about the frame centre, then translate
        A = np.array([[c, si], [-si, c]])
        off = np.array([cy + m + ty, cx + m + tx]) - A @ np.array([cy, cx])
        for ch in range(3):
            frames[i, ..., ch] = np.clip(np.rint(ndimage.affine_transform(canvas[..., ch], A, offset=off, output_shape=(rows, cols), order=1, mode="nearest")), 0, 255)
    return frames, path


def lens_distort(frames, corrected_xy):
    """Render what a distorting camera would have recorded: raw(s, t) = ideal(F^-1(s, t)) (bilinear), where
    corrected_xy [rows, cols, 2] holds F^-1 of every raw pixel (e.g. Oracle.lens_undistort_points of the pixel grid)."""
    from scipy import ndimage
    frames = np.asarray(frames)
    single = frames.ndim == 3
    fr = frames[None] if single else frames
    out = np.empty_like(fr)
    coords = np.stack([corrected_xy[..., 1], corrected_xy[..., 0]]).astype(np.float64)
    for i in range(len(fr)):
        for ch in range(3):
            out[i, ..., ch] = np.clip(np.rint(ndimage.map_coordinates(fr[i, ..., ch].astype(np.float32), coords, order=1, mode="nearest")), 0, 255)
    return out[0] if single else out


def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)
