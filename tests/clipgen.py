"""SURVEY.md section 8d's synthetic clip generator (deterministic, no external media), rendered with torch on whatever device it is given
(the GPU in bench.py, the CPU in the tests).  TEST / BENCH INFRASTRUCTURE, not product code.

Scene canvas (W + 2 m) x (H + 2 m), m = 8 % W: random-orientation sinusoid gratings + random axis-aligned rectangles of random luma
(FAST corners) + low-amplitude noise; chroma = two smooth fields.  Camera path per frame: a smooth pan (closed loop over the clip, peak
speed <= 0.15 % W per frame) plus AR(1) jitter (rho = 0.6) with sigma_t = 0.4 % W translation, sigma_theta = 0.15 deg, sigma_s = 0.2 %
zoom; an optional scene cut (a second canvas from frame `cut_at` on).  Frames are bilinear samples of the canvas.  Ground truth: the
per-frame frame -> canvas homography (hence every inter-frame homography) and the jitter-free "ideal" render."""
import numpy as np


class Clip:
    def __init__(self, rows, cols, n, seed=0x4C564B31, device="cpu", cut_at=None, jitter=1.0):
        import torch
        self.torch = torch
        self.rows, self.cols, self.n, self.device, self.cut_at = rows, cols, n, device, cut_at
        self.m = m = int(round(0.08 * cols))
        rng = np.random.default_rng(seed)
        self.canvases = [self._canvas(rng)]
        if cut_at is not None:
            self.canvases.append(self._canvas(rng))
        # camera path: (tx, ty, theta, zoom - 1) of the frame centre in the canvas
        t = np.arange(n)
        smooth = np.zeros((n, 4))
        smooth[:, 0] = 0.040 * cols * np.sin(2 * np.pi * t / n)                 # peak speed 0.04 W 2 pi / n  (0.042 % W per frame at n = 600)
        smooth[:, 1] = 0.018 * cols * np.sin(4 * np.pi * t / n + 0.7)
        sig = np.array([0.004 * cols, 0.004 * cols, np.deg2rad(0.15), 0.002]) * jitter
        ar = np.zeros(4); jit = np.zeros((n, 4))
        for i in range(n):
            ar = 0.6 * ar + np.sqrt(1 - 0.36) * rng.normal(0, 1, 4) * sig
            jit[i] = ar
        self.smooth, self.shaky = smooth, smooth + jit

    # ---- scene ---------------------------------------------------------------------------------------------------------
    def _canvas(self, rng):
        torch = self.torch
        H, W = self.rows + 2 * self.m, self.cols + 2 * self.m
        dev = self.device
        yy = torch.arange(H, device=dev, dtype=torch.float32)[:, None]
        xx = torch.arange(W, device=dev, dtype=torch.float32)[None, :]
        img = torch.full((H, W), 128.0, device=dev)
        for _ in range(24):
            th, f, ph, a = rng.uniform(0, np.pi), rng.uniform(0.004, 0.06), rng.uniform(0, 6.28), rng.uniform(2, 9)
            img += a * torch.sin((np.cos(th) * xx + np.sin(th) * yy) * (f * 6.2832) + ph)
        nrect = max(400, (H * W) // 3300)
        ys = rng.integers(0, H - 8, nrect); xs = rng.integers(0, W - 8, nrect)
        hs = rng.integers(8, max(9, H // 10), nrect); ws = rng.integers(8, max(9, W // 10), nrect)
        vs = rng.uniform(10, 245, nrect)
        for i in range(nrect):
            img[ys[i]:ys[i] + hs[i], xs[i]:xs[i] + ws[i]] = float(vs[i])
        g = torch.Generator(device="cpu"); g.manual_seed(int(rng.integers(0, 2 ** 31)))
        noise = torch.randn((H // 4 + 1, W // 4 + 1), generator=g).to(dev)
        img += 1.5 * torch.nn.functional.interpolate(noise[None, None], size=(H, W), mode="bilinear", align_corners=False)[0, 0]
        c = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        c[0] = img.clamp(0, 255)
        a1, a2, a3 = rng.uniform(3, 7, 3)
        c[1] = (128 + 60 * torch.sin(xx / W * a1 + 0.3) + 20 * torch.cos(yy / H * a2)).clamp(0, 255)
        c[2] = (128 + 50 * torch.cos(xx / W * a3 - yy / H * 4.0)).clamp(0, 255)
        return c

    # ---- geometry ------------------------------------------------------------------------------------------------------
    def matrix(self, i, smooth=False):
        """3 x 3 float64: frame pixel (x, y, 1) -> canvas pixel of frame i (rotation / zoom about the frame centre, then translation)."""
        tx, ty, th, z = (self.smooth if smooth else self.shaky)[i % self.n]
        s = 1.0 + z
        c, si = np.cos(th) * s, np.sin(th) * s
        cx, cy = (self.cols - 1) / 2.0, (self.rows - 1) / 2.0
        A = np.array([[c, -si, 0.0], [si, c, 0.0], [0, 0, 1.0]])
        T0 = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
        T1 = np.array([[1, 0, cx + self.m + tx], [0, 1, cy + self.m + ty], [0, 0, 1.0]])
        return T1 @ A @ T0

    def motion(self, i):
        """Ground-truth inter-frame homography: pixel of frame i - 1 -> pixel of frame i (same canvas only)."""
        return np.linalg.inv(self.matrix(i)) @ self.matrix(i - 1)

    def canvas_of(self, i):
        i = i % self.n
        return self.canvases[1 if (self.cut_at is not None and i >= self.cut_at) else 0]

    # ---- rendering -----------------------------------------------------------------------------------------------------
    def render444(self, i, smooth=False):
        """[rows, cols, 3] uint8 packed YUV 4:4:4 of frame i (smooth=True: the jitter-free ideal render)."""
        torch = self.torch
        M = self.matrix(i, smooth)
        canvas = self.canvas_of(i)
        Hc, Wc = canvas.shape[1:]
        ys = torch.arange(self.rows, device=self.device, dtype=torch.float32)[:, None]
        xs = torch.arange(self.cols, device=self.device, dtype=torch.float32)[None, :]
        gx = float(M[0, 0]) * xs + float(M[0, 1]) * ys + float(M[0, 2])
        gy = float(M[1, 0]) * xs + float(M[1, 1]) * ys + float(M[1, 2])
        grid = torch.stack([gx * (2.0 / (Wc - 1)) - 1.0, gy * (2.0 / (Hc - 1)) - 1.0], dim=-1)[None]
        out = torch.nn.functional.grid_sample(canvas[None], grid, mode="bilinear", padding_mode="border", align_corners=True)[0]
        return (out + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous()

    @staticmethod
    def to_i420(packed, nv12=False):
        """Packed 4:4:4 -> (y, u, v) / (y, uv) planes, chroma = (a + b + c + d + 2) >> 2 (the plugin's INTER_AREA subsampling)."""
        import torch
        y = packed[..., 0].contiguous()
        c = packed[..., 1:].to(torch.int32)
        s = (c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2] + 2) >> 2
        s = s.to(torch.uint8)
        if nv12:
            return y, s.contiguous()
        return y, s[..., 0].contiguous(), s[..., 1].contiguous()

    def render_i420(self, i, smooth=False, nv12=False):
        return self.to_i420(self.render444(i, smooth), nv12)


def psnr_region(a, b, margin=0.08):
    """PSNR over the central region of two [rows, cols, 3] (or [rows, cols]) uint8 arrays / tensors, luma channel only for 3-channel input."""
    import torch
    if not isinstance(a, torch.Tensor):
        a = torch.from_numpy(np.ascontiguousarray(a))
    if not isinstance(b, torch.Tensor):
        b = torch.from_numpy(np.ascontiguousarray(b))
    if a.dim() == 3:
        a = a[..., 0]; b = b[..., 0]
    rows, cols = a.shape
    my, mx = int(margin * rows), int(margin * cols)
    d = a[my:rows - my, mx:cols - mx].to(torch.float32) - b[my:rows - my, mx:cols - mx].to(b.device).to(torch.float32).to(a.device)
    mse = float((d * d).mean())
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)
