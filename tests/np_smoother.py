"""Independent numpy restatement of the stateful host arithmetic behind lvk::StabilizationFilter -- the quality-assurance trust factor
(Filters/StabilizationFilter.cpp:101-115), WarpMesh::set_to(Homography) / scaling / crop_in / clamp / combine (Math/WarpMesh.cpp:333-342,379-390,
411-417,445-448,493-507,548-551) and PathSmoother::configure / next (Vision/PathSmoother.cpp:36-135) -- written from the reference's sources with
whole-array float32 arithmetic, NOT from oracle/stabilizer.cpp or csrc/host_logic.hpp (those two are near-twins by the same hand: round-2
VERDICT, weak #1 (ii)).  tests/test_np_smoother.py holds the oracle to it, value for value; the GPU tests hold the product to the oracle.

Test infrastructure only.  Unpinnable OpenCV internals are restated as the oracle documents them: cv::getGaussianKernel in
binary64 with libm's exp, cv::scaleAdd as multiply then add, cv::perspectiveTransform of Point2f with a CV_64F matrix in binary64."""
import math

import numpy as np

F = np.float32
QA_UPDATE_RATE = F(0.1)        # StabilizationFilter.cpp:29
QA_BLEND_STEP = F(0.05)        # :30


def mesh_from_homography(H, rows, cols, scale_w, scale_h):
    """WarpMesh::set_to(const Homography&, motion_scale) (WarpMesh.cpp:333-342): offset = (p - H p) / scale at p = coord * scale / (size - 1)."""
    H = np.asarray(H, np.float64).reshape(3, 3)
    csx, csy = F(scale_w) / F(cols - 1), F(scale_h) / F(rows - 1)
    nfx, nfy = F(1.0) / F(scale_w), F(1.0) / F(scale_h)
    sx = (np.arange(cols, dtype=F) * csx)[None, :].repeat(rows, 0)
    sy = (np.arange(rows, dtype=F) * csy)[:, None].repeat(cols, 1)
    x, y = sx.astype(np.float64), sy.astype(np.float64)
    w = x * H[2, 0] + y * H[2, 1] + H[2, 2]
    ok = np.abs(w) > float(np.finfo(F).eps)
    iw = np.where(ok, 1.0 / np.where(ok, w, 1.0), 0.0)
    tx = np.where(ok, (x * H[0, 0] + y * H[0, 1] + H[0, 2]) * iw, 0.0).astype(F)
    ty = np.where(ok, (x * H[1, 0] + y * H[1, 1] + H[1, 2]) * iw, 0.0).astype(F)
    return np.stack([(sx - tx) * nfx, (sy - ty) * nfy], -1).astype(F)


class QualityAssurance:
    """m_SceneQuality / m_TrustFactor (StabilizationFilter.cpp:101-115; exp_moving_average and step: Functions/Math.tpp:133-142,198-204)."""

    def __init__(self, min_tracking_quality, min_scene_quality):
        self.min_tq, self.min_sq = F(min_tracking_quality), F(min_scene_quality)
        self.scene_quality, self.trust = F(0.0), F(0.0)

    @staticmethod
    def _step(current, target, amount):
        return max(F(current - amount), target) if current > target else min(F(current + amount), target)

    def update(self, tracking_quality):
        tq = F(tracking_quality)
        self.scene_quality = F(self.scene_quality + F(QA_UPDATE_RATE * F(tq - self.scene_quality)))
        if tq < self.min_tq:
            self.trust = F(0.0)
        elif self.scene_quality < self.min_sq:
            self.trust = self._step(self.trust, F(0.0), QA_BLEND_STEP)
        else:
            self.trust = self._step(self.trust, F(1.0), QA_BLEND_STEP)
        return self.trust


def gaussian_kernel_f32(n, sigma):
    """cv::getGaussianKernel(n, sigma > 0, CV_32F): exp(-x^2 / (2 sigma^2)) around (n - 1) / 2, normalised in binary64, stored as binary32."""
    s2 = -0.5 / (sigma * sigma)
    k = [math.exp(s2 * (i - (n - 1) * 0.5) * (i - (n - 1) * 0.5)) for i in range(n)]
    inv = 1.0 / _sum_in_order(k)
    return np.array([F(v * inv) for v in k], F)


def _sum_in_order(values):
    total = 0.0
    for v in values:
        total += v
    return total


class PathSmoother:
    """Vision/PathSmoother.cpp.  Meshes are float32 arrays [rows, cols, 2] of normalised offsets."""

    def __init__(self, rows, cols, predictive_samples, limit_x, limit_y, smoothing_steps, response_rate):
        self.shape = (rows, cols, 2)
        self.window = 2 * predictive_samples + 1                                  # :57
        self.trajectory = [np.zeros(self.shape, F) for _ in range(self.window)]    # resize + pad_front with identity meshes :63-64
        self.position = np.zeros(self.shape, F)                                   # :67-71: identity meshes sum to identity
        self.base = float(self.window) / 12.0                                     # :74
        self.smoothing_factor = 0.0
        self.steps, self.rate = float(F(smoothing_steps)), float(F(response_rate))
        # crop<float>({1, 1}, corrective_limits) (Functions/Math.tpp:218-233)
        hx, hy = F(F(1.0) * F(limit_x)), F(F(1.0) * F(limit_y))
        self.margin = (F(hx / F(2)), F(hy / F(2)), F(F(1.0) - hx), F(F(1.0) - hy))        # x, y, width, height
        # m_SceneCrop.crop_in(m_SceneMargins) (WarpMesh.cpp:379-390) on an identity mesh
        sx, sy = F(self.margin[2] - F(1.0)) / F(cols - 1), F(self.margin[3] - F(1.0)) / F(rows - 1)
        gx = (np.arange(cols, dtype=F) * sx + self.margin[0])[None, :].repeat(rows, 0)
        gy = (np.arange(rows, dtype=F) * sy + self.margin[1])[:, None].repeat(cols, 1)
        self.scene_crop = np.stack([gx, gy], -1).astype(F)

    def restart(self):                                                            # :139-145
        for m in self.trajectory:
            m[...] = 0
        self.position[...] = 0

    def next(self, motion):                                                       # :84-135
        motion = np.asarray(motion, F).reshape(self.shape)
        self.position = self.position - self.trajectory[0]
        self.trajectory = self.trajectory[1:] + [motion.copy()]                    # StreamBuffer::push on a full buffer
        centre = (self.window - 1) // 2
        self.position = self.position + self.trajectory[centre]
        filt = gaussian_kernel_f32(self.window, self.base + self.smoothing_factor)
        weight = F(1.0)
        trace = self.trajectory[0].copy()
        for i in range(1, self.window):
            weight = F(weight - filt[i - 1])
            trace = (self.trajectory[i] * weight + trace).astype(F)                # cv::scaleAdd (WarpMesh.cpp:445-448)
        correction = (trace - self.position).astype(F)
        mx, my = self.margin[0], self.margin[1]
        drift = F(0.0)
        if correction.size:
            drift = max(F(0.0), F(np.max(np.abs(correction[..., 0]) / mx)), F(np.max(np.abs(correction[..., 1]) / my)))
        if drift > F(1.0):
            correction[..., 0] = np.clip(correction[..., 0], -mx, mx)
            correction[..., 1] = np.clip(correction[..., 1], -my, my)
            drift = F(1.0)
        d = float(drift)
        target = 0.0 if d >= 0.7 else (self.steps if d <= 0.3 else d)             # hysteresis<double> (Functions/Logic.tpp:53-65)
        self.smoothing_factor = self.smoothing_factor + self.rate * (target - self.smoothing_factor)
        return correction
