"""The oracle's stateful host arithmetic (quality assurance, WarpMesh arithmetic, PathSmoother: SURVEY.md section 8 rows a2 / a11 / a12) held to
an independent numpy restatement written from the reference's sources (tests/np_smoother.py), value for value, over clips with a scene cut
(trust drops to zero and recovers), a stretch of heavy jitter (the drift clamp and the adaptive smoothing factor move) and both OBS presets.
The GPU tests hold the product's host logic (csrc/host_logic.hpp) to the oracle's per-frame meshes and statistics."""
import numpy as np
import pytest

from tests import np_smoother, oracle_lib, synth


def _clip(seed, n, cut, jitter):
    a, _ = synth.make_clip(360, 640, cut, seed=seed, jitter=jitter)
    b, _ = synth.make_clip(360, 640, n - cut, seed=seed + 101, jitter=jitter)       # another canvas: a scene cut
    return np.concatenate([a, b])


@pytest.mark.parametrize("preset,crop,jitter,samples", [("homography", 1, 1.0, 4), ("homography", 0, 6.0, 3), ("field", 1, 2.0, 3)])
def test_oracle_matches_the_numpy_restatement(oracle, preset, crop, jitter, samples):
    s = oracle_lib.preset(preset, predictive_samples=samples, crop_to_stable_region=crop, min_scene_quality=0.5, min_tracking_quality=0.3,
                          smoothing_steps=6.0, response_rate=0.3)
    frames = _clip(11 + samples, 60, 42, jitter)
    # constructed with the library defaults, then configured -- the OBS plugin's order (VSFilter.hpp:54, VSFilter.cpp:235-294).  (Constructed
    # straight from a 16 x 16 setting, the tracker keeps the similarity constraints of its default 256 x 256 region -- FrameTracker.cpp:73-82
    # regenerates them only when the motion resolution CHANGES -- and the field preset's inlier ratio stays near zero at 480 x 270.)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    qa = np_smoother.QualityAssurance(s.min_tracking_quality, s.min_scene_quality)
    sm = np_smoother.PathSmoother(s.motion_height, s.motion_width, s.predictive_samples, s.corrective_limit_x, s.corrective_limit_y,
                                  s.smoothing_steps, s.response_rate)
    emitted = trusted = distrusted = clamped = from_h = 0
    factors = []
    for i, f in enumerate(frames):
        out, _ = ost.push(f, ts=i)
        st = ost.stats()
        motion, correction = ost.meshes()
        # quality assurance: the trust factor and the scene quality follow from the tracker's stability alone
        trust = qa.update(st.tracking_stability)
        assert np.float32(st.trust) == trust and np.float32(st.scene_quality) == qa.scene_quality, i
        trusted += trust == 1.0; distrusted += trust == 0.0
        # homography preset: the (trust-scaled) motion mesh is WarpMesh::set_to(H, tracking size) of the frame's estimate
        if preset == "homography" and st.tracking_stability > 0:
            want = (np_smoother.mesh_from_homography(np.array(st.homography[:]), s.motion_height, s.motion_width, s.detection_width, s.detection_height)
                    * trust).astype(np.float32)
            assert np.array_equal(motion, want), i
            from_h += 1
        corr = sm.next(motion)
        assert float(st.smoothing_factor) == sm.smoothing_factor, i
        factors.append(sm.smoothing_factor)
        if out is not None:
            emitted += 1
            mx, my = sm.margin[0], sm.margin[1]
            clamped += bool((np.abs(corr[..., 0]) == mx).any() or (np.abs(corr[..., 1]) == my).any())
            want = (corr + sm.scene_crop).astype(np.float32) if crop else corr
            assert np.array_equal(correction, want), i
    ost.close()
    assert emitted == len(frames) - samples
    assert distrusted >= 2 and trusted >= 5                    # the scene cut resets the trust, the clip earns it back
    assert len(set(factors)) > 10                              # the adaptive smoothing factor moved
    if preset == "homography":
        assert from_h >= 20
    if jitter >= 6.0:
        assert clamped >= 1                                    # the drift clamp engaged


def test_restart_clears_the_path(oracle):
    s = oracle_lib.preset("homography", predictive_samples=2, crop_to_stable_region=0)
    frames, _ = synth.make_clip(360, 640, 12, seed=5, jitter=2.0)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    sm = np_smoother.PathSmoother(2, 2, 2, s.corrective_limit_x, s.corrective_limit_y, s.smoothing_steps, s.response_rate)
    for i, f in enumerate(frames):
        if i == 7:
            ost.restart(); sm.restart()
        out, _ = ost.push(f, ts=i)
        motion, correction = ost.meshes()
        corr = sm.next(motion)
        assert float(ost.stats().smoothing_factor) == sm.smoothing_factor, i
        if out is not None:
            assert np.array_equal(correction, corr), i
    ost.close()


def test_gaussian_kernel_convention_against_scipy():
    """cv::getGaussianKernel(n, sigma): samples exp(-x^2 / (2 sigma^2)) centred on (n - 1) / 2, normalised to 1 -- scipy.signal.windows.gaussian is
    the same window up to the normalisation (a third party's centre / sigma convention)."""
    from scipy.signal.windows import gaussian
    for n, sigma in ((21, 1.75), (7, 0.6), (9, 3.25), (2, 1.0)):
        w = gaussian(n, std=sigma); w = w / w.sum()
        assert np.abs(np_smoother.gaussian_kernel_f32(n, sigma).astype(np.float64) - w).max() < 1e-7
