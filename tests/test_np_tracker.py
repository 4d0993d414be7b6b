"""The oracle's tracker list logic (SURVEY.md section 8 rows a5's suppression grid, a6, a8, a13 and the control flow of a2 / FrameTracker::track)
held to an independent Python restatement written from the reference's sources (tests/np_tracker.py): after every frame the feature list the
detector will see next -- positions, FAST responses, ages, in order --, the detection / match counts, the distribution quality and the tracking
stability are identical.  The kernels (downscale, FAST, flow, motion estimate) are the oracle's own entry points on both sides; each has its
own parity tests.  Clips: steady tracking (features age, regions stop detecting, thresholds adapt), a scene cut (everything is lost and
re-detected), a restart, a flat stretch (too few features: the early-outs), both OBS presets."""
import numpy as np
import pytest

from tests import np_tracker, oracle_lib, synth


def _model(oracle, s, field):
    solver = oracle_lib.OracleMeshSolver(oracle, s.motion_width, s.motion_height, gen_region=(s.detection_width, s.detection_height),
                                         temporal=s.temporal_smoothing, local=s.local_smoothing) if field else None
    record = {}

    def fast(image, roi, threshold):
        return [tuple(int(v) for v in row) for row in oracle.fast(image, threshold, roi=roi)]

    def estimate(tracked, matched, homography):
        if field:
            rc, inl, _ = solver.solve(tracked, matched, region=(s.detection_width, s.detection_height), temporal=s.temporal_smoothing,
                                      threshold=s.acceptance_threshold)
            assert rc == 0
            return inl
        rc, H, mask = oracle.find_homography(tracked, matched, s.acceptance_threshold, region=(s.detection_width, s.detection_height), partial=not homography)
        record["H"] = H
        return mask

    t = np_tracker.Tracker(s, lambda f: oracle.luma_area_resize(f, s.detection_height, s.detection_width, channel=0), fast,
                           lambda p, c, pts: oracle.pyrlk(p, c, pts), estimate)
    return t, solver, record


def _compare(i, ost, model, record, field):
    st = ost.stats()
    assert st.n_detected == model.last["detected"] and st.n_matched == model.last["matched"], i
    assert np.float32(st.distribution) == model.last["distribution"], i
    assert np.float32(st.tracking_stability) == model.stability, i
    got = ost.features()
    want = np.array([[f.x, f.y, f.response, f.age] for f in model.features], np.float32).reshape(-1, 4)
    assert got.shape == want.shape and np.array_equal(got, want), i
    if model.last["estimated"] and not field:
        assert np.array_equal(np.array(st.homography[:]).reshape(3, 3), record["H"]), i


@pytest.mark.parametrize("preset", ["homography", "field"])
def test_feature_lists_frame_by_frame(oracle, preset):
    field = preset == "field"
    s = oracle_lib.preset(preset, predictive_samples=2, min_scene_quality=0.4, min_tracking_quality=0.2)
    a, _ = synth.make_clip(360, 640, 16, seed=21, jitter=1.5)
    b, _ = synth.make_clip(360, 640, 10, seed=77, jitter=1.5)                     # scene cut
    flat = np.full((3, 360, 640, 3), 128, np.uint8)                              # nothing to detect: the early-outs
    frames = np.concatenate([a, b, flat, b[:6]])
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)       # the OBS plugin's order
    model, solver, record = _model(oracle, s, field)
    estimated = aged = 0
    for i, f in enumerate(frames):
        if i == 9:
            ost.restart(); model.restart()
            if solver:
                solver.reset()
        ost.push(f, ts=i)
        model.track(f)
        _compare(i, ost, model, record, field)
        estimated += model.last["estimated"]
        aged = max(aged, max((ft.age for ft in model.features), default=0))
    ost.close()
    if solver:
        solver.close()
    assert estimated >= 24 and aged >= 5                                     # steady tracking happened, features aged


def test_forced_detection_and_small_grid(oracle):
    """force_detection (every region, every frame), a 4-column suppression grid (distribution quality = the map load) and a feature target
    far below what FAST finds (the regions' thresholds climb)."""
    s = oracle_lib.preset("homography", predictive_samples=1, force_detection=1, max_feature_density=0.009, min_feature_density=0.004,
                          accumulation_rate=0.5, min_motion_samples=4, uniformity_threshold=0.0, min_scene_quality=0.0, min_tracking_quality=0.0)
    frames, _ = synth.make_clip(360, 640, 8, seed=3, jitter=1.0)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    model, _, record = _model(oracle, s, False)
    assert model.detector.grid.cols == 4
    thresholds = set()
    for i, f in enumerate(frames):
        ost.push(f, ts=i)
        model.track(f)
        _compare(i, ost, model, record, False)
        thresholds.update(r["threshold"] for r in model.detector.regions)
    ost.close()
    assert len(thresholds) >= 4
