"""Row a7: how far is the specification's sparse optical flow from OpenCV's, where the two are allowed to differ?

oracle/pyrlk.cpp (and the HIP kernel, bit for bit) forms the window sums A11 / A12 / A22 / b1 / b2 EXACTLY in integers and converts them to
binary32 once; OpenCV 4.8's LKTrackerInvoker (video/src/lkpyramid.cpp; called at Vision/FrameTracker.cpp:42-48,140-146) accumulates them
in binary32, in an order that depends on the SIMD path taken: the scalar loop adds (float)(ixval * ixval) in window row-major order; the
universal-intrinsic loop takes 8 elements per step, pre-adds adjacent products exactly (v_dotprod on the int16 operands), adds the 4
resulting floats lane-wise into 4 accumulators, leaves the last 3 elements of an 11-wide row to the scalar loop and reduces the lanes at
the end.  lvko_pyrlk_float restates those orders (lanes 1 / 4 / 8 / 16, with and without the pairing) around the SAME fixed-point patch
arithmetic.  SURVEY.md section 8c's bar for this stage: |d| <= 0.01 px for >= 99.9 % of the points, status flags identical except <= 0.1 %.

Measured on SURVEY 8d's 600-frame clip (960 x 540 render, tracking at 480 x 270; every 6th frame pair, FAST corners of the previous
tracking frame + 64 random points per pair: ~105 000 tracked points) -- `pytest -s` prints the table, DESIGN.md section 2 quotes it."""
import numpy as np
import pytest

from tests import clipgen

MODES = [(1, 0), (4, 1), (4, 0), (8, 1), (8, 0), (16, 1)]


@pytest.fixture(scope="module")
def tracked(oracle):
    import torch
    torch.set_num_threads(8)
    rows, cols, n = 540, 960, 600
    clip = clipgen.Clip(rows, cols, n, cut_at=300)
    rng = np.random.default_rng(7)
    oracle.set_num_threads(8)
    res = {m: [] for m in MODES}
    spec = []
    for i in range(6, n, 6):
        if i == 300:
            continue                                                      # the scene cut: nothing to track
        prev = oracle.luma_area_resize(clip.render444(i - 1).numpy(), 270, 480)
        nxt = oracle.luma_area_resize(clip.render444(i).numpy(), 270, 480)
        kp = oracle.fast(prev, 15)
        pts = kp[:, :2].astype(np.float32)
        if len(pts) > 1100:
            pts = pts[rng.choice(len(pts), 1100, replace=False)]
        pts = np.concatenate([pts, rng.uniform([0, 0], [480, 270], (64, 2)).astype(np.float32)])
        out, st = oracle.pyrlk(prev, nxt, pts)
        spec.append((out, st))
        for m in MODES:
            res[m].append(oracle.pyrlk_float(prev, nxt, pts, m[0], m[1]))
    oracle.set_num_threads(1)
    cat = lambda xs: (np.concatenate([x[0] for x in xs]), np.concatenate([x[1] for x in xs]))
    return cat(spec), {m: cat(v) for m, v in res.items()}


def test_float_accumulation_orders_stay_within_the_survey_bound(tracked):
    (sp, ss), res = tracked
    n = len(ss)
    assert n > 90000 and ss.mean() > 0.8
    print(f"\n  {n} points, {100 * ss.mean():.1f} % tracked by the specification")
    print("  lanes pairs   status flips      |d| <= 0.01 px    p99 |d|    p99.9 |d|    max |d|  (px, both tracked)")
    for (lanes, pairs), (fp, fs) in res.items():
        flips = float((fs != ss).mean())
        both = (fs == 1) & (ss == 1)
        d = np.abs(fp[both] - sp[both]).max(axis=1)
        within = float((d <= 0.01).mean())
        print(f"  {lanes:5d} {pairs:5d}   {100 * flips:10.4f} %   {100 * within:12.4f} %   {np.percentile(d, 99):8.5f}   {np.percentile(d, 99.9):9.5f}   {d.max():8.4f}")
        assert flips <= 0.001, (lanes, pairs, flips)
        assert within >= 0.999, (lanes, pairs, within)


def test_the_orders_really_differ(tracked):
    """The variants are not the specification in disguise: binary32 accumulation moves some results by a few 1e-4 px."""
    (sp, ss), res = tracked
    fp, fs = res[(4, 1)]
    both = (fs == 1) & (ss == 1)
    assert 0 < np.abs(fp[both] - sp[both]).max() < 0.5
    assert (fp[both] != sp[both]).any()
