"""Row a9: how far is the product's robust estimator from the REFERENCE's?

The product's specification (oracle/ransac.cpp, frozen) is a deterministic 128-hypothesis MSAC + least-squares refits; the reference
calls cv::findHomography(UsacParams{MAGSAC, LO_SIGMA, 50 iterations, confidence 0.99, LO 10 x 20, no final polish}) and, for badly
distributed features, cv::estimateAffinePartial2D(RANSAC, thr, 50) (Vision/FrameTracker.cpp:337-371).  oracle/usac_ref.cpp restates
those two from the published OpenCV 4.8 algorithm (unpinned: the binary is absent).  This test runs both over the point sets the tracker
produces on SURVEY 8d's 600-frame clip (960 x 540 render, tracking at 480 x 270, scene cut at 300) and reports the maximum displacement
of the four corners of the 480 x 270 tracking frame between the two homographies, and of each against the clip's ground truth.

Measured over the 598 point sets (this file, `pytest -s` prints the table; DESIGN.md section 2 quotes it), px at 480 x 270:
                                                    p50     p95     p99     max
  product spec vs ground truth                      0.028   0.084   0.115   0.167
  reference leg (A: max_thr = thr) vs ground truth  0.068   0.190   0.317   0.703     <- a 20-point weighted DLT on a random inlier subset
  reference leg (B: max_thr = 7.5) vs ground truth  0.072   0.259   0.711   1.691
  product spec vs reference leg (A)                 0.075   0.188   0.297   0.686
  product spec vs reference leg (B)                 0.079   0.254   0.671   1.682
  reference leg WITHOUT its closing LO vs truth     0.685   3.70    5.04    6.16      <- the loop ends after 1-4 iterations: a raw 4-point model
i.e. the product's H sits INSIDE the reference estimator's own scatter around the truth: the 0.25 px bar of the round-2 VERDICT holds
at the 95th percentile under reading (A) (0.19 px), not for the maximum -- the reference's estimator is itself further than 0.25 px
from the truth on 2-5 % of the frames; the product's never is (max 0.17 px).  The asserts below encode exactly that."""
import json

import numpy as np
import pytest

from tests import clipgen, oracle_lib

CORNERS = np.array([[0, 0], [480, 0], [0, 270], [480, 270]], np.float64)


def _apply(H, p):
    q = np.c_[p, np.ones(len(p))] @ H.T
    return q[:, :2] / q[:, 2:]


def _disp(A, B):
    return float(np.linalg.norm(_apply(A, CORNERS) - _apply(B, CORNERS), axis=1).max())


@pytest.fixture(scope="module")
def point_sets(oracle):
    import torch
    torch.set_num_threads(8)
    rows, cols, n = 540, 960, 600
    clip = clipgen.Clip(rows, cols, n, cut_at=300)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("homography", predictive_samples=1))
    # tracking pixel t <-> frame pixel f (INTER_AREA box of k x k): f = k t + (k - 1) / 2
    kx, ky = cols / 480.0, rows / 270.0
    S = np.array([[kx, 0, (kx - 1) / 2], [0, ky, (ky - 1) / 2], [0, 0, 1]])
    Si = np.linalg.inv(S)
    sets = []
    for i in range(n):
        ost.push(clip.render444(i).numpy(), ts=i)
        p1, p2, est = ost.matches()
        if est in (1, 2) and i != 300:                                  # (frame 300: the cut, no ground-truth motion)
            sets.append((i, p1, p2, est, Si @ clip.motion(i) @ S))
    ost.close()
    return sets


def test_product_estimator_vs_reference_semantics_over_the_clip(oracle, point_sets):
    rows = []
    for i, p1, p2, est, Hgt in point_sets:
        if est != 1:
            continue
        rc, Hs, ms = oracle.find_homography(p1, p2, 3.0)
        _, Ha, ma, ita = oracle.usac_find_homography(p1, p2, 3.0, max_thr=3.0)        # reading (A): max threshold = acceptance threshold
        _, Hb, mb, itb = oracle.usac_find_homography(p1, p2, 3.0, max_thr=0.0)        # reading (B): max(7.5, threshold)
        _, Hn, mn, _ = oracle.usac_find_homography(p1, p2, 3.0, max_thr=3.0, final_lo=False)
        rows.append((_disp(Hs, Hgt), _disp(Ha, Hgt), _disp(Hb, Hgt), _disp(Hs, Ha), _disp(Hs, Hb), _disp(Hn, Hgt), ita,
                     float((ms != ma).mean()), float((ms != mb).mean()), len(p1)))
    r = np.array(rows)
    assert len(r) >= 560, len(r)                                        # the clip keeps the homography branch on ~all frames
    names = ["spec_vs_truth", "usacA_vs_truth", "usacB_vs_truth", "spec_vs_usacA", "spec_vs_usacB", "usac_without_final_lo_vs_truth", "usac_iterations",
             "mask_disagreement_A", "mask_disagreement_B", "pairs"]
    table = {nm: dict(zip(("p50", "p95", "p99", "max"), np.round(np.percentile(r[:, k], [50, 95, 99, 100]), 4).tolist())) for k, nm in enumerate(names)}
    print("\n[a9 reference-semantics bound, %d point sets, corner displacement in px at 480x270]\n%s" % (len(r), json.dumps(table, indent=1)))
    # the product's estimate is within 0.25 px of the truth on EVERY frame (SURVEY App. A.8's bar) ...
    assert table["spec_vs_truth"]["max"] <= 0.25
    # ... and at least as close to it as the reference's estimator on 4 frames in 5
    assert (r[:, 0] <= r[:, 1]).mean() >= 0.8 and (r[:, 0] <= r[:, 2]).mean() >= 0.8
    # distance between the two estimators: within 0.25 px at the 95th percentile (reading A; 0.30 for B); its tail IS the reference's own
    # scatter around the truth (triangle inequality: never more than the two distances to the truth together)
    for k, p95_bar, max_bar in ((3, 0.25, 1.0), (4, 0.30, 2.5)):
        assert np.percentile(r[:, k], 95) <= p95_bar, np.percentile(r[:, k], 95)
        assert r[:, k].max() <= max_bar, r[:, k].max()
        assert (r[:, k] <= r[:, 0] + r[:, k - 2] + 1e-9).all()
        assert np.percentile(r[:, k], 50) <= 0.1
    # the inlier masks agree on all but a sliver of the pairs (both test e^2 against threshold^2; the models differ by < 0.5 px)
    assert table["mask_disagreement_A"]["p99"] <= 0.01 and table["mask_disagreement_B"]["p99"] <= 0.01
    # why final_lo matters: the confidence bound ends these runs after 1-5 iterations; without the closing local optimisation the
    # reference would return a raw 4-point model, pixels away from the truth -- recorded, not asserted as the reference's behaviour
    assert table["usac_iterations"]["p99"] <= 10 and table["usac_without_final_lo_vs_truth"]["p50"] > 3 * table["usacA_vs_truth"]["p50"]


def test_reference_semantics_leg_on_known_homographies(oracle):
    """The restated USAC against ground truth with 30 % outliers: it is a working estimator in its own right (else the bound above
    would be a bound against a straw man), and deterministic off randomGeneratorState."""
    rng = np.random.default_rng(4)
    errs = []
    for k in range(40):
        th, s = rng.normal(0, 0.01), 1 + rng.normal(0, 0.01)
        H = np.array([[s * np.cos(th), -s * np.sin(th), rng.normal(0, 4)], [s * np.sin(th), s * np.cos(th), rng.normal(0, 4)], [rng.normal(0, 1e-5), rng.normal(0, 1e-5), 1.0]])
        p1 = np.c_[rng.uniform(0, 480, 700), rng.uniform(0, 270, 700)].astype(np.float32)
        p2 = _apply(H, p1) + rng.normal(0, 0.1, p1.shape)
        out = rng.random(700) < 0.3
        p2[out] += rng.uniform(-40, 40, (int(out.sum()), 2))
        rc, He, mask, it = oracle.usac_find_homography(p1, p2.astype(np.float32), 3.0, max_thr=3.0)
        rc2, He2, mask2, it2 = oracle.usac_find_homography(p1, p2.astype(np.float32), 3.0, max_thr=3.0)
        assert rc == rc2 and np.array_equal(He, He2) and it == it2
        assert mask[~out].mean() > 0.97 and mask[out].mean() < 0.1, k
        assert abs(He[2, 2] - 1.0) < 1e-12 and it <= 50
        errs.append(_disp(He, H))
    assert np.median(errs) < 0.25 and max(errs) < 1.0, (np.median(errs), max(errs))


def test_affine_fallback_vs_reference_semantics(oracle, point_sets):
    """The distribution <= 0.6 branch (FrameTracker.cpp:359-371): product spec (2-point hypotheses + LS refits) vs the restated
    estimateAffinePartial2D(RANSAC, thr, 50) + LM refine, on similarity motion with outliers and on the clip's own point sets."""
    rng = np.random.default_rng(9)
    worst = 0.0
    for k in range(30):
        th, s = rng.normal(0, 0.01), 1 + rng.normal(0, 0.01)
        H = np.array([[s * np.cos(th), -s * np.sin(th), rng.normal(0, 4)], [s * np.sin(th), s * np.cos(th), rng.normal(0, 4)], [0, 0, 1.0]])
        p1 = np.c_[rng.uniform(0, 200, 300), rng.uniform(0, 120, 300)].astype(np.float32)      # one corner of the frame: badly distributed
        p2 = _apply(H, p1) + rng.normal(0, 0.1, p1.shape)
        out = rng.random(300) < 0.25
        p2[out] += rng.uniform(-40, 40, (int(out.sum()), 2))
        _, Hs, ms = oracle.find_homography(p1, p2.astype(np.float32), 3.0, partial=True)
        _, Hr, mr = oracle.ref_estimate_affine_partial(p1, p2.astype(np.float32), 3.0)
        assert Hr[2, 0] == 0 and Hr[2, 1] == 0 and Hr[2, 2] == 1
        worst = max(worst, _disp(Hs, Hr))
        assert _disp(Hs, H) < 0.25 and _disp(Hr, H) < 0.25, k
        assert (ms != mr).mean() < 0.01
    d = [_disp(oracle.find_homography(p1, p2, 3.0, partial=True)[1], oracle.ref_estimate_affine_partial(p1, p2, 3.0)[1]) for _, p1, p2, _, _ in point_sets[::10]]
    print("\n[a9 affine fallback] synthetic worst %.4f px; clip sets (every 10th) p50 %.4f max %.4f px" % (worst, np.median(d), max(d)))
    assert worst < 0.1 and max(d) < 0.1         # both end in a least-squares fit on (nearly) the same inliers
