"""Row a9 against a THIRD party: scikit-image's RANSAC + normalised-DLT least squares (tests/golden/make_ransac_skimage.py; min_samples 4,
residual threshold 3 px, refit on all inliers) on 40 point sets the tracker produces on SURVEY 8d's clip -- a textbook robust homography by
other hands, neither OpenCV's USAC nor this repository's specification.  Measured (px, displacement of the corners of the 480 x 270 frame;
`pytest -s` prints the table):  product spec vs scikit-image p50 0.000 / p95 0.069 / max 0.277 (most sets: the SAME inliers, the same least squares);
scikit-image vs ground truth p50 0.031 / max 0.330;  product spec vs ground truth p50 0.038 / max 0.107;  inlier flags equal on 100.0 % of the pairs.
The one set on which the two differ by 0.28 px is the one on which scikit-image itself is 0.33 px from the truth (the product: 0.11)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNERS = np.array([[0, 0], [480, 0], [0, 270], [480, 270]], np.float64)


def _apply(H, p):
    q = np.c_[p, np.ones(len(p))] @ H.T
    return q[:, :2] / q[:, 2:]


def _disp(A, B):
    return float(np.linalg.norm(_apply(A, CORNERS) - _apply(B, CORNERS), axis=1).max())


def test_product_specification_agrees_with_scikit_image(oracle):
    d = np.load(os.path.join(ROOT, "tests", "golden", "ransac_skimage.npz"))
    n = int(d["count"])
    assert n == 40
    vs_sk, sk_truth, spec_truth, agree = [], [], [], []
    for k in range(n):
        p1, p2 = d["p1_%d" % k], d["p2_%d" % k]
        rc, H, mask = oracle.find_homography(p1, p2, 3.0)
        assert rc > 100
        Hs, T = d["H_skimage_%d" % k], d["truth_%d" % k]
        vs_sk.append(_disp(H, Hs)); sk_truth.append(_disp(Hs, T)); spec_truth.append(_disp(H, T))
        agree.append(float((mask.astype(bool) == d["inliers_%d" % k]).mean()))
    vs_sk, sk_truth, spec_truth = np.array(vs_sk), np.array(sk_truth), np.array(spec_truth)
    print("\nproduct spec vs scikit-image: p50 %.3f p95 %.3f max %.3f | scikit-image vs truth: p50 %.3f max %.3f | spec vs truth: p50 %.3f max %.3f | "
          "same inlier flags: %.1f %% of the pairs" % (np.median(vs_sk), np.percentile(vs_sk, 95), vs_sk.max(), np.median(sk_truth), sk_truth.max(),
                                                        np.median(spec_truth), spec_truth.max(), 100 * np.mean(agree)))
    assert np.percentile(vs_sk, 95) <= 0.25 and vs_sk.max() <= 0.5
    assert spec_truth.max() <= max(sk_truth.max(), 0.25)               # and it is no further from the truth than the third party is
    assert np.mean(agree) > 0.97
