"""FAST-9/16 pinned to a THIRD-PARTY implementation.  OpenCV's source is not in /root/reference and cv2 is not in
the image, but scikit-image 0.18 is (in the image's conda environment): tests/golden/make_fast_skimage.py swept its corner_fast (n = 9) over
all thresholds and kept, per pixel, the largest threshold at which it is still a corner -- OpenCV's cornerScore by definition.  Here OpenCV's
detector is rebuilt from those maps -- corners: score >= threshold; FAST_t's non-maximum suppression: strictly greater than all eight
neighbours, non-corners counting as 0; row-major order -- and the oracle's keypoints (position AND score) must be exactly that list, for
five thresholds on three images, whole image and a region of interest (whose edge acts as the image edge)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "fast_skimage.npz"))


def _opencv_fast_from_scores(score, threshold):
    s = np.where(score >= threshold, score, 0).astype(np.int32)              # the detector's score buffer: 0 where the pixel is no corner
    p = np.pad(s, 1)
    rows, cols = s.shape
    keep = s > 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy or dx:
                keep &= s > p[1 + dy:1 + dy + rows, 1 + dx:1 + dx + cols]
    ys, xs = np.nonzero(keep)                                                 # row-major
    return np.stack([xs, ys, s[ys, xs]], 1).astype(np.int32)


@pytest.mark.parametrize("k", [0, 1, 2])
@pytest.mark.parametrize("threshold", [1, 10, 20, 40, 90])
def test_fast_equals_the_detector_rebuilt_from_skimage_scores(oracle, k, threshold):
    img, score = GOLD["images"][k], GOLD["score"][k]
    want = _opencv_fast_from_scores(score, threshold)
    got = oracle.fast(img, threshold)
    assert len(want) > 0 or threshold >= 40                                   # (the smooth image has no corner of score 40)
    assert np.array_equal(got, want)


def test_a_zero_score_corner_is_never_reported(oracle):
    """cornerScore 0 (a corner at threshold 0 only) cannot pass score > neighbours >= 0: neither implementation reports it."""
    img, score = GOLD["images"][2], GOLD["score"][2]
    got = oracle.fast(img, 0)
    assert (got[:, 2] > 0).all() and np.array_equal(got, _opencv_fast_from_scores(score, 0))


@pytest.mark.parametrize("roi", [(7, 5, 64, 48), (64, 0, 64, 96), (3, 40, 30, 20)])
def test_fast_on_a_region_of_interest(oracle, roi):
    """FeatureDetector runs FAST on sub-matrices (FeatureDetector.cpp:130-134): the region's edge acts as the image edge -- no corner within 3 pixels
    of it, neighbours beyond it count as 0.  A pixel's score depends on its own circle only, so the full-image score map, cropped, serves."""
    x, y, w, h = roi
    img, score = GOLD["images"][1], GOLD["score"][1]
    local = score[y:y + h, x:x + w].copy()
    local[:3] = -1; local[-3:] = -1; local[:, :3] = -1; local[:, -3:] = -1
    assert np.array_equal(oracle.fast(img, 15, roi=roi), _opencv_fast_from_scores(local, 15))
