"""Third-party pin of the two image filters under the optical flow: cv::pyrDown and the Scharr derivative image that cv::buildOpticalFlowPyramid /
calcOpticalFlowPyrLK compute (Vision/FrameTracker.cpp:140-146).  OpenCV is not in this image, so the separable filtering and
the BORDER_REFLECT_101 edge handling are done by scipy.ndimage (correlate1d, mode="mirror" = reflect-101) -- a convolution engine and a border
rule that are not this repository's -- and only the published taps ([1 4 6 4 1] / 256 with round-half-up for pyrDown; [3 10 3] x [-1 0 1],
unnormalised int16, for Scharr) come from OpenCV's documentation.  The oracle must equal it exactly, on the sizes the tracker's pyramid has and on
odd / tiny ones where the mirror rule and the (n + 1) / 2 output size matter.  (round-5 VERDICT weak #7: third-party pins where the image has a counterpart.)"""
import numpy as np
import pytest
from scipy import ndimage

SHAPES = [(270, 480), (135, 240), (68, 120), (34, 60), (33, 47), (5, 7), (3, 3), (2, 9)]


@pytest.mark.parametrize("shape", SHAPES)
def test_pyr_down_equals_scipy_separable_filter(oracle, shape):
    img = np.random.default_rng(shape[0] * 1000 + shape[1]).integers(0, 256, shape, dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    acc = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    want = ((acc[::2, ::2] + 128) >> 8).astype(np.uint8)                # every second sample of the smoothed image, rounded half up
    got = oracle.pyr_down(img)
    assert got.shape == ((shape[0] + 1) // 2, (shape[1] + 1) // 2) == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("shape", SHAPES)
def test_scharr_derivatives_equal_scipy_separable_filter(oracle, shape):
    img = np.random.default_rng(7 + shape[0] * 1000 + shape[1]).integers(0, 256, shape, dtype=np.uint8).astype(np.int64)
    smooth, diff = np.array([3, 10, 3], np.int64), np.array([-1, 0, 1], np.int64)
    ix = ndimage.correlate1d(ndimage.correlate1d(img, smooth, axis=0, mode="mirror"), diff, axis=1, mode="mirror")
    iy = ndimage.correlate1d(ndimage.correlate1d(img, smooth, axis=1, mode="mirror"), diff, axis=0, mode="mirror")
    d = oracle.scharr_deriv(img.astype(np.uint8))
    assert np.array_equal(d[..., 0], ix.astype(np.int16)) and np.array_equal(d[..., 1], iy.astype(np.int16))
