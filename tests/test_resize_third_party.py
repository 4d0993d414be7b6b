"""Two resampling conventions against a third party (scikit-image, fixtures by tests/golden/make_resize_skimage.py): the oracle's fixed-point
arithmetic may differ from the third party's float result by rounding only."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "resize_skimage.npz"))


def test_chroma_enlargement_has_the_third_partys_alignment(oracle):
    """cv::resize(INTER_LINEAR) x2 of a chroma plane: pixel centres aligned (phases .25 / .75), borders clamped -- skimage.transform.resize(order 1,
    edge) in float; the oracle's 11-bit fixed point stays within 1 LSB of it everywhere and is its rounding on > 90 % of the samples."""
    u = GOLD["chroma"]
    f = oracle.ingest_yuv420(np.zeros((36, 48), np.uint8), u, u)
    d = np.abs(f[..., 1].astype(np.float64) - GOLD["chroma_up2"])
    assert d.max() < 1.0 and (f[..., 1] == np.clip(np.rint(GOLD["chroma_up2"]), 0, 255)).mean() > 0.9


def test_integer_area_downscale_is_the_third_partys_box_mean(oracle):
    """cv::resize(INTER_AREA) by 8 and by 4: the box mean (skimage.transform.downscale_local_mean), rounded -- ties aside, the same integers."""
    luma = GOLD["luma"]
    for k, key in ((8, "box8"), (4, "box4")):
        got = oracle.luma_area_resize(luma, luma.shape[0] // k, luma.shape[1] // k).astype(np.float64)
        want = GOLD[key]
        tie = np.abs(want - np.floor(want) - 0.5) < 1e-9
        assert np.array_equal(got[~tie], np.rint(want[~tie])) and np.abs(got - want).max() <= 0.5
