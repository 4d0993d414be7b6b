"""Third-party anchor for FAST-9/16 (OpenCV's source is not in /root/reference): for every pixel of a few synthetic
images, the LARGEST threshold at which scikit-image's corner_fast (n = 9: the segment test on the 16-pixel Bresenham circle, strict
comparisons, arcs that wrap around) still calls it a corner -- which is OpenCV's cornerScore by definition.  tests/test_fast_third_party.py
rebuilds OpenCV's detector from these maps (score >= threshold, strict 3 x 3 non-maximum suppression) and compares it with the oracle.

Run ONCE with the image's conda interpreter, which has scikit-image (the test interpreter does not):
    /opt/conda/bin/python3.9 tests/golden/make_fast_skimage.py
Writes tests/golden/fast_skimage.npz: images uint8 [k, rows, cols], score int16 [k, rows, cols] (-1: not a corner at any threshold)."""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
import skimage  # noqa: E402
from skimage.feature import corner_fast  # noqa: E402

rng = np.random.default_rng(20260927)
rows, cols = 96, 128
yy, xx = np.mgrid[0:rows, 0:cols]
images = []
# 1: checkerboard + rectangles + mild noise (clean corners), 2: uniform noise (every arc pattern), 3: smooth gradients with a few hard edges (low scores)
a = np.where(((xx // 11) + (yy // 9)) % 2 == 0, 60, 180).astype(np.int32)
a[20:50, 30:70] = 230; a[60:80, 90:120] = 15
a = np.clip(a + rng.integers(-6, 7, a.shape), 0, 255)
images.append(a.astype(np.uint8))
images.append(rng.integers(0, 256, (rows, cols), dtype=np.uint8))
c = (96 + 50 * np.sin(xx / 9.0) + 40 * np.cos(yy / 7.0) + 0.4 * xx).astype(np.int32)
c[40:44, :] += 35; c[:, 64:67] -= 30
images.append(np.clip(c, 0, 255).astype(np.uint8))
images = np.stack(images)

score = np.full(images.shape, -1, np.int16)
for k, img in enumerate(images):
    f = img.astype(np.float64)                       # integer-valued floats: the comparisons with pixel +- threshold are exact
    for t in range(254, -1, -1):
        resp = corner_fast(f, n=9, threshold=float(t))
        new = (resp > 0) & (score[k] < 0)
        score[k][new] = t
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fast_skimage.npz")
np.savez_compressed(out, images=images, score=score, skimage_version=np.array(skimage.__version__))
print("wrote", out, "corners at threshold 10:", [(score[k] >= 10).sum() for k in range(len(images))])
