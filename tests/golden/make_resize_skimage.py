"""Third-party anchors for two resampling conventions (scikit-image 0.18 in the image's conda environment; run with /opt/conda/bin/python3.9):
  * the 2x bilinear enlargement of the chroma planes (FrameIngest's cv::resize INTER_LINEAR, FrameIngest.cpp:494-557): skimage.transform.resize,
    order 1, edge mode -- pixel-centre alignment, i.e. phases .25 / .75 and clamped borders, in float;
  * the integer-factor box average of the tracking frame (cv::resize INTER_AREA, FrameTracker.cpp:117): skimage.transform.downscale_local_mean.
Writes tests/golden/resize_skimage.npz (inputs uint8, third-party outputs float64)."""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
import skimage  # noqa: E402
from skimage.transform import downscale_local_mean, resize  # noqa: E402

rng = np.random.default_rng(7)
chroma = rng.integers(0, 256, (18, 24), dtype=np.uint8)
up = resize(chroma.astype(np.float64), (36, 48), order=1, mode="edge", anti_aliasing=False, preserve_range=True)
luma = rng.integers(0, 256, (64, 96), dtype=np.uint8)
box8 = downscale_local_mean(luma.astype(np.float64), (8, 8))
box4 = downscale_local_mean(luma.astype(np.float64), (4, 4))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_skimage.npz")
np.savez_compressed(out, chroma=chroma, chroma_up2=up, luma=luma, box8=box8, box4=box4, skimage_version=np.array(skimage.__version__))
print("wrote", out)
