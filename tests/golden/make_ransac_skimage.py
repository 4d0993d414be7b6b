"""Third-party anchor for the robust global motion estimate: scikit-image's RANSAC + normalised-DLT least squares
(skimage.measure.ransac with ProjectiveTransform: min_samples 4, residual_threshold = the acceptance threshold, the model refitted to all
inliers) on point sets the tracker produces on SURVEY 8d's clip.  Neither OpenCV's USAC nor this repository's specification -- a textbook
robust homography by other hands.  tests/test_ransac_third_party.py bounds the product specification's H against it.

Two steps, two interpreters (the test interpreter has torch but no scikit-image, the image's conda interpreter the reverse):
    python tests/golden/make_ransac_skimage.py --dump            # every 15th point set of the 600-frame clip + ground truth -> /tmp/ransac_sets.npz
    /opt/conda/bin/python3.9 tests/golden/make_ransac_skimage.py --solve     # adds skimage's H per set -> tests/golden/ransac_skimage.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TMP = "/tmp/ransac_sets.npz"

if "--dump" in sys.argv:
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import torch
    from tests import clipgen, oracle_lib
    torch.set_num_threads(8)
    oracle = oracle_lib.load()
    rows, cols, n = 540, 960, 600
    clip = clipgen.Clip(rows, cols, n, cut_at=300)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("homography", predictive_samples=1))
    kx, ky = cols / 480.0, rows / 270.0
    S = np.array([[kx, 0, (kx - 1) / 2], [0, ky, (ky - 1) / 2], [0, 0, 1]]); Si = np.linalg.inv(S)
    out = {}
    k = 0
    for i in range(n):
        ost.push(clip.render444(i).numpy(), ts=i)
        p1, p2, est = ost.matches()
        if est == 1 and i != 300 and i % 15 == 7:
            out["p1_%d" % k] = p1; out["p2_%d" % k] = p2; out["truth_%d" % k] = Si @ clip.motion(i) @ S; out["frame_%d" % k] = np.array(i)
            k += 1
    out["count"] = np.array(k)
    np.savez_compressed(TMP, **out)
    print("dumped", k, "point sets to", TMP)
elif "--solve" in sys.argv:
    import warnings
    warnings.filterwarnings("ignore")
    import skimage
    from skimage.measure import ransac
    from skimage.transform import ProjectiveTransform
    d = np.load(TMP)
    out = {k: d[k] for k in d.files}
    for k in range(int(d["count"])):
        model, inl = ransac((d["p1_%d" % k].astype(np.float64), d["p2_%d" % k].astype(np.float64)), ProjectiveTransform, min_samples=4,
                            residual_threshold=3.0, max_trials=2000, random_state=1000 + k)
        H = model.params / model.params[2, 2]
        out["H_skimage_%d" % k] = H; out["inliers_%d" % k] = inl
    out["skimage_version"] = np.array(skimage.__version__)
    np.savez_compressed(os.path.join(HERE, "ransac_skimage.npz"), **out)
    print("solved", int(d["count"]), "sets with scikit-image", skimage.__version__)
else:
    print(__doc__)
