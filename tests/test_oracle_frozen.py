"""The product's robust-estimator specification (oracle/ransac.cpp) is FROZEN since the start of round 3.

Round 2 changed the specification together with the kernel and regenerated the golden (commit 9310ed2): an oracle that follows the
product is a regression test, not a parity oracle (round-2 VERDICT, weak #1).  From here on:
  * tests/golden/ransac_frozen.npz holds inputs and outputs (H bit for bit, mask, inlier count) of 28 point sets -- homography and
    affine fallback, 4 ... 1500 pairs, 0 ... 55 % outliers, four thresholds, three regions, a degenerate and an exact case;
  * this test checks oracle/ransac.cpp against them, and the arrays against the digest below -- a regenerated fixture fails here;
  * tests/test_golden.py / tests/test_stabilizer_gpu.py check the HIP kernels against the same specification.
A change of the estimator therefore needs an edit of FROZEN_DIGEST in this file, in the open."""
import hashlib
import os

import numpy as np

from tests import oracle_lib

FROZEN_DIGEST = "72502fbca14a4777b1eb281dcab7b70e0a42dae846333b52732ef2765174e8df"
PATH = os.path.join(oracle_lib.ROOT, "tests", "golden", "ransac_frozen.npz")


def _cases():
    d = np.load(PATH)
    n = len([k for k in d.files if k.startswith("rc_")])
    return d, n


def test_frozen_fixture_is_the_committed_one():
    d, n = _cases()
    h = hashlib.sha256()
    for i in range(n):
        for k in ("p1", "p2", "cfg", "rc", "H", "mask"):
            h.update(np.ascontiguousarray(d[f"{k}_{i}"]).tobytes())
    assert n == 28 and h.hexdigest() == FROZEN_DIGEST, h.hexdigest()


def test_specification_still_returns_the_frozen_outputs(oracle):
    d, n = _cases()
    for i in range(n):
        thr, rw, rh, full = d[f"cfg_{i}"]
        rc, H, mask = oracle.find_homography(d[f"p1_{i}"], d[f"p2_{i}"], thr, region=(rw, rh), partial=not bool(full))
        assert rc == int(d[f"rc_{i}"]), i
        assert np.array_equal(H.view(np.uint64), d[f"H_{i}"].view(np.uint64)), (i, np.abs(H - d[f"H_{i}"]).max())
        assert np.array_equal(mask, d[f"mask_{i}"]), i


def test_round2_motion_golden_unchanged(oracle):
    """The single point set of tests/golden/motion.npz as round 2 left it (its outputs by digest)."""
    d = np.load(os.path.join(oracle_lib.ROOT, "tests", "golden", "motion.npz"))
    h = hashlib.sha256()
    for k in ("p1", "p2", "rc_h", "H_h", "mask_h", "rc_a", "H_a", "mask_a"):
        h.update(np.ascontiguousarray(d[k]).tobytes())
    assert h.hexdigest() == "6e30075f1242491a1bf016caeff637a50f6713b87866b0aa28aa0262c6740f1e", h.hexdigest()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_hip_estimator_returns_the_frozen_outputs(ctx):
    """k_ransac_hypotheses / k_ransac_finalize against the frozen outputs directly (no oracle in the loop)."""
    d, n = _cases()
    for i in range(n):
        thr, rw, rh, full = d[f"cfg_{i}"]
        if int(d[f"rc_{i}"]) < 0:                    # no hypothesis at all (all points coincide): error paths are compared elsewhere
            continue
        rc, H, mask = ctx.estimate_global_motion(d[f"p1_{i}"], d[f"p2_{i}"], thr, region=(rw, rh), full_homography=bool(full))
        assert rc == int(d[f"rc_{i}"]), i
        assert np.array_equal(np.asarray(H).reshape(3, 3).view(np.uint64), d[f"H_{i}"].view(np.uint64)), i
        assert np.array_equal(np.asarray(mask), d[f"mask_{i}"]), i
