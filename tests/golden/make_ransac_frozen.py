"""Writes tests/golden/ransac_frozen.npz: inputs AND outputs of the product's robust-estimator specification (oracle/ransac.cpp) on 28 point
sets, as it stood at the start of round 3.  The specification is FROZEN from here on (round-2 VERDICT: it had followed the kernel):
tests/test_oracle_frozen.py checks oracle/ransac.cpp against these arrays and checks the arrays against a digest written into the test,
so neither the specification nor this fixture can move without the other two noticing.  Run only to add cases, never to refresh outputs.
    python tests/golden/make_ransac_frozen.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib  # noqa: E402


def apply(H, p):
    q = np.c_[p, np.ones(len(p))] @ H.T
    return q[:, :2] / q[:, 2:]


def cases():
    rng = np.random.default_rng(0x46524F5A)
    out = []
    for k in range(28):
        n = int(rng.choice([4, 5, 9, 75, 130, 240, 511, 777, 1024, 1500]))
        region = [(480, 270), (256, 256), (320, 180)][k % 3]
        th, s = rng.normal(0, 0.01), 1 + rng.normal(0, 0.01)
        H = np.array([[s * np.cos(th), -s * np.sin(th), rng.normal(0, 4)], [s * np.sin(th), s * np.cos(th), rng.normal(0, 4)],
                      [rng.normal(0, 2e-5) * (k % 2), rng.normal(0, 2e-5) * (k % 2), 1.0]])
        p1 = np.c_[rng.uniform(0, region[0], n), rng.uniform(0, region[1], n)].astype(np.float32)
        p2 = apply(H, p1) + rng.normal(0, [0.05, 0.15, 0.4][k % 3], p1.shape)
        frac = [0.0, 0.1, 0.3, 0.55][k % 4]
        bad = rng.random(n) < frac
        p2[bad] += rng.uniform(-40, 40, (int(bad.sum()), 2))
        if k == 20:
            p2 = p1.copy()                                   # identity, exact
        if k == 21:
            p1[:] = p1[0]                                    # degenerate: all points coincide
        thr = [3.0, 10.0, 8.0, 1.0][k % 4]
        out.append((p1, p2.astype(np.float32), thr, region, bool(k % 5 != 4)))
    return out


if __name__ == "__main__":
    oracle = oracle_lib.load()
    arrays = {}
    for i, (p1, p2, thr, region, full) in enumerate(cases()):
        rc, H, mask = oracle.find_homography(p1, p2, thr, region=region, partial=not full)
        arrays[f"p1_{i}"] = p1; arrays[f"p2_{i}"] = p2
        arrays[f"cfg_{i}"] = np.array([thr, region[0], region[1], 1.0 if full else 0.0], np.float64)
        arrays[f"rc_{i}"] = np.int32(rc); arrays[f"H_{i}"] = H; arrays[f"mask_{i}"] = mask
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ransac_frozen.npz"), **arrays)
    print("wrote", len(arrays) // 6, "cases")
