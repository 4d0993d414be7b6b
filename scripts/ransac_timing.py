import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, livevisionkit_amd as lvk
ctx = lvk.Context(0)
rng = np.random.default_rng(1)
for n, out_frac in ((700, 0.1), (700, 0.3), (400, 0.1)):
    a = np.c_[rng.uniform(5, 475, n), rng.uniform(5, 265, n)].astype(np.float32)
    H = np.array([[1.002, 0.003, 2.0], [-0.003, 1.001, -1.5], [1e-6, -2e-6, 1.0]])
    p = np.c_[a, np.ones(n)] @ H.T; b = (p[:, :2] / p[:, 2:]).astype(np.float32) + rng.normal(0, 0.2, (n, 2)).astype(np.float32)
    bad = rng.random(n) < out_frac; b[bad] += rng.uniform(-30, 30, (int(bad.sum()), 2)).astype(np.float32)
    for _ in range(2):
        rc, Hm, mask = ctx.estimate_global_motion(a, b, 8.0)
    print("n", n, "outliers", out_frac, "inliers", rc)
