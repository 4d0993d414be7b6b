"""One-off fuzz: overlap-mode 4:2:0 stabilizer vs the oracle at random even frame sizes (persistent and full remap grids, both presets, I420 / NV12)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk
from tests import oracle_lib, synth
from tests.test_stabilizer_gpu import _to_settings
oracle = oracle_lib.load()
ctx = lvk.Context(0)
rng = np.random.default_rng(7)
bad = 0
for trial in range(14):
    rows = int(rng.integers(300, 800)) * 2; cols = int(rng.integers(500, 1400)) * 2
    nv12 = bool(trial & 1)
    n = 6
    small, _ = synth.make_clip(rows // 2, cols // 2, n, seed=trial + 100, jitter=1.0)
    frames = np.ascontiguousarray(small.repeat(2, axis=1).repeat(2, axis=2))
    s = oracle_lib.preset("homography" if trial % 3 else "field", predictive_samples=2)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(_to_settings(s)); gst.set_overlap(True)
    wants, gots = [], []
    for i, f in enumerate(frames):
        planes = oracle.egress_yuv420(f, nv12=nv12)
        want, _ = ost.push(oracle.ingest_yuv420(*planes), ts=i, nthreads=32)
        got, _ = gst.apply_yuv420(tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in planes), timestamp=i)
        assert (want is None) == (got is None)
        if want is not None:
            wants.append(oracle.egress_yuv420(want, nv12=nv12)); gots.append(got)
    ctx.sync()
    ok = all(np.array_equal(a.cpu().numpy(), b) for w, g in zip(wants, gots) for a, b in zip(g, w))
    strips = ((cols + 255) // 256) * ((rows + 3) // 4)
    print(trial, (rows, cols), "nv12" if nv12 else "i420", "field" if trial % 3 == 0 else "homography", "strips", strips, "OK" if ok else "MISMATCH", len(wants))
    bad += 0 if ok else 1
    ost.close(); gst.close()
print("mismatches:", bad)
