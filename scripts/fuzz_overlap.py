"""One-off fuzz: overlap-mode 4:2:0 stabilizer vs the oracle at random even frame sizes (persistent and full remap grids, both presets,
I420 / NV12, frame delays 1..4), the GPU pushes FREE RUNNING (no synchronisation between them) with the 4:2:0 conversion pinned to the
tracking stream, pinned to the bulk stream, or placed per push."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk
from tests import oracle_lib, synth
from tests.test_stabilizer_gpu import _to_settings
oracle = oracle_lib.load()
ctx = lvk.Context(0, stream=torch.cuda.Stream())
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "7")))
bad = 0
for trial in range(int(os.environ.get("FUZZ_TRIALS", "18"))):
    rows = int(rng.integers(300, 1080)) * 2; cols = int(rng.integers(500, 1920)) * 2
    nv12 = bool(trial & 1)
    n = 14
    placement = ("tracker", "bulk", "")[trial % 3]
    if placement: os.environ["LVK_HIP_INGEST_PLACEMENT"] = placement
    else: os.environ.pop("LVK_HIP_INGEST_PLACEMENT", None)
    small, _ = synth.make_clip(rows // 2, cols // 2, n, seed=trial + 100, jitter=1.0)
    frames = np.ascontiguousarray(small.repeat(2, axis=1).repeat(2, axis=2))
    delay = int(rng.integers(1, 5))
    s = oracle_lib.preset("homography" if trial % 4 else "field", predictive_samples=delay)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(_to_settings(s)); gst.set_overlap(True)
    host_planes = [oracle.egress_yuv420(f, nv12=nv12) for f in frames]
    dev_planes = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in planes) for planes in host_planes]
    torch.cuda.synchronize()
    gots = [gst.apply_yuv420(dev_planes[i], timestamp=i)[0] for i in range(n)]            # free running
    ctx.sync()
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(s)
    ok = True
    for i in range(n):
        want, _ = ost.push(oracle.ingest_yuv420(*host_planes[i]), ts=i, nthreads=32)
        if (want is None) != (gots[i] is None): ok = False; break
        if want is not None:
            ok = ok and all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(gots[i], oracle.egress_yuv420(want, nv12=nv12)))
    print(trial, (rows, cols), "nv12" if nv12 else "i420", "field" if trial % 4 == 0 else "homography", "delay", delay, "placement", placement or "auto", "OK" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
    ost.close(); gst.close()
print("mismatches:", bad)
