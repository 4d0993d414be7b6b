"""How does the free-running 4K stream react to a slower host?  A busy-wait of d microseconds between two pushes (the caller's own work) for d in a sweep:
frames/s, the period, the schedule counters and the per-stage GPU times.  A period that grows by more than d says the two streams fall into another
phase relationship (what scripts/slowmode_probe.sh looks for on the pool's slow boxes)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import livevisionkit_amd as lvk  # noqa: E402


def main():
    rig = bench.Rig(lvk, 0, torch.device("cuda:0"), 5, 2160, 3840, "homography", "i420", "off", True, 64, cut=False, pingpong=True)
    for _ in range(600):
        rig.step()
    rig.sync()
    for d in (0, 3, 6, 10, 15, 20, 30, 45, 60, 0):
        rig.filt.schedule_counters(reset=True)
        rig.filt.set_profiling(True, every=4)
        rig.sync(); torch.cuda.synchronize()
        n = 1200
        t0 = time.perf_counter()
        for _ in range(n):
            rig.step()
            if d:
                t = time.perf_counter() + d * 1e-6
                while time.perf_counter() < t:
                    pass
        rig.sync()
        dt = time.perf_counter() - t0
        prof = rig.filt.profile(); rig.filt.set_profiling(False)
        c = rig.filt.schedule_counters(reset=True)
        print(f"delay {d:3d} us: {n / dt:7.0f} frames/s  period {dt / n * 1e6:6.1f} us (period - delay {dt / n * 1e6 - d:6.1f})  ingest on tracker / bulk {c['ingest_on_tracker']} / {c['ingest_on_bulk']}  "
              + " ".join(f"{k} {v[0] / v[1] * 1e3:.1f}" for k, v in prof.items() if v[1] and k in ("pyrlk", "motion", "remap", "ingest")))
    rig.close()


if __name__ == "__main__":
    main()
