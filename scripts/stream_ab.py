"""One free-running 4K I420 stream (bench.Rig, the pipeline's own launches) under the library LVK_HIP_LIB selects: frames/s over 3 x 1 200 pushes after 600 of
fill, the schedule counters and the per-stage GPU times (HIP events of one push in four).  A / B partner runs: scripts/stream_variants.sh.
Usage: python scripts/stream_ab.py [rows cols [preset [lens]]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import livevisionkit_amd as lvk  # noqa: E402


def main():
    rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
    preset = sys.argv[3] if len(sys.argv) > 3 else "homography"
    lens = sys.argv[4] if len(sys.argv) > 4 else "off"
    rig = bench.Rig(lvk, 0, torch.device("cuda:0"), 5, rows, cols, preset, "i420", lens, True, 64, cut=False, pingpong=True)
    for _ in range(600):
        rig.step()
    rig.sync()
    for _ in range(3):
        rig.filt.schedule_counters(reset=True)
        rig.filt.set_profiling(True, every=4)
        rig.sync(); torch.cuda.synchronize()
        n = 1200
        t0 = time.perf_counter()
        for _ in range(n):
            rig.step()
        rig.sync()
        dt = time.perf_counter() - t0
        prof = rig.filt.profile(); rig.filt.set_profiling(False)
        c = rig.filt.schedule_counters(reset=True)
        print(f"{n / dt:7.0f} frames/s  period {dt / n * 1e6:6.1f} us   persistent / full grids {c['remap_persistent']} / {c['remap_full']}   "
              + " ".join(f"{k} {v[0] / v[1] * 1e3:.1f}" for k, v in prof.items() if v[1] and k in ("downscale", "pyramid", "fast", "pyrlk", "motion", "remap", "ingest")))
    rig.close()


if __name__ == "__main__":
    main()
