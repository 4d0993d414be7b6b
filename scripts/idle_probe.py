"""What a push costs after the GPU has been idle for a while (a 60 fps source leaves 16 ms between frames; the barrier in front of bench.py's timed
region leaves ~0.3 ms): synchronise, wait `gap`, push, synchronise -- median / p90 of the push's own time and of push + sync, per gap.
Usage: python scripts/idle_probe.py [rows cols]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk  # noqa: E402
import bench  # noqa: E402


def main():
    rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
    dev = torch.device("cuda", 0)
    rig = bench.Rig(lvk, 0, dev, 0x4C564B31, rows, cols, "homography", "i420", "off", True, 64, cut=False, pingpong=True)
    for _ in range(1000):
        rig.step()
    rig.sync()
    print(f"{cols}x{rows} I420, overlap on; push = lvk_hip_stab_push_yuv420 alone, total = push + lvk_hip_sync")
    for gap_us in (0, 20, 100, 300, 1000, 5000, 16667):
        push, total = [], []
        for _ in range(80):
            rig.sync()
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < gap_us:
                pass
            t0 = time.perf_counter()
            rig.step()
            t1 = time.perf_counter()
            rig.sync()
            t2 = time.perf_counter()
            push.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
        c = rig.filt.schedule_counters(reset=True)
        print(f"idle {gap_us:6d} us: push p50 {np.percentile(push, 50):.3f} p90 {np.percentile(push, 90):.3f} ms;  push + sync p50 {np.percentile(total, 50):.3f} "
              f"p90 {np.percentile(total, 90):.3f} p99 {np.percentile(total, 99):.3f} ms;  schedule {c['push_synchronised']} synchronised / {c['push_free_running']} free-running, "
              f"{c['wait_signal_word']} word / {c['wait_event']} event / {c['wait_word_timeout']} timeouts")
    # the pattern in front of bench.py's timed region: free-running pushes, a device-wide synchronisation, then free-running pushes again
    firsts = []
    for _ in range(30):
        for _ in range(30):
            rig.step()
        rig.sync(); torch.cuda.synchronize()
        ts = [time.perf_counter()]
        for _ in range(6):
            rig.step(); ts.append(time.perf_counter())
        firsts.append(np.diff(ts) * 1e3)
    firsts = np.array(firsts)
    print("after a device-wide sync, free-running pushes 1..6 (median ms):", np.round(np.median(firsts, axis=0), 3), " p90:", np.round(np.percentile(firsts, 90, axis=0), 3))
    rig.close()


if __name__ == "__main__":
    main()
