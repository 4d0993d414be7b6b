"""Free-running cross-stream kernel timeline from in-kernel wall-clock stamps (library built with -DLVK_TIMELINE by
scripts/variant_build.sh timeline -DLVK_TIMELINE and selected with LVK_HIP_LIB).  rocprofv3's kernel trace slows the host enough to change how the tracker
and the bulk stream overlap; this costs two atomics per workgroup.  Usage: LVK_HIP_LIB=<timeline .so> python scripts/timeline_free.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import clipgen  # noqa: E402
import livevisionkit_amd as lvk  # noqa: E402

SLOTS, RING = 4, 8192
UNITS = {"imgproc": ["area", "pyramid"], "pyrlk": ["flow"], "motion": ["hypotheses", "finalize", "compact"], "ingest": ["ingest"], "remap": ["remap"]}


def main():
    rows, cols, pool, steps = 2160, 3840, 64, 700
    device = torch.device("cuda", 0)
    lvk.shard.bind_to_gpu_numa(0)
    ws = torch.cuda.Stream(device)
    ctx = lvk.Context(0, stream=ws)
    filt = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
    filt.configure(lvk.StabilizationFilterSettings.obs_preset("homography"))
    filt.set_overlap(True)
    clip = clipgen.Clip(rows, cols, 600, device=device, cut_at=None)                 # the bench's clip, first `pool` poses
    planes = [clip.render_i420(i) for i in range(pool)]
    torch.cuda.synchronize()
    outs = [tuple(torch.empty_like(p) for p in planes[0]) for _ in range(4)]
    pa = [filt.prepare_yuv420(p) for p in planes]; oa = [filt.prepare_yuv420(o) for o in outs]
    torch.cuda.synchronize()
    import time
    t0 = None
    for i in range(steps):
        if i == 200:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if os.environ.get("LVK_TIMELINE_ANNOUNCE"):                                  # round 4: every frame announced one push ahead
            filt.prefetch_yuv420_prepared(pa[(i + 1) % pool])
        filt.apply_yuv420_prepared(pa[i % pool], i, oa[i & 3])
    torch.cuda.synchronize()
    print(f"{(steps - 200) / (time.perf_counter() - t0):.0f} frames/s free-running")
    lib = lvk._native.load()
    ev = []
    for unit, names in UNITS.items():
        buf = (ctypes.c_longlong * (SLOTS * (1 + 2 * RING)))()
        fn = getattr(lib, "lvk_tl_read_" + unit)
        fn.restype = ctypes.c_int
        assert fn(buf) == 0
        a = np.frombuffer(buf, dtype=np.int64).reshape(SLOTS, 1 + 2 * RING)
        for s, name in enumerate(names):
            n = int(a[s, 0]); log = a[s, 1:].reshape(RING, 2)
            rec = sorted((int(log[k % RING, 0]), int(log[k % RING, 1])) for k in range(max(0, n - RING), n))
            # records of one launch start within a few us of each other (a wave of workgroups) or, for the long remap, within one
            # workgroup round (< 25 us); launches of the same kernel are a frame period apart
            cur = None
            for st, en in rec:
                if cur is None or st - cur[2] > 2500:
                    if cur:
                        ev.append((cur[0], cur[1], name))
                    cur = [st, en, st]
                else:
                    cur[1] = max(cur[1], en); cur[2] = st
            if cur:
                ev.append((cur[0], cur[1], name))
    ev.sort()
    # the last ~5 frames
    rem = [i for i, e in enumerate(ev) if e[2] == "remap"]
    first = rem[-7]
    base = ev[first][0]
    for s, e, name in ev[first:]:
        lane = "B" if name in ("remap", "ingest") else "T"
        print(f"{(s - base) / 100:9.2f} {(e - base) / 100:9.2f} {(e - s) / 100:7.2f} us  {lane}  {name}")


if __name__ == "__main__":
    main()
