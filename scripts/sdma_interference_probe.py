"""Does a host->device copy in flight slow the tracker's kernels?  The device-resident 4K stream (lvk_hip_stab_push_yuv420) alone, and next to
back-to-back 12.4 MB pinned-host -> HBM copies on another stream (what the look-ahead upload of lvk_hip_stab_push_yuv420_host does).
    python scripts/sdma_interference_probe.py [h2d|d2h|none]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk  # noqa: E402
from tests import clipgen  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "h2d"
rows, cols = 2160, 3840
dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(dev)
ctx = lvk.Context(0, stream=ws)
filt = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
filt.configure(lvk.StabilizationFilterSettings.obs_preset("homography"))
filt.set_overlap(True)
clip = clipgen.Clip(rows, cols, 32, device=dev)
planes = [clip.render_i420(k) for k in range(32)]
outs = [tuple((torch.empty(p.shape, dtype=p.dtype).pin_memory() if os.environ.get('OUT') == 'host' else torch.empty_like(p)) for p in planes[0]) for _ in range(4)]
pa = [filt.prepare_yuv420(p) for p in planes]; oa = [filt.prepare_yuv420(o) for o in outs]
torch.cuda.synchronize()
for i in range(60):
    filt.apply_yuv420_prepared(pa[i % 32], i, oa[i & 3])
ctx.sync()
nbytes = rows * cols * 3 // 2
hbuf = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
dbuf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(dev)
n = 300
filt.set_profiling(True)
if mode != "none":
    with torch.cuda.stream(side):
        for _ in range(260):                                   # ~70 ms of copies queued ahead
            if mode == "h2d":
                dbuf.copy_(hbuf, non_blocking=True)
            else:
                hbuf.copy_(dbuf, non_blocking=True)
t0 = time.perf_counter()
for i in range(60, 60 + n):
    filt.apply_yuv420_prepared(pa[i % 32], i, oa[i & 3])
ctx.sync()
dt = time.perf_counter() - t0
busy = not side.query()
prof = filt.profile()
print("out=%s copies %s (" % (os.environ.get("OUT", "device"), "") + "" if False else "out=" + os.environ.get("OUT", "device") + " copies %s (still running at the end: %s): %.0f frames/s; stage us: %s" % (mode, busy, n / dt, {k: round(1e3 * ms / max(cnt, 1), 1) for k, (ms, cnt) in prof.items()}))
torch.cuda.synchronize()
filt.close(); ctx.close()
