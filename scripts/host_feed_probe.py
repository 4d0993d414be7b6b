"""Free-running loop over lvk_hip_stab_push_yuv420_host at 4K (the bench's pcie pass alone), for timelines:
    cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/hostfeed -- python scripts/host_feed_probe.py 300"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk  # noqa: E402
from tests import clipgen  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rows, cols = 2160, 3840
dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(dev)
ctx = lvk.Context(0, stream=ws)
filt = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
filt.configure(lvk.StabilizationFilterSettings.obs_preset("homography"))
filt.set_overlap(True)
clip = clipgen.Clip(rows, cols, 32, device=dev)
hin = [filt.host_planes(rows, cols) for _ in range(32)]
for k in range(32):
    for d, p in zip(hin[k], clip.render_i420(k)):
        d[...] = p.cpu().numpy()
hout = [filt.host_planes(rows, cols) for _ in range(4)]
ia = [filt.prepare_yuv420_host(p) for p in hin]; oa = [filt.prepare_yuv420_host(p) for p in hout]
torch.cuda.synchronize()
for i in range(60):
    filt.apply_yuv420_host_prepared(ia[i % 32], i, oa[i & 3])
ctx.sync()
if os.environ.get("PROFILE") == "1":
    filt.set_profiling(True)
t0 = time.perf_counter(); stamps = []
look = os.environ.get("LOOKAHEAD") == "1"
if look:
    filt.prefetch_yuv420_host_prepared(ia[60 % 32])
for i in range(60, 60 + n):
    if look and i + 1 < 60 + n:
        filt.prefetch_yuv420_host_prepared(ia[(i + 1) % 32])     # frame i + 1 goes onto the link while frame i is tracked
    filt.apply_yuv420_host_prepared(ia[i % 32], i, oa[i & 3])
    stamps.append(time.perf_counter())
ctx.sync()
dt = time.perf_counter() - t0
import numpy as np
d = np.diff(np.array(stamps)) * 1e3
print("host-fed free running: %.0f frames/s, push p50 %.3f p90 %.3f ms" % (n / dt, np.percentile(d, 50), np.percentile(d, 90)))
if os.environ.get("PROFILE") == "1":
    print("stage us:", {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in filt.profile().items()})
filt.close(); ctx.close()
