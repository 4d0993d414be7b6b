"""Soak: tens of thousands of free-running pushes through the device and the host entry points of one filter each (restart / reconfigure in
between), device memory use before and after, and the last emitted frame of the host path against the device path's."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk  # noqa: E402
from tests import clipgen  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rows, cols = 1080, 1920
dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(dev)
ctx = lvk.Context(0, stream=ws)
clip = clipgen.Clip(rows, cols, 64, device=dev)
planes = [clip.render_i420(k) for k in range(64)]
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]


def device_run(count, announce=False):
    f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
    f.configure(lvk.StabilizationFilterSettings.obs_preset("homography")); f.set_overlap(True)
    outs = [tuple(torch.empty_like(p) for p in planes[0]) for _ in range(4)]
    pa = [f.prepare_yuv420(p) for p in planes]; oa = [f.prepare_yuv420(o) for o in outs]
    t0 = time.perf_counter()
    for i in range(count):
        if i == count // 2:
            f.restart()
        if i == count // 3:
            f.configure(lvk.StabilizationFilterSettings.obs_preset("field")); f.configure(lvk.StabilizationFilterSettings.obs_preset("homography"))
        if announce and i % 97 != 96:                          # round 4: frames announced one push ahead (now and then not, now and then wrongly)
            f.prefetch_yuv420_prepared(pa[(i + (5 if i % 89 == 88 else 1)) % 64])
        f.apply_yuv420_prepared(pa[i % 64], i, oa[i & 3])
    ctx.sync()
    dt = time.perf_counter() - t0
    last = [p.cpu().numpy().copy() for p in outs[(count - 1) & 3]]
    hits = f.lookahead_frames()
    f.close()
    return last, count / dt, hits


def host_run(count):
    f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
    f.configure(lvk.StabilizationFilterSettings.obs_preset("homography")); f.set_overlap(True)
    hin = [f.host_planes(rows, cols) for _ in range(64)]
    for k in range(64):
        for d, p in zip(hin[k], planes[k]):
            d[...] = p.cpu().numpy()
    hout = [f.host_planes(rows, cols) for _ in range(4)]
    ia = [f.prepare_yuv420_host(p) for p in hin]; oa = [f.prepare_yuv420_host(p) for p in hout]
    t0 = time.perf_counter()
    f.prefetch_yuv420_host_prepared(ia[0])
    for i in range(count):
        if i == count // 2:
            ctx.sync(); f.restart()
            f.prefetch_yuv420_host_prepared(ia[i % 64])       # restart() forgets the announced frames (round 4): announce this one again
        if i == count // 3:
            ctx.sync()
            f.configure(lvk.StabilizationFilterSettings.obs_preset("field")); f.configure(lvk.StabilizationFilterSettings.obs_preset("homography"))
        if i + 1 < count:
            f.prefetch_yuv420_host_prepared(ia[(i + 1) % 64])
        f.apply_yuv420_host_prepared(ia[i % 64], i, oa[i & 3])
    ctx.sync()
    dt = time.perf_counter() - t0
    last = [np.array(p) for p in hout[(count - 1) & 3]]
    f.close()
    return last, count / dt


a, fa, _ = device_run(n)
c, fc, hits = device_run(n, announce=True)
b, fb = host_run(n)
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
same = all(np.array_equal(x, y) for x, y in zip(a, b)) and all(np.array_equal(x, y) for x, y in zip(a, c))
print("soak: %d pushes each; device %.0f frames/s, device with announcements %.0f frames/s (%d pushes found their pyramid built), host %.0f frames/s; "
      "last frames identical: %s; device memory delta %+.1f MB" % (n, fa, fc, hits, fb, same, (free0 - free1) / 1e6))
ctx.close()
sys.exit(0 if same and abs(free0 - free1) < 64e6 else 1)
