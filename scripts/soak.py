"""Soak + determinism check on one GPU: 30k 1080p I420 frames through the overlap-mode filter (device memory must not grow after the
warm-up), and two fresh filters over the same clip must emit identical planes.  Usage: python scripts/soak.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import livevisionkit_amd as lvk
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import clipgen
ctx = lvk.Context(0, stream=torch.cuda.Stream())
s = lvk.StabilizationFilterSettings.obs_preset("homography")
f = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); f.configure(s); f.set_overlap(True)
clip = clipgen.Clip(1080, 1920, 16, device=torch.device("cuda", 0), cut_at=None)
planes = [clip.render_i420(i) for i in range(16)]; torch.cuda.synchronize()
pa = [f.prepare_yuv420(p) for p in planes]; outs = [tuple(torch.empty_like(q) for q in planes[0]) for _ in range(4)]; oa = [f.prepare_yuv420(o) for o in outs]
f.set_profiling(True, stages=("remap",))
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time(); n = 30000; sums = []
for i in range(n):
    f.apply_yuv420_prepared(pa[i % 16], i, oa[i & 3])
    if i == 999:
        ctx.sync(); int(outs[0][0].sum().item())                 # (lets torch's caching allocator take its reduction scratch first)
        free1 = torch.cuda.mem_get_info()[0]
    if i % 10000 == 9999:
        ctx.sync(); sums.append(int(outs[i & 3][0].sum().item()))
ctx.sync()
print("frames/s %.0f" % (n / (time.time() - t0)), "device memory: warm-up %.1f MB, growth over the following 29k frames %.1f MB" % ((free0 - free1) / 1e6, (free1 - torch.cuda.mem_get_info()[0]) / 1e6), f.profile()["remap"], sums)
# determinism: two fresh filters over the same 60 frames give identical planes
def run():
    g = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); g.configure(s); g.set_overlap(True)
    acc = []
    for i in range(60):
        o = tuple(torch.empty_like(q) for q in planes[0])
        r, _ = g.apply_yuv420(planes[i % 16], timestamp=i, out=o)
        if r is not None:
            acc.append(o)
    ctx.sync(); g.close()
    return acc
a, b = run(), run()
print("deterministic:", all(torch.equal(x, y) for p, q in zip(a, b) for x, y in zip(p, q)), len(a))
