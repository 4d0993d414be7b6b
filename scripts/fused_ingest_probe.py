"""The 4:2:0 -> 4:4:4 conversion as side work of the remap (k_remap_*_420_ingest) against the two kernels launched one after the other: same bits, and the
time of  remap | conversion | remap then conversion | fused  at 4K, full grid and the persistent grid of the overlap mode (HIP events on the launch stream,
back-to-back launches, 12 distinct frames).  Usage: python scripts/fused_ingest_probe.py [rows cols]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk  # noqa: E402
from tests import synth  # noqa: E402


def timed(fn, iters=60):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
    ctx = lvk.Context(0)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    n = 12
    srcs = [torch.randint(0, 256, (rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)]
    planes = [(torch.randint(0, 256, (rows, cols), dtype=torch.uint8, device="cuda", generator=g),
               torch.randint(0, 256, (rows // 2, cols // 2), dtype=torch.uint8, device="cuda", generator=g),
               torch.randint(0, 256, (rows // 2, cols // 2), dtype=torch.uint8, device="cuda", generator=g)) for _ in range(n)]
    new = [torch.empty((rows, cols, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
    rng = np.random.default_rng(0)
    mesh2 = np.array([[[0.004, -0.003], [-0.002, 0.004]], [[0.003, 0.002], [-0.004, -0.002]]], np.float32)
    mesh16 = synth.random_mesh(16, 16, rng, amp=0.01)
    for name, mesh in (("homography", mesh2), ("mesh16", mesh16)):
        # parity: fused == separate, both grids
        for co in (False, True):
            ref_out = ctx.warpmesh_apply_yuv420(srcs[0], mesh)
            ref_new = ctx.ingest_yuv420(*planes[0])
            out, nf = ctx.warpmesh_apply_yuv420_ingest(srcs[0], mesh, planes[0], co=co)
            ctx.sync()
            ok = all(torch.equal(a, b) for a, b in zip(ref_out, out)) and torch.equal(ref_new, nf)
            print(f"{name} co={int(co)}: fused == separate: {ok}")
        t_remap = timed(lambda i: ctx.warpmesh_apply_yuv420(srcs[i % n], mesh))
        t_ing = timed(lambda i: ctx.ingest_yuv420(*planes[i % n], out=new[i % n]))
        def both(i):
            ctx.warpmesh_apply_yuv420(srcs[i % n], mesh); ctx.ingest_yuv420(*planes[i % n], out=new[i % n])
        t_both = timed(both)
        t_fused = timed(lambda i: ctx.warpmesh_apply_yuv420_ingest(srcs[i % n], mesh, planes[i % n], new_frame=new[i % n]))
        t_fused_co = timed(lambda i: ctx.warpmesh_apply_yuv420_ingest(srcs[i % n], mesh, planes[i % n], new_frame=new[i % n], co=True))
        byts = 4.5 * rows * cols
        print(f"{name}: remap_420 {t_remap:.1f} us   ingest {t_ing:.1f} us   remap then ingest {t_both:.1f} us   fused {t_fused:.1f} us (full grid) / {t_fused_co:.1f} us (persistent grid)")
        print(f"   algorithmic bytes: remap {byts / 1e6:.1f} MB -> {byts / t_remap / 1e6:.3f} TB/s;  fused {2 * byts / 1e6:.1f} MB -> {2 * byts / t_fused / 1e6:.3f} TB/s = {2 * byts / t_fused / 1e6 / 8 * 100:.1f} % of 8 TB/s")


if __name__ == "__main__":
    main()
