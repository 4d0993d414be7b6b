"""How far are the HIP kernels and the CPU oracle from the reference's OWN compiled OpenCL kernels (oracle/_ref)?
Prints mismatch counts / max |d| / PSNR per case and times the reference kernel beside the product's at 4K.
Usage (GPU box): python scripts/ref_pin_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import livevisionkit_amd as lvk  # noqa: E402
from tests import oracle_lib, ref_cl, synth  # noqa: E402


def stats(a, b):
    a = a.astype(np.int32); b = b.astype(np.int32)
    d = np.abs(a - b)
    n = int((d.max(axis=-1) > 0).sum()) if d.ndim == 3 else int((d > 0).sum())
    return n, int(d.max()), synth.psnr(a, b)


def time_gpu(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ref = ref_cl.RefKernels()
    oracle = oracle_lib.load()
    ctx = lvk.Context(0)
    rng = np.random.default_rng(11)
    for (rows, cols) in [(360, 640), (1080, 1920), (2160, 3840)]:
        src = synth.textured_frame(rows, cols, seed=rows)
        d = torch.from_numpy(src).cuda()
        for yuv in (True, False):
            H = synth.random_homography(rows, cols, rng, strength=1.0)
            g = ctx.remap_homography(d, H, bg=(3, 200, 77), yuv=yuv); ctx.sync()
            r = ref.remap_homography(d, H, bg=(3, 200, 77), yuv=yuv); torch.cuda.synchronize()
            print(f"homography {cols}x{rows} yuv={yuv}: hip vs ref   px differ {stats(g.cpu().numpy(), r.cpu().numpy())}", flush=True)
            if rows <= 1080:
                o = oracle.remap_homography(src, H, bg=(3, 200, 77), yuv=yuv)
                print(f"                                oracle vs ref {stats(o, r.cpu().numpy())}   hip vs oracle {stats(g.cpu().numpy(), o)}", flush=True)
        # map path (field preset): product interpolates the mesh in-kernel; the reference reads a materialised map
        mesh = synth.random_mesh(16, 16, rng, amp=0.01)
        omap = oracle.mesh_to_map(mesh, rows, cols) if hasattr(oracle, "mesh_to_map") else None
        if omap is not None:
            dm = torch.from_numpy(np.ascontiguousarray(omap)).cuda()
            g = ctx.remap_mesh(d, mesh, yuv=True); ctx.sync()
            r = ref.remap_map(d, dm, yuv=True); torch.cuda.synchronize()
            print(f"mesh/map   {cols}x{rows}: hip(mesh) vs ref(map) {stats(g.cpu().numpy(), r.cpu().numpy())}", flush=True)
        # RCAS
        g = ctx.sharpen(d, 0.7); ctx.sync()
        pad = torch.zeros((rows + 8, cols + 8, 3), dtype=torch.uint8, device="cuda")
        rs = pad[:rows, :cols]
        ref.sharpen(d, 0.7, out=rs); torch.cuda.synchronize()
        print(f"rcas       {cols}x{rows}: hip vs ref {stats(g.cpu().numpy(), rs.cpu().numpy())}", flush=True)
    # upscale 1080p -> 4K
    src = synth.textured_frame(1080, 1920, seed=5); d = torch.from_numpy(src).cuda()
    for yuv in (True, False):
        g = ctx.upscale(d, (3840, 2160), yuv=yuv); ctx.sync()
        r = ref.upscale(d, (3840, 2160), yuv=yuv); torch.cuda.synchronize()
        print(f"upscale 1080p->4K yuv={yuv}: hip vs ref {stats(g.cpu().numpy(), r.cpu().numpy())}", flush=True)
    # timing at 4K.  The launches are marshalled ONCE (ref_cl's `prepared`): a loop that rebuilds the ctypes argument block per launch is
    # host-bound (~1.8 ms of Python per iteration) and two GPU events around it time Python, not the kernel -- that is how rounds 1-3 came to
    # quote 1 860 us for the reference's remap (it takes ~115 us; bench.py's `reference_kernel` leg and profiles/r04_kernel_stats.csv).
    src = synth.textured_frame(2160, 3840, seed=9); d = torch.from_numpy(src).cuda()
    H = synth.random_homography(2160, 3840, rng, strength=0.5)
    out_r = torch.zeros_like(d); out_g = torch.zeros_like(d)
    go_ref, _ = ref.remap_homography(d, H, yuv=True, out=out_r, prepared=True)
    t_ref = time_gpu(go_ref)
    t_hip = time_gpu(lambda: ctx.remap_homography(d, H, yuv=True, out=out_g))
    print(f"4K easu_remap_homography: reference OpenCL kernel {t_ref:.1f} us, k_remap_homography {t_hip:.1f} us (the latter incl. its Python call: see bench.py)")


if __name__ == "__main__":
    main()
