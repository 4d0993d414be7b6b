"""Micro-benchmark of the remap kernels (HIP events on the launch stream). Usage: python scripts/bench_remap.py [rows cols]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk  # noqa: E402
from tests import synth  # noqa: E402


def main():
    rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
    ctx = lvk.Context(0)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    # several distinct source frames so the working set exceeds what stays in L2
    srcs = [torch.randint(0, 256, (rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g) for _ in range(12)]
    out = torch.empty_like(srcs[0])
    rng = np.random.default_rng(0)
    H = synth.random_homography(rows, cols, rng, strength=0.5)
    mesh = synth.random_mesh(16, 16, rng, amp=0.01)
    # the fused remap + 4:2:0 egress kernel of the default stream (k_remap_homography_420): a 2 x 2 mesh of small corner offsets
    mesh2 = np.array([[[0.004, -0.003], [-0.002, 0.004]], [[0.003, 0.002], [-0.004, -0.002]]], np.float32)
    for name, fn in [("homography", lambda s: ctx.remap_homography(s, H, yuv=True, out=out)),
                     ("homography_420", lambda s: ctx.warpmesh_apply_yuv420(s, mesh2)),
                     ("mesh16", lambda s: ctx.remap_mesh(s, mesh, yuv=True, out=out))]:
        for s in srcs[:3]:
            fn(s)
        torch.cuda.synchronize()
        iters = 60
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(srcs[i % len(srcs)])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        byts = 6 * rows * cols
        print(f"{name}: {ms*1e3:.1f} us/frame  algorithmic {byts/1e6:.2f} MB -> {byts/ms/1e9*1e3/1e3:.3f} TB/s "
              f"({byts/(ms*1e-3)/8e12*100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
