"""Micro-benchmark of ScalingFilter's two kernels (HIP events on the launch stream): RCAS at rows x cols and EASU upscale from half
that size.  Usage: python scripts/bench_scaling.py [rows cols]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk  # noqa: E402


def timed(fn, srcs, iters=60):
    for s in srcs[:3]:
        fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(srcs[i % len(srcs)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
    ctx = lvk.Context(0)
    g = torch.Generator(device="cuda"); g.manual_seed(1)

    def smooth(r, c):      # natural-image-like content: random low-res field upsampled, plus noise
        low = torch.rand((1, 3, r // 16 + 2, c // 16 + 2), device="cuda", generator=g)
        img = torch.nn.functional.interpolate(low, size=(r, c), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
        img = img + 0.05 * torch.rand((r, c, 3), device="cuda", generator=g)
        return (img.clamp(0, 1) * 255).to(torch.uint8).contiguous()

    full = [smooth(rows, cols) for _ in range(10)]
    half = [smooth(rows // 2, cols // 2) for _ in range(10)]
    out = torch.empty_like(full[0])
    ms = timed(lambda s: ctx.sharpen(s, 0.8, out=out), full)
    byts = 6 * rows * cols
    print(f"rcas {cols}x{rows}: {ms*1e3:.1f} us  algorithmic {byts/1e6:.2f} MB -> {byts/(ms*1e-3)/1e9:.0f} GB/s ({byts/(ms*1e-3)/8e12*100:.1f}% of 8 TB/s)")
    ms = timed(lambda s: ctx.upscale(s, (cols, rows), yuv=True, out=out), half)
    byts = 3 * rows * cols + 3 * (rows // 2) * (cols // 2)
    print(f"easu_scale {cols//2}x{rows//2} -> {cols}x{rows}: {ms*1e3:.1f} us  algorithmic {byts/1e6:.2f} MB -> {byts/(ms*1e-3)/1e9:.0f} GB/s "
          f"({byts/(ms*1e-3)/8e12*100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
