"""Tolerance-mode A / B partner of the remap (livevisionkit_amd/variants/liblvk_hip_tolerant.so = scripts/variant_build.sh tolerant -DLVK_EASU_TOLERANT) against the
committed, bit-exact library: the same frames and warps through both, per-case maximum byte difference, share of differing bytes and PSNR (SURVEY 8c's
tolerance for row a16: <= 1 LSB per channel, PSNR >= 50 dB).

  python scripts/tolerant_probe.py run <tag>       outputs of the library LVK_HIP_LIB selects (default: the committed one) -> gpurun_out/tolerant/<tag>.npz
  python scripts/tolerant_probe.py compare a b     the two output sets against each other
"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
OUT = os.path.join(R, "gpurun_out", "tolerant")


def frames(rows, cols):
    """(name, packed frame): the texture classes an EASU kernel tells apart"""
    from tests import synth
    from tests.clipgen import Clip
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    out = [("textured", synth.textured_frame(rows, cols))]
    clip = Clip(rows, cols, 4, device="cpu")
    out.append(("clip_8d", clip.render444(2).numpy()))
    out.append(("noise", rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)))
    out.append(("flat_255", np.full((rows, cols, 3), 255, np.uint8)))
    out.append(("flat_17_128_240", np.broadcast_to(np.array([17, 128, 240], np.uint8), (rows, cols, 3)).copy()))
    ramp = np.stack([np.clip(xx * 255.0 / cols, 0, 255), np.clip(yy * 255.0 / rows, 0, 255), np.clip((xx + yy) * 127.0 / cols, 0, 255)], -1).astype(np.uint8)
    out.append(("ramps", ramp))
    low = (128 + 1.4 * np.sin(xx * 0.7) * np.cos(yy * 0.9) + rng.normal(0, 0.6, (rows, cols)))[..., None].repeat(3, -1)
    out.append(("low_contrast", np.clip(low, 0, 255).astype(np.uint8)))               # around the `zro` threshold of the direction estimate
    edges = np.where(((xx * 0.8 + yy * 0.6) % 37 < 18)[..., None], np.array([230, 60, 200], np.float32), np.array([20, 190, 40], np.float32))
    out.append(("diagonal_edges", edges.astype(np.uint8)))
    checker = np.where((((xx // 1) + (yy // 1)) % 2 == 0)[..., None], 255, 0).astype(np.uint8).repeat(3, -1)
    out.append(("checker_1px", checker))
    lines = np.where(((xx % 5) == 0)[..., None], 255, 0).astype(np.uint8).repeat(3, -1)
    out.append(("lines_5px", lines))
    return out


def run(tag):
    import torch
    import livevisionkit_amd as lvk
    from tests import synth
    rows, cols = 1080, 1920
    ctx = lvk.Context(0)
    rng = np.random.default_rng(3)
    Hs = [synth.random_homography(rows, cols, rng, strength=s) for s in (0.3, 1.0)]
    Hs.append(np.array([[0.95, 0, 0.025 * cols], [0, 0.95, 0.025 * rows], [0, 0, 1]], np.float32))                  # the 5 % crop of the bench stream
    mesh16 = synth.random_mesh(16, 16, rng, amp=0.01)
    mesh2 = np.array([[[0.004, -0.003], [-0.002, 0.004]], [[0.003, 0.002], [-0.004, -0.002]]], np.float32)
    res = {}
    for name, f in frames(rows, cols):
        s = torch.from_numpy(f).cuda()
        for i, H in enumerate(Hs):
            res[f"{name}/homography{i}/yuv"] = ctx.remap_homography(s, H, yuv=True).cpu().numpy()
        res[f"{name}/homography1/bgr"] = ctx.remap_homography(s, Hs[1], yuv=False).cpu().numpy()
        res[f"{name}/mesh16/yuv"] = ctx.remap_mesh(s, mesh16, yuv=True).cpu().numpy()
        y, u, v = ctx.warpmesh_apply_yuv420(s, mesh2)
        res[f"{name}/apply420/y"] = y.cpu().numpy(); res[f"{name}/apply420/u"] = u.cpu().numpy(); res[f"{name}/apply420/v"] = v.cpu().numpy()
    ctx.sync()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(f"{tag}: {len(res)} outputs, library {os.environ.get('LVK_HIP_LIB', '(committed)')}")


def compare(a, b):
    A = np.load(os.path.join(OUT, a + ".npz")); B = np.load(os.path.join(OUT, b + ".npz"))
    print(f"# {a} vs {b}: per case  max |diff|   share of bytes that differ   PSNR (dB)")
    worst, worst_psnr, tot_diff, tot = 0, 1e9, 0, 0
    for k in A.files:
        x, y = A[k].astype(np.int32), B[k].astype(np.int32)
        d = np.abs(x - y)
        mse = float((d.astype(np.float64) ** 2).mean())
        psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        worst = max(worst, int(d.max())); worst_psnr = min(worst_psnr, psnr); tot_diff += int((d != 0).sum()); tot += d.size
        print(f"{k:40s} {int(d.max()):3d}   {(d != 0).mean():9.6f}   {psnr:7.2f}" + ("" if d.max() <= 1 else f"   (> 1 LSB: {(d > 1).sum()} bytes)"))
    print(f"# overall: max |diff| {worst}, {tot_diff / tot:.6f} of all bytes differ, worst-case PSNR {worst_psnr:.2f} dB")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        compare(sys.argv[2], sys.argv[3])
