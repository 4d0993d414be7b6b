"""Generates tests/golden/ref_kernels.npz ON THE GPU BOX: outputs of the REFERENCE'S OWN OpenCL kernels (oracle/_ref/*.hsaco =
LiveVisionKit/Functions/OpenCL/Sources/{FSR,Drawing}.cl compiled for gfx950 by `make -C oracle ref`) on small seeded inputs, launched with
the reference's host-side argument lists (tests/ref_cl.py).  These are reference outputs, not oracle outputs: tests/test_ref_golden.py
checks the CPU oracle against them bit for bit in the CPU-only suite, and the HIP kernels against them on the GPU.

    gpurun -- 'python tests/golden/make_ref_golden.py gpurun_out/ref_kernels.npz'   then copy the file to tests/golden/
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import oracle_lib, ref_cl, synth  # noqa: E402


def inputs():
    """Seeded inputs shared by the generator and the tests (the .npz stores them too, so the tests do not depend on this code)."""
    rng = np.random.default_rng(0x52454631)
    d = {}
    d["src"] = synth.textured_frame(96, 144, seed=41)
    d["noise"] = rng.integers(0, 256, (72, 104, 3), dtype=np.uint8)
    d["H"] = np.stack([synth.random_homography(96, 144, rng, strength=s) for s in (0.5, 1.5, 3.0)])
    d["H_noise"] = synth.random_homography(72, 104, rng, strength=2.0)
    d["mesh"] = synth.random_mesh(16, 16, rng, amp=0.015)
    d["mesh_small"] = synth.random_mesh(5, 7, rng, amp=0.03)
    flat = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    flat[8:24, 8:40] = 0; flat[30:44, 8:40] = 255
    d["rcas_src"] = flat
    d["pts"] = np.concatenate([rng.uniform(-5, 150, (60, 2)), np.array([[0, 0], [143.5, 95.5], [1.5, 2.5], [2.5, 3.5]])]).astype(np.float32)
    return d


def padded(a, pad=16):
    rows, cols = a.shape[:2]
    big = torch.zeros((rows + pad, cols + pad) + tuple(a.shape[2:]), dtype=torch.uint8, device="cuda")
    v = big[:rows, :cols]
    v.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return v


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_kernels.npz")
    ref = ref_cl.RefKernels()
    o = oracle_lib.load()
    d = inputs()
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    src, noise = g(d["src"]), g(d["noise"])
    for i, H in enumerate(d["H"]):
        d[f"hom_yuv_{i}"] = ref.remap_homography(src, H, bg=(3, 200, 77), yuv=True).cpu().numpy()
        d[f"hom_bgr_{i}"] = ref.remap_homography(src, H, bg=(3, 200, 77), yuv=False).cpu().numpy()
    d["hom_noise"] = ref.remap_homography(noise, d["H_noise"], bg=(0, 0, 0), yuv=True).cpu().numpy()
    d["hom_roi"] = ref.remap_homography(src, d["H"][1], bg=(1, 2, 3), dst_size=(40, 56), offset=(17, 9)).cpu().numpy()
    # the W x H map of WarpMesh::apply (cv::resize of the mesh, restated by the oracle -- OpenCV is not in the image) through easu_remap
    for name in ("mesh", "mesh_small"):
        m = o.mesh_to_map(d[name], 96, 144)
        d["map_" + name] = ref.remap_map(src, g(m), bg=(9, 8, 7), yuv=True).cpu().numpy()
    d["map_bgr"] = ref.remap_map(src, g(o.mesh_to_map(d["mesh"], 96, 144)), bg=(9, 8, 7), yuv=False).cpu().numpy()
    d["up_yuv"] = ref.upscale(src, (233, 150), yuv=True).cpu().numpy()
    d["up_bgr"] = ref.upscale(src, (288, 192), yuv=False).cpu().numpy()
    d["up_noise"] = ref.upscale(noise, (160, 111), yuv=True).cpu().numpy()
    for k, (name, s) in enumerate((("rcas_src", 1.0), ("rcas_src", 0.35), ("src", 0.8))):
        outp = padded(np.zeros_like(d[name]))
        ref.sharpen(padded(d[name]), s, out=outp)
        d[f"rcas_{k}"] = outp.cpu().numpy()
    gr = padded(d["src"]); ref.draw_grid(gr, np.float32(144) / np.float32(16), np.float32(96) / np.float32(16), 1, (29, 255, 107))
    d["grid_16"] = gr.cpu().numpy()
    gr = padded(d["src"]); ref.draw_grid(gr, np.float32(144) / np.float32(5), np.float32(96) / np.float32(3), 2, (29, 255, 107))
    d["grid_5x3"] = gr.cpu().numpy()
    pi = np.stack([np.rint(d["pts"][:, 0]), np.rint(d["pts"][:, 1])], axis=1).astype(np.int32)      # scaling (1, 1)
    cr = padded(d["src"], pad=64); ref.draw_crosses(cr, g(pi), 8, 4, (76, 84, 255))
    d["crosses"] = cr.cpu().numpy()
    torch.cuda.synchronize()
    d["device"] = np.array(torch.cuda.get_device_name(0))
    np.savez_compressed(out_path, **d)
    print("wrote", out_path, os.path.getsize(out_path), "bytes,", len(d), "arrays")


if __name__ == "__main__":
    main()
