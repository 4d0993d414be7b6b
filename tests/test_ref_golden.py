"""tests/golden/ref_kernels.npz holds OUTPUTS OF THE REFERENCE'S OWN KERNELS: FSR.cl / Drawing.cl compiled for gfx950 (oracle/_ref) and
run on an MI355X by tests/golden/make_ref_golden.py with the reference's host-side argument lists.  The CPU oracle must reproduce
them bit for bit (CPU-only suite: this is what pins the oracle of rows a16 / f4 to the reference), and so must the HIP kernels (GPU)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kernels.npz")


@pytest.fixture(scope="module")
def d():
    return np.load(G)


def _eq(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, what
    if not np.array_equal(a, b):
        diff = np.abs(a.astype(np.int32) - b.astype(np.int32))
        raise AssertionError(f"{what}: {int((diff.reshape(diff.shape[0], diff.shape[1], -1).max(axis=2) > 0).sum())} pixels differ, max |d| = {diff.max()}")


def _crop(ref_out, like):
    return ref_out[:like.shape[0], :like.shape[1]]


def _check(impl, d, to_dev=lambda a: a, to_host=lambda a: a):
    src, noise = to_dev(d["src"]), to_dev(d["noise"])
    for i, H in enumerate(d["H"]):
        _eq(to_host(impl.remap_homography(src, H, bg=(3, 200, 77), yuv=True)), d[f"hom_yuv_{i}"], f"easu_remap_homography yuv {i}")
        _eq(to_host(impl.remap_homography(src, H, bg=(3, 200, 77), yuv=False)), d[f"hom_bgr_{i}"], f"easu_remap_homography bgr {i}")
    _eq(to_host(impl.remap_homography(noise, d["H_noise"], bg=(0, 0, 0), yuv=True)), d["hom_noise"], "easu_remap_homography noise")
    _eq(to_host(impl.remap_homography(src, d["H"][1], bg=(1, 2, 3), dst_size=(40, 56), offset=(17, 9))), d["hom_roi"], "easu_remap_homography ROI")
    _eq(to_host(impl.remap_mesh(src, d["mesh"], bg=(9, 8, 7), yuv=True)), d["map_mesh"], "easu_remap 16x16")
    _eq(to_host(impl.remap_mesh(src, d["mesh_small"], bg=(9, 8, 7), yuv=True)), d["map_mesh_small"], "easu_remap 5x7")
    _eq(to_host(impl.remap_mesh(src, d["mesh"], bg=(9, 8, 7), yuv=False)), d["map_bgr"], "easu_remap bgr")
    _eq(to_host(impl.upscale(src, (233, 150), yuv=True)), d["up_yuv"], "easu_scale yuv")
    _eq(to_host(impl.upscale(src, (288, 192), yuv=False)), d["up_bgr"], "easu_scale bgr")
    _eq(to_host(impl.upscale(noise, (160, 111), yuv=True)), d["up_noise"], "easu_scale noise")
    for k, (name, s) in enumerate((("rcas_src", 1.0), ("rcas_src", 0.35), ("src", 0.8))):
        _eq(to_host(impl.sharpen(to_dev(d[name]), s)), _crop(d[f"rcas_{k}"], d[name]), f"rcas {k}")
    _eq(to_host(impl.draw_grid(to_dev(d["src"].copy()), (16, 16), (29, 255, 107), 1)), _crop(d["grid_16"], d["src"]), "grid 16x16")
    _eq(to_host(impl.draw_grid(to_dev(d["src"].copy()), (5, 3), (29, 255, 107), 2)), _crop(d["grid_5x3"], d["src"]), "grid 5x3")
    _eq(to_host(impl.draw_crosses(to_dev(d["src"].copy()), d["pts"], (76, 84, 255), 8, 4, scaling=(1.0, 1.0))), _crop(d["crosses"], d["src"]), "crosses")


def test_oracle_reproduces_the_reference_kernels(oracle, d):
    _check(oracle, d)


def test_oracle_without_the_device_reciprocal_is_within_one_lsb(oracle, d):
    """With native_recip = the correctly rounded 1/x instead of gfx950's v_rcp_f32 the result moves by at most 1 LSB in a handful of
    bytes (RCAS and the map / scale kernels) -- the size of what OpenCL leaves to the device."""
    oracle.set_device_rcp(False)
    try:
        for name, got in (("map_mesh", oracle.remap_mesh(d["src"], d["mesh"], bg=(9, 8, 7), yuv=True)), ("up_yuv", oracle.upscale(d["src"], (233, 150), yuv=True)),
                          ("rcas_2", oracle.sharpen(d["src"], 0.8))):
            ref = _crop(d[name], got)
            diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
            assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (name, int(diff.max()), float((diff > 0).mean()))
    finally:
        oracle.set_device_rcp(True)


@pytest.mark.gpu
def test_hip_reproduces_the_reference_kernels(ctx, d):
    import torch
    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    def host(t):
        ctx.sync()
        return t.cpu().numpy()
    _check(ctx, d, dev, host)
