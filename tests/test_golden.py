"""Committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py): the oracle is checked against them on
the CPU, the HIP path against them through the C-ABI on the GPU.  The vectors were produced by the oracle (the reference cannot
run here and has no vectors of its own), so they guard both implementations against silent drift -- a deliberate change of a
stage's definition means regenerating them in the same commit."""
import hashlib
import os

import numpy as np
import pytest

from tests import oracle_lib

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STAB_OVER = dict(predictive_samples=3, detection_width=320, detection_height=180, min_scene_quality=0.4, min_tracking_quality=0.2)


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


# ------------------------------------------------------------------------------------------------ oracle vs golden (CPU)
def test_oracle_remap_golden(oracle):
    d = load("remap")
    assert np.array_equal(oracle.remap_homography(d["src"], d["H"], bg=(3, 200, 77), yuv=True), d["hom_yuv"])
    assert np.array_equal(oracle.remap_homography(d["src"], d["H"], bg=(3, 200, 77), yuv=False), d["hom_rgb"])
    assert np.array_equal(oracle.remap_mesh(d["src"], d["mesh"], bg=(3, 200, 77), yuv=True), d["mesh_yuv"])
    assert np.array_equal(oracle.warpmesh_apply(d["src"], d["mesh2"], bg=(3, 200, 77), yuv=True), d["apply2"])


def test_oracle_imgproc_fast_lk_golden(oracle):
    d = load("imgproc")
    assert np.array_equal(oracle.luma_area_resize(d["img"], 24, 32), d["area_int"])
    assert np.array_equal(oracle.luma_area_resize(d["img"], 36, 50), d["area_frac"])
    assert np.array_equal(oracle.luma_area_resize(d["img"], 24, 32, channel=-1), d["area_bgr"])
    assert np.array_equal(oracle.luma_area_resize(d["img"], 24, 32, channel=-2), d["area_rgb"])
    assert np.array_equal(oracle.pyr_down(d["gray"]), d["pyr"]) and np.array_equal(oracle.scharr_deriv(d["gray"]), d["scharr"])
    f = load("fast")
    assert np.array_equal(oracle.fast(f["img"], 20), f["kp_full"]) and np.array_equal(oracle.fast(f["img"], 12, roi=(8, 4, 70, 50)), f["kp_roi"])
    k = load("pyrlk")
    m, st = oracle.pyrlk(k["prev"], k["nxt"], k["pts"])
    assert np.array_equal(st, k["status"]) and np.array_equal(m.view(np.uint32), k["matched"].view(np.uint32))


def test_oracle_motion_mesh_golden(oracle):
    d = load("motion")
    rc, H, mask = oracle.find_homography(d["p1"], d["p2"], 3.0)
    assert rc == int(d["rc_h"]) and np.array_equal(H.view(np.uint64), d["H_h"].view(np.uint64)) and np.array_equal(mask, d["mask_h"])
    rc, H, mask = oracle.find_homography(d["p1"], d["p2"], 3.0, partial=True)
    assert rc == int(d["rc_a"]) and np.array_equal(H.view(np.uint64), d["H_a"].view(np.uint64)) and np.array_equal(mask, d["mask_a"])
    m = load("mesh")
    ms = oracle_lib.OracleMeshSolver(oracle, 16, 16, gen_region=(480, 270))
    rc, inl, off = ms.solve(m["p1"], m["p2"], region=(480, 270), temporal=1.0, threshold=10.0)
    ms.close()
    assert rc == int(m["rc"]) and np.array_equal(inl, m["inliers"]) and np.array_equal(off.view(np.uint32), m["offsets"].view(np.uint32))


def test_oracle_yuv420_lens_draw_golden(oracle):
    d = load("yuv420")
    y, u, v = oracle.egress_yuv420(d["packed"])
    assert np.array_equal(y, d["y"]) and np.array_equal(u, d["u"]) and np.array_equal(v, d["v"])
    assert np.array_equal(oracle.ingest_yuv420(d["y"], d["u"], d["v"]), d["ingest"])
    l = load("lens")
    lmap, view = oracle.lens_offset_map(l["params"], 67, 131)
    assert np.array_equal(sha(lmap), l["map_sha"]) and list(view) == list(l["view"])
    assert np.array_equal(oracle.remap_map(l["src"], lmap, bg=(0, 0, 0)), l["remap_map"])
    assert np.array_equal(oracle.warpmesh_apply_lens(l["src"], load("remap")["mesh"], l["params"], bg=(9, 9, 9)), l["fused"])
    assert np.array_equal(oracle.lens_undistort_points(l["params"], 67, 131, 2.0, 2.0, l["pts"]).view(np.uint32), l["undistorted"].view(np.uint32))
    w = load("draw")
    assert np.array_equal(oracle.draw_grid(w["src"], (5, 3), (29, 255, 107), 1), w["grid"])
    assert np.array_equal(oracle.draw_crosses(w["src"], w["pts"], (76, 84, 255), 7, 4), w["crosses"])


def test_oracle_scaling_golden(oracle):
    d = load("scaling")
    assert np.array_equal(oracle.upscale(d["src"], (100, 61), yuv=True), d["up_yuv"])
    assert np.array_equal(oracle.upscale(d["src"], (112, 80), yuv=False), d["up_rgb"])
    assert np.array_equal(oracle.sharpen(d["up_yuv"], 0.8), d["sharp"]) and np.array_equal(oracle.sharpen(d["src"], 0.35), d["sharp_src"])


@pytest.mark.parametrize("name", ["homography", "field"])
def test_oracle_stabilizer_golden(oracle, name):
    d = load("stabilizer")
    st = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); st.configure(oracle_lib.preset(name, **STAB_OVER))
    for i, f in enumerate(d["clip"]):
        res, _ = st.push(f, ts=1000 + i)
        ss = st.stats()
        assert np.array_equal(np.array([ss.n_detected, ss.n_matched, ss.n_tracked, ss.tracking_stability, ss.trust], np.float64), d[name + "_stats"][i]), i
        assert np.array_equal(st.meshes()[0].view(np.uint32), d[name + "_motion"][i].view(np.uint32)), i
        assert np.array_equal(sha(res) if res is not None else np.zeros(32, np.uint8), d[name + "_sha"][i]), i
    st.close()


# ------------------------------------------------------------------------------------------------ HIP path vs golden (GPU)
def _gpu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_hip_stage_golden(ctx):
    d = load("remap")
    src = _gpu(d["src"])
    assert np.array_equal(ctx.remap_homography(src, d["H"], bg=(3, 200, 77), yuv=True).cpu().numpy(), d["hom_yuv"])
    assert np.array_equal(ctx.remap_homography(src, d["H"], bg=(3, 200, 77), yuv=False).cpu().numpy(), d["hom_rgb"])
    assert np.array_equal(ctx.remap_mesh(src, d["mesh"], bg=(3, 200, 77), yuv=True).cpu().numpy(), d["mesh_yuv"])
    assert np.array_equal(ctx.warpmesh_apply(src, d["mesh2"], bg=(3, 200, 77), yuv=True).cpu().numpy(), d["apply2"])
    i = load("imgproc")
    img, gray = _gpu(i["img"]), _gpu(i["gray"])
    assert np.array_equal(ctx.luma_area_resize(img, 24, 32).cpu().numpy(), i["area_int"])
    assert np.array_equal(ctx.luma_area_resize(img, 36, 50).cpu().numpy(), i["area_frac"])
    assert np.array_equal(ctx.luma_area_resize(img, 24, 32, channel=-1).cpu().numpy(), i["area_bgr"])
    assert np.array_equal(ctx.luma_area_resize(img, 24, 32, channel=-2).cpu().numpy(), i["area_rgb"])
    assert np.array_equal(ctx.pyr_down(gray).cpu().numpy(), i["pyr"]) and np.array_equal(ctx.scharr(gray).cpu().numpy(), i["scharr"])
    f = load("fast")
    res, counts = ctx.fast_detect(_gpu(f["img"]), [(0, 0, 96, 64, 20, 1), (8, 4, 70, 50, 12, 1)])
    assert np.array_equal(res[0], f["kp_full"]) and np.array_equal(res[1], f["kp_roi"])
    k = load("pyrlk")
    m, st = ctx.pyrlk(_gpu(k["prev"]), _gpu(k["nxt"]), k["pts"])
    assert np.array_equal(st, k["status"]) and np.array_equal(m.view(np.uint32), k["matched"].view(np.uint32))
    g = load("motion")
    rc, H, mask = ctx.estimate_global_motion(g["p1"], g["p2"], 3.0, full_homography=True)
    assert rc == int(g["rc_h"]) and np.array_equal(H.reshape(-1).view(np.uint64), g["H_h"].reshape(-1).view(np.uint64)) and np.array_equal(mask, g["mask_h"])
    rc, H, mask = ctx.estimate_global_motion(g["p1"], g["p2"], 3.0, full_homography=False)
    assert rc == int(g["rc_a"]) and np.array_equal(H.reshape(-1).view(np.uint64), g["H_a"].reshape(-1).view(np.uint64)) and np.array_equal(mask, g["mask_a"])
    y = load("yuv420")
    planes = ctx.egress_yuv420(_gpu(y["packed"]))
    assert all(np.array_equal(a.cpu().numpy(), y[n]) for a, n in zip(planes, ("y", "u", "v")))
    assert np.array_equal(ctx.ingest_yuv420(_gpu(y["y"]), _gpu(y["u"]), _gpu(y["v"])).cpu().numpy(), y["ingest"])
    l = load("lens")
    lmap, view = ctx.lens_map(l["params"], 67, 131)
    assert np.array_equal(sha(lmap.cpu().numpy()), l["map_sha"]) and list(view) == list(l["view"])
    assert np.array_equal(ctx.remap_map(_gpu(l["src"]), lmap, bg=(0, 0, 0)).cpu().numpy(), l["remap_map"])
    assert np.array_equal(ctx.warpmesh_apply_lens(_gpu(l["src"]), d["mesh"], l["params"], bg=(9, 9, 9)).cpu().numpy(), l["fused"])
    assert np.array_equal(ctx.lens_undistort_points(l["params"], 67, 131, 2.0, 2.0, l["pts"]).view(np.uint32), l["undistorted"].view(np.uint32))
    w = load("draw")
    assert np.array_equal(ctx.draw_grid(_gpu(w["src"]), (5, 3), (29, 255, 107), 1).cpu().numpy(), w["grid"])
    assert np.array_equal(ctx.draw_crosses(_gpu(w["src"]), w["pts"], (76, 84, 255), 7, 4).cpu().numpy(), w["crosses"])
    z = load("scaling")
    assert np.array_equal(ctx.upscale(_gpu(z["src"]), (100, 61), yuv=True).cpu().numpy(), z["up_yuv"])
    assert np.array_equal(ctx.upscale(_gpu(z["src"]), (112, 80), yuv=False).cpu().numpy(), z["up_rgb"])
    assert np.array_equal(ctx.sharpen(_gpu(z["up_yuv"]), 0.8).cpu().numpy(), z["sharp"])
    assert np.array_equal(ctx.sharpen(_gpu(z["src"]), 0.35).cpu().numpy(), z["sharp_src"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["homography", "field"])
def test_hip_stabilizer_golden(ctx, name):
    import ctypes
    import livevisionkit_amd as lvk

    def conv(o):
        s = lvk.StabilizationFilterSettings()
        ctypes.memmove(ctypes.byref(s), ctypes.byref(o), ctypes.sizeof(o))
        return s
    d = load("stabilizer")
    st = lvk.StabilizationFilter(conv(oracle_lib.preset("default")), context=ctx)
    st.configure(conv(oracle_lib.preset(name, **STAB_OVER)))
    for i, f in enumerate(d["clip"]):
        res, _ = st.apply(_gpu(f), timestamp=1000 + i)
        ctx.sync()
        ss = st.stats()
        assert np.array_equal(np.array([ss.n_detected, ss.n_matched, ss.n_tracked, ss.tracking_stability, ss.trust], np.float64), d[name + "_stats"][i]), i
        assert np.array_equal(st.meshes()[0].view(np.uint32), d[name + "_motion"][i].view(np.uint32)), i
        assert np.array_equal(sha(res.cpu().numpy()) if res is not None else np.zeros(32, np.uint8), d[name + "_sha"][i]), i
    st.close()
