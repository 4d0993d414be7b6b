"""Generates tests/golden/*.npz: small input / expected-output vectors for every stage of the path and for a short stabilizer run.

The reference (C++ on OpenCV 4.8 + OpenCL + Eigen) cannot be built or run in this environment and ships no vectors of its own
(SURVEY.md section 8c), so these are produced by the CPU oracle (oracle/, a restatement of the reference) -- they do NOT pin the oracle
to the reference; they freeze the oracle and the HIP path against silent drift: tests/test_golden.py checks the oracle against
them on CPU (-m "not gpu") and the HIP path against them through the C-ABI (-m gpu).

    python tests/golden/make_golden.py          # rewrites the fixtures (only after a deliberate change of a stage's definition)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import oracle_lib, synth  # noqa: E402


def digest(a):
    a = np.ascontiguousarray(a)
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), np.uint8).copy()


def main():
    o = oracle_lib.load()
    rng = np.random.default_rng(0x4C564B31)
    out = {}

    # ---- a14-a16 remap
    src = synth.textured_frame(48, 64, seed=1)
    H = synth.random_homography(48, 64, rng, strength=2.0)
    mesh = synth.random_mesh(5, 7, rng, amp=0.03)
    mesh2 = synth.random_mesh(2, 2, rng, amp=0.03)
    out["remap"] = dict(src=src, H=H, mesh=mesh, mesh2=mesh2,
                        hom_yuv=o.remap_homography(src, H, bg=(3, 200, 77), yuv=True), hom_rgb=o.remap_homography(src, H, bg=(3, 200, 77), yuv=False),
                        mesh_yuv=o.remap_mesh(src, mesh, bg=(3, 200, 77), yuv=True), apply2=o.warpmesh_apply(src, mesh2, bg=(3, 200, 77), yuv=True))

    # ---- a3 / a4 / a7 image ops
    img = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    gray = synth.textured_frame(90, 130, seed=2)[..., 0].copy()
    out["imgproc"] = dict(img=img, gray=gray,
                          area_int=o.luma_area_resize(img, 24, 32), area_frac=o.luma_area_resize(img, 36, 50),
                          area_bgr=o.luma_area_resize(img, 24, 32, channel=-1), area_rgb=o.luma_area_resize(img, 24, 32, channel=-2),
                          pyr=o.pyr_down(gray), scharr=o.scharr_deriv(gray))

    # ---- a5 FAST
    fimg = synth.textured_frame(64, 96, seed=3)[..., 0].copy()
    kp_full = o.fast(fimg, 20)
    kp_roi = o.fast(fimg, 12, roi=(8, 4, 70, 50))
    out["fast"] = dict(img=fimg, kp_full=kp_full, kp_roi=kp_roi)

    # ---- a7 LK
    frames, _ = synth.make_clip(135, 240, 2, seed=4, jitter=1.5)
    prev, nxt = frames[0][..., 0].copy(), frames[1][..., 0].copy()
    kp = o.fast(prev, 20)
    pts = kp[:: max(1, len(kp) // 60), :2].astype(np.float32)[:60]
    pts = np.concatenate([pts, np.array([[0.5, 0.5], [239.0, 134.0], [-3.0, 10.0], [120.3, 67.7]], np.float32)])
    m, st = o.pyrlk(prev, nxt, pts)
    out["pyrlk"] = dict(prev=prev, nxt=nxt, pts=pts, matched=m, status=st)

    # ---- a8 / a9 motion
    n = 240
    Ht = np.array([[1.01, 0.012, 3.1], [-0.011, 0.995, -2.2], [2e-5, -1e-5, 1.0]])
    p1 = np.c_[rng.uniform(0, 480, n), rng.uniform(0, 270, n)].astype(np.float32)
    q = np.c_[p1, np.ones(n)] @ Ht.T
    p2 = (q[:, :2] / q[:, 2:] + rng.normal(0, 0.2, (n, 2)))
    bad = rng.random(n) < 0.3
    p2[bad] += rng.uniform(-40, 40, (bad.sum(), 2))
    p2 = p2.astype(np.float32)
    rc_h, Hh, mh = o.find_homography(p1, p2, 3.0)
    rc_a, Ha, ma = o.find_homography(p1, p2, 3.0, partial=True)
    out["motion"] = dict(p1=p1, p2=p2, rc_h=np.int32(rc_h), H_h=Hh, mask_h=mh, rc_a=np.int32(rc_a), H_a=Ha, mask_a=ma)

    # ---- a10 mesh solver
    ms = oracle_lib.OracleMeshSolver(o, 16, 16, gen_region=(480, 270))
    rc, inl, off = ms.solve(p1, p2, region=(480, 270), temporal=1.0, threshold=10.0)
    out["mesh"] = dict(p1=p1, p2=p2, rc=np.int32(rc), inliers=inl, offsets=off)
    ms.close()

    # ---- 8f-2 4:2:0
    pf = synth.textured_frame(32, 48, seed=5)
    y, u, v = o.egress_yuv420(pf)
    out["yuv420"] = dict(packed=pf, y=y, u=u, v=v, ingest=o.ingest_yuv420(y, u, v))

    # ---- 8f-1 lens
    params = np.array([0.8 * 131, 0.8 * 131, 131 / 2, 67 / 2, -0.12, 0.03, 1e-3, -2e-3, 0.01])
    lmap, view = o.lens_offset_map(params, 67, 131)
    lsrc = synth.textured_frame(67, 131, seed=6)
    lp = np.c_[rng.uniform(0, 60, 50), rng.uniform(0, 33, 50)].astype(np.float32)
    out["lens"] = dict(params=params, src=lsrc, map_sha=digest(lmap), map_samples=lmap[::11, ::13].copy(), view=np.array(view, np.int32),
                       remap_map=o.remap_map(lsrc, lmap, bg=(0, 0, 0)), fused=o.warpmesh_apply_lens(lsrc, mesh, params, bg=(9, 9, 9)),
                       pts=lp, undistorted=o.lens_undistort_points(params, 67, 131, 2.0, 2.0, lp))

    # ---- 8f-4 overlays
    dsrc = synth.textured_frame(40, 56, seed=7)
    dpts = np.c_[rng.uniform(-5, 60, 30), rng.uniform(-5, 45, 30)].astype(np.float32)
    out["draw"] = dict(src=dsrc, pts=dpts, grid=o.draw_grid(dsrc, (5, 3), (29, 255, 107), 1), crosses=o.draw_crosses(dsrc, dpts, (76, 84, 255), 7, 4, scaling=(1.0, 1.0)))

    # ---- 8f-4 second half: ScalingFilter's EASU upscale + RCAS
    ssrc = synth.textured_frame(40, 56, seed=9)
    ssrc[4:8, 4:20] = 0; ssrc[20:24, 30:50] = 255
    up = o.upscale(ssrc, (100, 61), yuv=True)
    out["scaling"] = dict(src=ssrc, up_yuv=up, up_rgb=o.upscale(ssrc, (112, 80), yuv=False), sharp=o.sharpen(up, 0.8), sharp_src=o.sharpen(ssrc, 0.35))

    # ---- a2 end to end: a short clip through both presets (outputs as digests, meshes and statistics in full)
    clip, _ = synth.make_clip(180, 320, 10, seed=8)
    e2e = dict(clip=clip)
    for name in ("homography", "field"):
        s = oracle_lib.preset(name, predictive_samples=3, detection_width=320, detection_height=180, min_scene_quality=0.4, min_tracking_quality=0.2)
        st = oracle_lib.OracleStabilizer(o, oracle_lib.preset("default")); st.configure(s)
        shas, meshes, stats = [], [], []
        for i, f in enumerate(clip):
            res, ts = st.push(f, ts=1000 + i)
            mo, co = st.meshes()
            ss = st.stats()
            stats.append([ss.n_detected, ss.n_matched, ss.n_tracked, ss.tracking_stability, ss.trust])
            meshes.append(mo.copy())
            shas.append(digest(res) if res is not None else np.zeros(32, np.uint8))
        st.close()
        e2e[name + "_sha"] = np.stack(shas); e2e[name + "_motion"] = np.stack(meshes); e2e[name + "_stats"] = np.array(stats, np.float64)
    out["stabilizer"] = e2e

    total = 0
    only = set(sys.argv[1:])                 # python make_golden.py scaling  -> rewrite just that fixture
    for k, d in out.items():
        if only and k not in only:
            continue
        path = os.path.join(HERE, k + ".npz")
        np.savez_compressed(path, **d)
        total += os.path.getsize(path)
        print(f"{k}.npz: {os.path.getsize(path)} bytes, {len(d)} arrays")
    print("total", total)


if __name__ == "__main__":
    main()
