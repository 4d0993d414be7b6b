"""Companion of opencv_ref_dump.cpp: `export DIR` writes the inputs of every OpenCV call the reference makes on the hot path; `compare DIR`
holds the CPU oracle (tests/oracle_lib.py, hence -- through the GPU suite -- the HIP kernels) against what OpenCV 4.8 produced from them.

Nothing in the product, the tests or the bench depends on this tool, and it is not run here: the image has no OpenCV.
It is the hook that lets someone with OpenCV 4.8.0 pin the rows whose arithmetic the reference delegates to that library:

    python scripts/opencv_ref/opencv_ref_compare.py export /tmp/lvk_cv
    g++ -O2 -std=c++17 scripts/opencv_ref/opencv_ref_dump.cpp -o /tmp/opencv_ref_dump $(pkg-config --cflags --libs opencv4)
    /tmp/opencv_ref_dump /tmp/lvk_cv
    python scripts/opencv_ref/opencv_ref_compare.py compare /tmp/lvk_cv

Bars: a3 / a4 exact (+-1 LSB at non-integer scales); a5 exact set equality of (x, y, score) in row-major order; a7
|d| <= 0.01 px for >= 99.9 % of the points and <= 0.1 % status flips; a9 reported, not asserted bit for bit (the product's estimator is its
own: corner displacement of H against OpenCV's USAC, p95 <= 0.25 px, inlier flags >= 99 % equal); a12 <= 1e-7; a14 matrix 1e-9 relative,
mesh map exact; the 4:2:0 chroma planes exact; a10 (needs Eigen 3.4 as well, -DLVK_WITH_EIGEN): the mesh Eigen's LeastSquaresConjugateGradient::
solveWithGuess stops at, on the exporter's five warm-started systems of the vector-field preset, within 1e-5 (normalised offsets) of the
oracle's least-squares solution.  `scripts/opencv_ref/run.sh` does the four steps in one command."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
DT = {np.dtype(np.uint8): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}
RDT = {v: k for k, v in DT.items()}


def save(path, a):
    a = np.ascontiguousarray(a)
    with open(path, "wb") as f:
        f.write(b"LVKA" + struct.pack("<ii", DT[a.dtype], a.ndim) + struct.pack("<%di" % a.ndim, *a.shape))
        f.write(a.tobytes())


def load(path):
    with open(path, "rb") as f:
        assert f.read(4) == b"LVKA", path
        dt, nd = struct.unpack("<ii", f.read(8))
        shape = struct.unpack("<%di" % nd, f.read(4 * nd))
        return np.frombuffer(f.read(), RDT[dt]).reshape(shape).copy()


def clip_pairs():
    """Tracking-frame pairs and the tracker's own point sets from SURVEY 8d's clip (960 x 540 render, tracking at 480 x 270)."""
    from tests import clipgen, oracle_lib
    oracle = oracle_lib.load()
    clip = clipgen.Clip(540, 960, 120, cut_at=None)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("homography", predictive_samples=1))
    frames, sets = [], []
    for i in range(60):
        f = clip.render444(i).numpy()
        ost.push(f, ts=i)
        frames.append(oracle.luma_area_resize(f, 270, 480))
        p1, p2, est = ost.matches()
        if est in (1, 2) and i % 6 == 5:
            sets.append((p1, p2))
    ost.close()
    return oracle, clip, frames, sets


def export(d):
    os.makedirs(d, exist_ok=True)
    from tests import synth
    oracle, clip, frames, sets = clip_pairs()
    rng = np.random.default_rng(3)
    # a4: integer box (8 x 8, 4 x 4), fractional (720p, 1440p), enlarging
    for k, (r, c) in enumerate([(2160, 3840), (1080, 1920), (720, 1280), (1440, 2560), (180, 320)]):
        save(f"{d}/area_{k}.src", synth.textured_frame(r, c, seed=40 + k, channels=1))
        save(f"{d}/area_{k}.size", np.array([480, 270], np.int32))
    save(f"{d}/gray.src", synth.textured_frame(360, 640, seed=9))
    # a5: whole frame and the two regions of the homography preset, three thresholds
    k = 0
    for img in (frames[10], frames[40]):
        for (x, y, w, h) in ((0, 0, 480, 270), (0, 0, 240, 270), (240, 0, 240, 270)):
            for thr in (10, 35, 70):
                save(f"{d}/fast_{k}.img", img); save(f"{d}/fast_{k}.roi_thr", np.array([x, y, w, h, thr], np.int32)); k += 1
    # a7: consecutive tracking frames, FAST corners + random points (borders, flat areas)
    for k, i in enumerate(range(5, 60, 6)):
        kp = oracle.fast(frames[i - 1], 15)[:, :2].astype(np.float32)
        pts = np.concatenate([kp[:1200], rng.uniform([-4, -4], [484, 274], (64, 2)).astype(np.float32)])
        save(f"{d}/lk_{k}.prev", frames[i - 1]); save(f"{d}/lk_{k}.next", frames[i]); save(f"{d}/lk_{k}.pts", pts)
    # a9: the tracker's match sets
    for k, (p1, p2) in enumerate(sets):
        save(f"{d}/motion_{k}.p1", p1.astype(np.float32)); save(f"{d}/motion_{k}.p2", p2.astype(np.float32)); save(f"{d}/motion_{k}.thr", np.array([3.0], np.float32))
    save(f"{d}/gauss.n_sigma", np.array([[21, 21 / 12.0], [21, 21 / 12.0 + 7.3], [11, 11 / 12.0 + 20.0], [121, 121 / 12.0 + 3.0]], np.float64))
    quads = []
    for _ in range(8):
        dst = np.array([0, 0, 3840, 0, 0, 2160, 3840, 2160], np.float32)
        quads.append(np.concatenate([dst, dst + rng.uniform(-60, 60, 8).astype(np.float32)]))
    save(f"{d}/persp.quads", np.array(quads, np.float32))
    save(f"{d}/meshmap.mesh", synth.random_mesh(16, 16, rng, amp=0.01).astype(np.float32)); save(f"{d}/meshmap.size", np.array([1280, 720], np.int32))
    save(f"{d}/chroma.plane", synth.textured_frame(270, 480, seed=77, channels=1))
    # a10: the sparse least-squares systems of FrameTracker::estimate_local_motions (Vision/FrameTracker.cpp:219-268) on five warm-started frames
    # of the vector-field preset (16 x 16 mesh over 480 x 270), as triplets + right-hand side + the guess (the oracle's previous mesh), for
    # Eigen::LeastSquaresConjugateGradient::solveWithGuess (:274-276)
    for k, (A, b, x0, _) in enumerate(lscg_systems(oracle)):
        r, c = np.nonzero(A)
        save(f"{d}/lscg_{k}.rows", r.astype(np.int32)); save(f"{d}/lscg_{k}.cols", c.astype(np.int32)); save(f"{d}/lscg_{k}.vals", A[r, c].astype(np.float32))
        save(f"{d}/lscg_{k}.b", b.astype(np.float32)); save(f"{d}/lscg_{k}.x0", x0.astype(np.float32)); save(f"{d}/lscg_{k}.shape", np.array(A.shape, np.int32))
    print("inputs written to", d)


LSCG_MESH, LSCG_REGION = (16, 16), (480, 270)


def lscg_systems(oracle):
    """[(A, b, x0, oracle offsets)] of five consecutive frames: every system is built around the ORACLE's previous mesh (temporal rows and
    guess), so each one stands alone -- a real Eigen solves it from the same state the specification solved it from."""
    from tests import oracle_lib, test_mesh_lstsq as ml
    cols, rows = LSCG_MESH
    rng = np.random.default_rng(cols * 31 + rows)
    static = ml.static_rows(cols, rows, LSCG_REGION, 1.0, 20.0)
    ref = oracle_lib.OracleMeshSolver(oracle, cols, rows, gen_region=LSCG_REGION, temporal=1.0, local=20.0)
    out = []
    for frame in range(5):
        a, b2 = ml.field_pairs(rng, 700 - 50 * frame, LSCG_REGION, frame)
        x0 = ref.mesh().reshape(-1).astype(np.float32)                      # m_OptimizedMesh before this frame's solve
        A, b, feats = ml.build_system(cols, rows, static, LSCG_REGION, 1.0, x0, a, b2)
        rc, _, off = ref.solve(a, b2, region=LSCG_REGION, temporal=1.0, threshold=10.0)
        assert rc == 0
        out.append((A, b, x0, off.astype(np.float64).reshape(rows, cols, 2)))
    ref.close()
    return out


def lscg_offsets(x):
    """normalised offsets (FrameTracker.cpp:316-320) of a solved mesh vector"""
    from tests import test_mesh_lstsq as ml
    cols, rows = LSCG_MESH
    kw, kh = ml.key_size(cols, rows, LSCG_REGION)
    return ml.mesh_to_result(cols, rows, LSCG_REGION, kw, kh, np.asarray(x, np.float32), [], [], 10.0)[1]


def compare(d):
    from tests import np_smoother, oracle_lib
    oracle = oracle_lib.load()
    bad = []

    def report(name, ok, text):
        print(("ok   " if ok else "FAIL ") + name + ": " + text)
        if not ok:
            bad.append(name)
    k = 0
    while os.path.exists(f"{d}/area_{k}.out"):
        src, (w, h), cv = load(f"{d}/area_{k}.src"), load(f"{d}/area_{k}.size"), load(f"{d}/area_{k}.out")
        got = oracle.luma_area_resize(src, int(h), int(w))
        diff = np.abs(got.astype(int) - cv.astype(int))
        integer = src.shape[0] % h == 0 and src.shape[1] % w == 0
        report(f"a4 INTER_AREA {src.shape[1]}x{src.shape[0]}", diff.max() <= (0 if integer else 1), f"max |d| {diff.max()}, {100 * (diff > 0).mean():.4f} % of the pixels differ")
        k += 1
    if os.path.exists(f"{d}/gray.out"):
        bgr, cv = load(f"{d}/gray.src"), load(f"{d}/gray.out")
        got = oracle.luma_area_resize(bgr, bgr.shape[0], bgr.shape[1], channel=-1)       # channel -1: BGR -> gray, then the (identity) box
        report("a3 cvtColor(BGR2GRAY)", np.array_equal(got, cv), f"max |d| {np.abs(got.astype(int) - cv.astype(int)).max()}")
    k = 0
    while os.path.exists(f"{d}/fast_{k}.out"):
        img, (x, y, w, h, thr), cv = load(f"{d}/fast_{k}.img"), load(f"{d}/fast_{k}.roi_thr"), load(f"{d}/fast_{k}.out")
        got = oracle.fast(img, int(thr), (int(x), int(y), int(w), int(h))).astype(np.float32)
        report(f"a5 FAST roi {(x, y, w, h)} thr {thr}", got.shape == cv.shape and np.array_equal(got, cv), f"{len(cv)} keypoints (OpenCV) / {len(got)} (oracle)")
        k += 1
    k = 0
    allp, alls = [], []
    while os.path.exists(f"{d}/lk_{k}.out_pts"):
        got_p, got_s = oracle.pyrlk(load(f"{d}/lk_{k}.prev"), load(f"{d}/lk_{k}.next"), load(f"{d}/lk_{k}.pts"))
        cv_p, cv_s = load(f"{d}/lk_{k}.out_pts"), load(f"{d}/lk_{k}.out_status")
        both = (got_s == 1) & (cv_s == 1)
        allp.append(np.abs(got_p[both] - cv_p[both]).max(axis=1)); alls.append(got_s != cv_s)
        k += 1
    if allp:
        dd, ff = np.concatenate(allp), np.concatenate(alls)
        report("a7 PyrLK", (dd <= 0.01).mean() >= 0.999 and ff.mean() <= 0.001,
               f"{len(ff)} points: |d| <= 0.01 px for {100 * (dd <= 0.01).mean():.3f} %, p99.9 {np.percentile(dd, 99.9):.5f} px, status flips {100 * ff.mean():.4f} %")
    k = 0
    corners = np.array([[0, 0, 1], [480, 0, 1], [0, 270, 1], [480, 270, 1]], np.float64)
    disp, same = [], []
    while os.path.exists(f"{d}/motion_{k}.out_H"):
        p1, p2, thr = load(f"{d}/motion_{k}.p1"), load(f"{d}/motion_{k}.p2"), float(load(f"{d}/motion_{k}.thr")[0])
        rc, H, mask = oracle.find_homography(p1, p2, thr)
        Hcv, mcv = load(f"{d}/motion_{k}.out_H"), load(f"{d}/motion_{k}.out_mask")
        if abs(Hcv[2, 2]) > 0:
            a = corners @ H.T; b = corners @ Hcv.T
            disp.append(np.linalg.norm(a[:, :2] / a[:, 2:] - b[:, :2] / b[:, 2:], axis=1).max()); same.append((mask == mcv).mean())
        k += 1
    if disp:
        report("a9 findHomography(USAC) vs the product's estimator", np.percentile(disp, 95) <= 0.25 and np.mean(same) >= 0.99,
               f"{len(disp)} sets: corner displacement p50 {np.median(disp):.3f} / p95 {np.percentile(disp, 95):.3f} / max {max(disp):.3f} px, inlier flags equal {100 * np.mean(same):.2f} %")
    if os.path.exists(f"{d}/gauss_0.out"):
        worst = 0.0
        for k, (n, sigma) in enumerate(load(f"{d}/gauss.n_sigma")):
            worst = max(worst, float(np.abs(np_smoother.gaussian_kernel_f32(int(n), float(sigma)) - load(f"{d}/gauss_{k}.out")).max()))
        report("a12 getGaussianKernel", worst <= 1e-7, f"max |d| {worst:.2e}")
    if os.path.exists(f"{d}/persp.out"):
        q, cv = load(f"{d}/persp.quads"), load(f"{d}/persp.out")
        worst = 0.0
        for k in range(len(q)):
            rc, M = oracle.get_perspective_transform(q[k, :8], q[k, 8:])
            worst = max(worst, float(np.abs(M.reshape(-1) - cv[k]).max() / np.abs(cv[k]).max()))
        report("a14 getPerspectiveTransform", worst <= 1e-9, f"max relative |d| {worst:.2e}")
    if os.path.exists(f"{d}/meshmap.out"):
        mesh, (w, h), cv = load(f"{d}/meshmap.mesh"), load(f"{d}/meshmap.size"), load(f"{d}/meshmap.out")
        got = oracle.mesh_to_map(mesh, int(h), int(w))
        # lvko_mesh_to_map returns the map of WarpMesh::apply (offsets * (cols, rows)); cv::resize leaves them normalised
        got = got / np.array([w, h], np.float32)
        report("a14 mesh -> map (cv::resize f32)", np.abs(got - cv).max() <= 1e-7, f"max |d| {np.abs(got - cv).max():.2e} (normalised offsets)")
    if os.path.exists(f"{d}/chroma.out_up"):
        plane = load(f"{d}/chroma.plane")
        y = np.zeros((plane.shape[0] * 2, plane.shape[1] * 2), np.uint8)
        packed = oracle.ingest_yuv420(y, plane, plane)
        report("f2 chroma INTER_LINEAR x2", np.array_equal(packed[..., 1], load(f"{d}/chroma.out_up")), "bilinear upsampling of a plane")
        yy, u, v = oracle.egress_yuv420(packed)
        report("f2 chroma INTER_AREA 0.5", np.array_equal(u, load(f"{d}/chroma.out_down")), "2 x 2 box")
    if os.path.exists(f"{d}/lscg_0.out"):
        worst, iters = 0.0, []
        for k, (_, _, _, off_oracle) in enumerate(lscg_systems(oracle)):
            if not os.path.exists(f"{d}/lscg_{k}.out"):
                break
            worst = max(worst, float(np.abs(lscg_offsets(load(f"{d}/lscg_{k}.out")) - off_oracle).max()))
            if os.path.exists(f"{d}/lscg_{k}.out_iters"):
                iters.append(int(load(f"{d}/lscg_{k}.out_iters")[0]))
        report("a10 Eigen LSCG solveWithGuess vs the oracle's least-squares mesh", worst <= 1e-5,
               f"max |offset difference| {worst:.2e} (normalised; {worst * LSCG_REGION[0]:.4f} px), iterations {iters}")
    else:
        print("--   a10 Eigen LSCG: no lscg_*.out (the dump tool was built without -DLVK_WITH_EIGEN)")
    print("%d stage(s) outside their bar" % len(bad) if bad else "all stages within their bars")
    return 1 if bad else 0


def selftest_outputs(d):
    """tests/test_opencv_hook.py only: fills the `.out` files with the ORACLE's own results (standing in for the dump tool, which cannot be
    built here) so that the comparer's reading, indexing and bars are exercised.  Proves nothing about OpenCV."""
    from tests import np_smoother, oracle_lib
    oracle = oracle_lib.load()
    k = 0
    while os.path.exists(f"{d}/area_{k}.src"):
        w, h = load(f"{d}/area_{k}.size"); save(f"{d}/area_{k}.out", oracle.luma_area_resize(load(f"{d}/area_{k}.src"), int(h), int(w))); k += 1
    bgr = load(f"{d}/gray.src"); save(f"{d}/gray.out", oracle.luma_area_resize(bgr, bgr.shape[0], bgr.shape[1], channel=-1))
    k = 0
    while os.path.exists(f"{d}/fast_{k}.img"):
        x, y, w, h, thr = load(f"{d}/fast_{k}.roi_thr")
        save(f"{d}/fast_{k}.out", oracle.fast(load(f"{d}/fast_{k}.img"), int(thr), (int(x), int(y), int(w), int(h))).astype(np.float32)); k += 1
    k = 0
    while os.path.exists(f"{d}/lk_{k}.prev"):
        p, s = oracle.pyrlk_float(load(f"{d}/lk_{k}.prev"), load(f"{d}/lk_{k}.next"), load(f"{d}/lk_{k}.pts"), 4, 1)
        save(f"{d}/lk_{k}.out_pts", p); save(f"{d}/lk_{k}.out_status", s); k += 1
    k = 0
    while os.path.exists(f"{d}/motion_{k}.p1"):
        rc, H, mask, _ = oracle.usac_find_homography(load(f"{d}/motion_{k}.p1"), load(f"{d}/motion_{k}.p2"), float(load(f"{d}/motion_{k}.thr")[0]))
        save(f"{d}/motion_{k}.out_H", H.astype(np.float64)); save(f"{d}/motion_{k}.out_mask", mask); k += 1
    for k, (n, sigma) in enumerate(load(f"{d}/gauss.n_sigma")):
        save(f"{d}/gauss_{k}.out", np_smoother.gaussian_kernel_f32(int(n), float(sigma)))
    q = load(f"{d}/persp.quads")
    save(f"{d}/persp.out", np.array([oracle.get_perspective_transform(r[:8], r[8:])[1].reshape(-1) for r in q], np.float64))
    mesh = load(f"{d}/meshmap.mesh"); w, h = load(f"{d}/meshmap.size")
    save(f"{d}/meshmap.out", (oracle.mesh_to_map(mesh, int(h), int(w)) / np.array([w, h], np.float32)).astype(np.float32))
    plane = load(f"{d}/chroma.plane")
    packed = oracle.ingest_yuv420(np.zeros((plane.shape[0] * 2, plane.shape[1] * 2), np.uint8), plane, plane)
    save(f"{d}/chroma.out_up", packed[..., 1]); save(f"{d}/chroma.out_down", oracle.egress_yuv420(packed)[1])
    from tests import test_mesh_lstsq as ml
    k = 0
    while os.path.exists(f"{d}/lscg_{k}.rows"):                             # the exported triplets, solved by the RESTATED Eigen LSCG (tests/test_mesh_lstsq.py)
        m, n = load(f"{d}/lscg_{k}.shape")
        A = np.zeros((int(m), int(n)), np.float64)
        A[load(f"{d}/lscg_{k}.rows"), load(f"{d}/lscg_{k}.cols")] = load(f"{d}/lscg_{k}.vals")
        x, it, _ = ml.eigen_lscg(A, load(f"{d}/lscg_{k}.b").astype(np.float64), load(f"{d}/lscg_{k}.x0"))
        save(f"{d}/lscg_{k}.out", x.astype(np.float32)); save(f"{d}/lscg_{k}.out_iters", np.array([it], np.int32)); k += 1


if __name__ == "__main__":
    if len(sys.argv) != 3 or sys.argv[1] not in ("export", "compare"):
        print(__doc__); sys.exit(2)
    sys.exit(export(sys.argv[2]) or 0 if sys.argv[1] == "export" else compare(sys.argv[2]))
