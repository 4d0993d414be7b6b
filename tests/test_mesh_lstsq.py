"""Row a10: is the oracle's mesh (oracle/mesh_solver.cpp, and with it the device kernels that match it bit for bit) THE least-squares
minimiser of the reference's system?

A third, independent restatement: this file rebuilds the constraint matrix A and the right-hand side b in numpy straight from the
reference's lines -- generate_mesh_constraints (Vision/FrameTracker.cpp:380-457), the temporal / feature rows of estimate_local_motions
(:219-262), VirtualGrid::key_of / key_to_point (Math/VirtualGrid.cpp), barycentric_rect (Functions/Math.tpp:247-265) -- as a dense
matrix, solves it with np.linalg.lstsq in binary64 (no normal equations, no band, no Q32 sums: SVD), and compares the normalised mesh
offsets (:316-320) with the oracle's.  Bar: 1e-5 normalised, on a rotation + zoom + perspective-free
shear field with outliers (a pure translation would satisfy the similarity rows trivially -- round-2 VERDICT), over warm-started frames
(each side feeds its OWN previous solution back into the temporal rows).  The reference's solver, Eigen's LSCG, iterates towards this
same minimiser (:274-276)."""
import numpy as np
import pytest

from tests import oracle_lib

f32 = np.float32


def key_size(cols, rows, region):
    """VirtualGrid(mesh_size, Rect2f(tl, (Size2f(mesh_size) / Size2f(grid_size)) * region.size())).key_size(), binary32 (:207-216, :388-391)."""
    aw = f32(f32(cols) / f32(cols - 1)) * f32(region[0]); ah = f32(f32(rows) / f32(rows - 1)) * f32(region[1])
    return f32(aw / f32(cols)), f32(ah / f32(rows))


def static_rows(cols, rows, region, temporal, local, mutate=None):
    """generate_mesh_constraints (:380-457): list of rows, each a list of (column, value)."""
    kw, kh = key_size(cols, rows, region)
    out = []
    for index in range(cols * rows):                                      # temporal rows (:397-401)
        out.append([(2 * index, f32(temporal))]); out.append([(2 * index + 1, f32(temporal))])
    v1 = -(float(kw) / float(kh)); v2 = -1.0 / v1                          # :405, Size2f::aspectRatio -> double
    if mutate == "aspect":
        v1, v2 = v2, v1
    for r in range(rows):
        for c in range(cols):
            index = r * cols + c
            quad = 1                                                       # :410-418
            if c % 4 == 0 and r % 4 == 0:
                quad = 3
            elif (c + r) % 2 != 1 and c != 0 and r != 0 and c != cols - 2 and r != rows - 2:
                continue
            if c >= cols - quad or r >= rows - quad:                       # :421-422
                continue
            i00 = 2 * index; i10 = i00 + 2 * quad                          # :425-426
            i01 = 2 * (index + quad * cols); i11 = i01 + 2 * quad
            w = f32(local); w1 = f32(v1 * float(w)); w2 = f32(v2 * float(w))
            if mutate == "sign":
                w2 = -w2
            out.append([(i00, -w), (i01, w), (i01 + 1, -w2), (i11 + 1, w2)])            # upper triangle (:432-441)
            out.append([(i00 + 1, -w), (i01, w2), (i01 + 1, w), (i11, -w2)])
            out.append([(i00, -w), (i10, w), (i10 + 1, -w1), (i11 + 1, w1)])            # lower triangle (:444-453)
            out.append([(i00 + 1, -w), (i10, w1), (i10 + 1, w), (i11, -w1)])
    return out


def eigen_lscg(A, b, x0):
    """Eigen 3.4 LeastSquaresConjugateGradient<SparseMatrix<float>>::solveWithGuess(b, x0) with its defaults, restated from
    Eigen/src/IterativeLinearSolvers/LeastSquareConjugateGradient.h (the library is absent: unpinned): binary32 throughout,
    LeastSquareDiagonalPreconditioner (1 / squared column norm), tolerance = epsilon<float> on ||A'(b - A x)|| / ||A' b||, at most
    2 * cols iterations.  Returns (x, iterations, final relative normal-residual).  (Dense binary32 products here: the sparse products of
    the library sum the same terms in another order -- a 1e-7 relative effect, against the 1e-5 distances measured below.)"""
    A = A.astype(f32); b = b.astype(f32); x = x0.astype(f32).copy()
    n = A.shape[1]
    tol = np.finfo(f32).eps; max_iters = 2 * n
    col2 = (A * A).sum(axis=0, dtype=f32)
    invdiag = np.where(col2 > 0, f32(1) / col2, f32(1)).astype(f32)
    residual = (b - A @ x).astype(f32)
    normal = (A.T @ residual).astype(f32)
    rhs2 = f32(np.dot(A.T @ b, A.T @ b))
    if rhs2 == 0:
        return np.zeros(n, f32), 0, 0.0
    threshold = f32(tol * tol * rhs2)
    res2 = f32(np.dot(normal, normal))
    if res2 < threshold:
        return x, 0, float(np.sqrt(res2 / rhs2))
    p = (invdiag * normal).astype(f32)
    abs_new = f32(np.dot(normal, p))
    i = 0
    while i < max_iters:
        tmp = (A @ p).astype(f32)
        alpha = f32(abs_new / f32(np.dot(tmp, tmp)))
        x = (x + alpha * p).astype(f32)
        residual = (residual - alpha * tmp).astype(f32)
        normal = (A.T @ residual).astype(f32)
        res2 = f32(np.dot(normal, normal))
        if res2 < threshold:
            break
        z = (invdiag * normal).astype(f32)
        abs_old = abs_new
        abs_new = f32(np.dot(normal, z))
        p = (z + f32(abs_new / abs_old) * p).astype(f32)
        i += 1
    return x, i, float(np.sqrt(res2 / rhs2))


def build_system(cols, rows, static, region, temporal_now, prev_mesh, tracked, matched):
    """The sparse least-squares system of estimate_local_motions (:219-268) as a dense (A, b) pair plus, per feature, its four vertex indices and
    barycentric weights.  (Also what scripts/opencv_ref/opencv_ref_compare.py exports for a real Eigen to solve.)"""
    return _system(cols, rows, static, region, temporal_now, prev_mesh, tracked, matched)


def solve_frame(cols, rows, static, region, temporal_now, prev_mesh, tracked, matched, threshold, solver="lstsq", info=None):
    """estimate_local_motions (:200-321); solver "lstsq": np.linalg.lstsq (binary64 SVD) in place of Eigen::LeastSquaresConjugateGradient,
    "lscg": the restated Eigen solver itself, warm-started from prev_mesh (:274-276)."""
    kw, kh = key_size(cols, rows, region)
    A, b, feats = _system(cols, rows, static, region, temporal_now, prev_mesh, tracked, matched)
    return _finish(cols, rows, region, kw, kh, A, b, feats, prev_mesh, matched, threshold, solver, info)


def _system(cols, rows, static, region, temporal_now, prev_mesh, tracked, matched):
    kw, kh = key_size(cols, rows, region)
    n = 2 * cols * rows
    m = len(static) + 2 * len(tracked)
    A = np.zeros((m, n), np.float64); b = np.zeros(m, np.float64)
    for i, row in enumerate(static):
        for col, val in row:
            A[i, col] += float(val)
    for i in range(n):                                                     # :224-230 (x, y interleaved: row 2 index + comp)
        b[i] = float(f32(temporal_now) * f32(prev_mesh[i]))
    feats = []
    at = len(static)
    for (sx, sy), (dx, dy) in zip(tracked, matched):
        kx = min(max(int(f32(sx) / kw), 0), cols - 1); ky = min(max(int(f32(sy) / kh), 0), rows - 1)       # key_of + clamp (:241-244)
        i00 = 2 * (ky * cols + kx); i11 = 2 * ((ky + 1) * cols + kx + 1)
        i10 = i00 + 2; i01 = i11 - 2                                        # :248-250
        x1 = f32(kx) * kw; y1 = f32(ky) * kh                                # key_to_point; Rect_(pt1, pt2): width = x2 - x1
        rw = f32(f32(kx + 1) * kw - x1); rh = f32(f32(ky + 1) * kh - y1)
        inv = f32(1) / f32(rw * rh)
        x2 = f32(x1 + rw); y2 = f32(y1 + rh)
        rx1 = f32(x2 - f32(sx)); ry1 = f32(y2 - f32(sy)); rx2 = f32(f32(sx) - x1); ry2 = f32(f32(sy) - y1)
        w = [f32(f32(rx1 * ry1) * inv), f32(f32(rx1 * ry2) * inv), f32(f32(rx2 * ry2) * inv), f32(f32(rx2 * ry1) * inv)]      # TL BL BR TR
        ids = [i00, i01, i11, i10]
        for comp, dst in ((0, dx), (1, dy)):                                # :256-268
            for q in range(4):
                A[at, ids[q] + comp] += float(w[q])
            b[at] = float(dst); at += 1
        feats.append((ids, w))
    return A, b, feats


def _finish(cols, rows, region, kw, kh, A, b, feats, prev_mesh, matched, threshold, solver, info):
    if solver == "lscg":
        # the reference's operands are binary32: A's triplets and b are floats (:221-222)
        x, iters, rel = eigen_lscg(A, b, np.asarray(prev_mesh, f32))
        if info is not None:
            info.append((iters, rel))
    else:
        x = np.linalg.lstsq(A, b, rcond=None)[0]
    mesh = x.astype(f32)                                                    # Eigen::VectorXf m_OptimizedMesh
    return (mesh,) + mesh_to_result(cols, rows, region, kw, kh, mesh, feats, matched, threshold)


def mesh_to_result(cols, rows, region, kw, kh, mesh, feats, matched, threshold):
    """inlier flags (:279-310) and normalised offsets (:316-320) of a solved mesh"""
    inl = np.zeros(len(matched), np.uint8)
    for k, ((ids, w), (dx, dy)) in enumerate(zip(feats, matched)):          # :279-310
        px = sum(float(w[q]) * float(mesh[ids[q]]) for q in range(4)); py = sum(float(w[q]) * float(mesh[ids[q] + 1]) for q in range(4))
        inl[k] = (abs(px - dx) + abs(py - dy)) < threshold
    off = np.zeros((rows, cols, 2), np.float64)                             # :316-320
    for r in range(rows):
        for c in range(cols):
            off[r, c, 0] = (float(f32(c) * kw) - float(mesh[2 * (r * cols + c)])) / region[0]
            off[r, c, 1] = (float(f32(r) * kh) - float(mesh[2 * (r * cols + c) + 1])) / region[1]
    return inl, off


def field_pairs(rng, n, region, frame, outliers=0.12):
    """rotation + zoom + anisotropic shear about an off-centre point, small noise, gross outliers; the points stay clear of the last cell
    row / column (where the reference indexes past the mesh: separate tests)."""
    w, h = region
    a = np.c_[rng.uniform(2, w * 0.92, n), rng.uniform(2, h * 0.92, n)].astype(f32)
    th = 0.012 * (frame + 1); s = 1.0 + 0.01 * (frame + 1)
    cx, cy = 0.4 * w, 0.55 * h
    x, y = a[:, 0] - cx, a[:, 1] - cy
    b = np.c_[s * (np.cos(th) * x - np.sin(th) * y) + 0.004 * y + cx + 1.7, s * (np.sin(th) * x + np.cos(th) * y) - 0.003 * x + cy - 0.9]
    b += rng.normal(0, 0.08, b.shape)
    bad = rng.random(n) < outliers
    b[bad] += rng.uniform(-30, 30, (int(bad.sum()), 2))
    return a, b.astype(f32)


@pytest.mark.parametrize("cols,rows,region,gen_region,ts_gen,ts_now", [
    (16, 16, (480, 270), (480, 270), 1.0, 1.0),          # the OBS vector-field preset
    (16, 16, (480, 270), (256, 256), 1.0, 1.0),          # constraints generated for FrameTracker's default region (the stale-constraint quirk)
    (16, 16, (480, 270), (480, 270), 1.0, 0.5),          # temporal weight changed after the constraints were generated (:229-230 vs :399-400)
    (9, 7, (320, 180), (320, 180), 2.0, 2.0),
    (17, 17, (480, 270), (480, 270), 1.0, 1.0),          # beyond the register-window device solver
    (12, 10, (480, 270), (480, 270), 1.0, 1.0),          # nested dissection (oracle S5'): separators at rows 4 and 8, a one-row last block
    (16, 9, (480, 270), (480, 270), 1.0, 1.0),           # ... the last row itself a separator
    (8, 13, (320, 400), (320, 400), 1.0, 1.0),           # ... three separators, narrow mesh
])
def test_oracle_mesh_is_the_least_squares_minimiser(oracle, cols, rows, region, gen_region, ts_gen, ts_now):
    rng = np.random.default_rng(cols * 31 + rows)
    static = static_rows(cols, rows, gen_region, ts_gen, 20.0)
    ref = oracle_lib.OracleMeshSolver(oracle, cols, rows, gen_region=gen_region, temporal=ts_gen, local=20.0)
    # the row / triplet counts of the generator against the oracle's own (1108 rows / 2896 triplets for the preset)
    assert (len(static), sum(len(r) for r in static)) == ref.static_counts()
    if (cols, rows) == (16, 16):
        assert len(static) == 1108 and sum(len(r) for r in static) == 2896
    prev = np.zeros(2 * cols * rows, f32)
    worst = 0.0; fracs = []
    for frame in range(5):
        a, b = field_pairs(rng, 700 - 50 * frame, region, frame)
        rc, inl_o, off_o = ref.solve(a, b, region=region, temporal=ts_now, threshold=10.0)
        assert rc == 0
        prev, inl_n, off_n = solve_frame(cols, rows, static, region, ts_now, prev, a, b, 10.0)
        d = np.abs(off_o.astype(np.float64).reshape(rows, cols, 2) - off_n).max()
        worst = max(worst, d)
        assert d <= 1e-5, (frame, d)
        assert (inl_o != inl_n).mean() <= 0.005, frame                     # identical except a pair sitting on the threshold
        fracs.append(float(inl_o.mean()))
        # the field is not a translation: the solved mesh bends (offsets vary across the mesh by far more than the tolerance)
        assert np.ptp(off_n[..., 0]) > 20e-5 and np.ptp(off_n[..., 1]) > 20e-5
    # m_OptimizedMesh starts at ZERO (FrameTracker.cpp:52,103) and the temporal rows pull towards it: the first solves after a (re)start
    # sit between the origin and the true vertex positions (hardly any inliers), then the warm start catches up -- reference behaviour
    assert fracs[0] < 0.2 and fracs[-1] > fracs[0]
    print("\n[a10 lstsq] %dx%d gen %s ts %.1f/%.1f: max |offset difference| %.2e (normalised), inlier fraction per frame %s"
          % (cols, rows, gen_region, ts_gen, ts_now, worst, np.round(fracs, 3).tolist()))
    ref.close()


@pytest.mark.parametrize("mutate", ["sign", "aspect"])
def test_the_lstsq_check_would_catch_a_wrong_similarity_row(oracle, mutate):
    """Sensitivity: flip the sign of w2 or swap the aspect terms v1 / v2 (FrameTracker.cpp:405,429) in the numpy generator and the
    minimiser moves by far more than the 1e-5 bar -- i.e. the test above does pin those coefficients of the oracle."""
    cols = rows = 16; region = (480, 270)
    rng = np.random.default_rng(5)
    ref = oracle_lib.OracleMeshSolver(oracle, cols, rows, gen_region=region, temporal=1.0, local=20.0)
    a, b = field_pairs(rng, 700, region, 2)
    rc, _, off_o = ref.solve(a, b, region=region, temporal=1.0, threshold=10.0)
    assert rc == 0
    good = solve_frame(cols, rows, static_rows(cols, rows, region, 1.0, 20.0), region, 1.0, np.zeros(512, f32), a, b, 10.0)[2]
    bad = solve_frame(cols, rows, static_rows(cols, rows, region, 1.0, 20.0, mutate=mutate), region, 1.0, np.zeros(512, f32), a, b, 10.0)[2]
    assert np.abs(off_o.astype(np.float64).reshape(rows, cols, 2) - good).max() <= 1e-5
    assert np.abs(off_o.astype(np.float64).reshape(rows, cols, 2) - bad).max() > 1e-4
    ref.close()


@pytest.mark.parametrize("cols,rows,region", [(16, 16, (480, 270)), (9, 7, (320, 180))])
def test_distance_to_where_eigens_lscg_stops(oracle, cols, rows, region):
    """The reference does not compute the minimiser: it runs Eigen's LSCG in binary32 from the previous mesh (FrameTracker.cpp:274-276) and
    takes whatever the iteration has reached after at most 2 n steps.  eigen_lscg restates that solver; each side feeds ITS OWN previous
    solution into the temporal rows and the warm start, as two real filters would.  Measured (16 x 16 preset, 8 frames): the iteration
    meets its epsilon<float> tolerance after 159-173 of its 1024 allowed steps (relative normal residual 1e-7) and ends within 2.0e-6
    (normalised; 0.001 px at 480 px) of the specification's exact minimiser -- inside SURVEY 8c's 1e-5; 9 x 7: 65 steps, 8.6e-7."""
    rng = np.random.default_rng(cols * 7 + rows)
    static = static_rows(cols, rows, region, 1.0, 20.0)
    ref = oracle_lib.OracleMeshSolver(oracle, cols, rows, gen_region=region, temporal=1.0, local=20.0)
    prev = np.zeros(2 * cols * rows, f32)
    info = []
    worst = 0.0
    for frame in range(8):
        a, b = field_pairs(rng, 700 - 40 * frame, region, frame % 5)
        rc, inl_o, off_o = ref.solve(a, b, region=region, temporal=1.0, threshold=10.0)
        assert rc == 0
        prev, inl_c, off_c = solve_frame(cols, rows, static, region, 1.0, prev, a, b, 10.0, solver="lscg", info=info)
        d = np.abs(off_o.astype(np.float64).reshape(rows, cols, 2) - off_c).max()
        worst = max(worst, d)
        assert d <= 1e-5, (frame, d, info[-1])
        assert (inl_o != inl_c).mean() <= 0.005
    print("\n[a10 lscg] %dx%d: max |offset difference| product specification vs restated Eigen LSCG %.2e (normalised) = %.4f px; "
          "iterations / final relative normal residual per frame: %s" % (cols, rows, worst, worst * region[0], [(i, float("%.1e" % r)) for i, r in info]))
    ref.close()
