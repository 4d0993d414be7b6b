"""Row a10 on the device: lvk_hip_mesh_solver_* (FrameTracker::estimate_local_motions, Vision/FrameTracker.cpp:200-321,380-457 -- band
L D L^T of the normal equations in k_mesh_assemble / k_mesh_prepare / k_mesh_solve) against the CPU oracle (oracle/mesh_solver.cpp).
Bar: bit-identical inlier flags and mesh offsets, frame after frame (the solver carries the previous solution)."""
import numpy as np
import pytest

from tests import oracle_lib

pytestmark = pytest.mark.gpu


def _pairs(rng, n, region, frame, outliers=0.11):
    w, h = region
    a = np.c_[rng.uniform(2, w - 2, n), rng.uniform(2, h - 2, n)].astype(np.float32)
    sx, th = 1.0 + 0.004 * frame, 0.003 * (frame + 1)
    b = np.empty_like(a)
    b[:, 0] = sx * (a[:, 0] * np.cos(th) - a[:, 1] * np.sin(th)) + 1.5
    b[:, 1] = sx * (a[:, 0] * np.sin(th) + a[:, 1] * np.cos(th)) - 0.8
    b += rng.uniform(-0.3, 0.3, b.shape).astype(np.float32)
    bad = rng.random(n) < outliers
    b[bad] += rng.uniform(-40, 40, (int(bad.sum()), 2)).astype(np.float32)
    return a, b.astype(np.float32)


@pytest.mark.parametrize("mesh,region", [((16, 16), (480, 270)), ((16, 16), (256, 256)), ((2, 2), (256, 256)), ((5, 7), (320, 180)), ((16, 9), (480, 270)), ((3, 40), (200, 600)),
                                         ((16, 64), (480, 1200)), ((4, 3), (320, 180)), ((9, 11), (300, 300)), ((13, 5), (480, 270)),
                                         # beyond the register-window kernels (Math/WarpMesh.cpp:34-41,79-90 allows any N x M): the generic kernels
                                         ((17, 17), (480, 270)), ((32, 32), (480, 270)), ((20, 6), (480, 270)), ((16, 70), (480, 1300)), ((40, 3), (640, 120)),
                                         # ... and the generic kernels on meshes the register-window kernels take too (LVK_HIP_MESH_GENERIC): same bits
                                         ((16, 8, "generic"), (480, 270)), ((5, 7, "generic"), (320, 180)), ((2, 2, "generic"), (256, 256)),
                                         # nested dissection (oracle S5': 8..16 columns, >= 9 rows -- (16, 16), (16, 9), (16, 64), (9, 11), (16, 70) above):
                                         # a last row that is itself a separator, a one-row last block, narrow meshes, odd column counts
                                         ((16, 13), (480, 270)), ((12, 10), (480, 270)), ((8, 9), (320, 180)), ((9, 14), (300, 300)), ((13, 21), (480, 800)),
                                         ((16, 16, "wide"), (1920, 1080))])
def test_mesh_solver_bit_exact_over_frames(ctx, oracle, mesh, region, monkeypatch):
    if len(mesh) == 3:
        if mesh[2] == "generic":
            monkeypatch.setenv("LVK_HIP_MESH_GENERIC", "1")
        mesh = mesh[:2]
    cols, rows = mesh
    rng = np.random.default_rng(cols * 100 + rows)
    ref = oracle_lib.OracleMeshSolver(oracle, cols, rows, gen_region=region)
    dev = ctx.mesh_solver(cols, rows, gen_region=region, max_points=2048)
    kw, kh = region[0] / (cols - 1), region[1] / (rows - 1)
    for frame in range(5):
        n = 900 - 60 * frame
        a, b = _pairs(rng, n, region, frame)
        # keep the points out of the last cell row / column, where the reference indexes past the mesh (separate test)
        a[:, 0] = np.minimum(a[:, 0], np.float32(kw * (cols - 1) - 1.0)); a[:, 1] = np.minimum(a[:, 1], np.float32(kh * (rows - 1) - 1.0))
        rc_o, inl_o, off_o = ref.solve(a, b, region=region, temporal=1.0, threshold=10.0)
        rc_d, inl_d, off_d = dev.solve(a, b, region=region, temporal=1.0, threshold=10.0)
        assert rc_o == 0 and rc_d == 0, (frame, rc_o, rc_d)
        assert np.array_equal(inl_o, inl_d), frame
        assert np.array_equal(off_o.reshape(-1).view(np.uint32), off_d.reshape(-1).view(np.uint32)), (mesh, frame, np.abs(off_o.reshape(-1) - off_d.reshape(-1)).max())
    # restart: the previous solution is forgotten on both sides
    ref.reset(); dev.reset()
    a, b = _pairs(rng, 500, region, 0)
    a[:, 0] = np.minimum(a[:, 0], np.float32(kw * (cols - 1) - 1.0)); a[:, 1] = np.minimum(a[:, 1], np.float32(kh * (rows - 1) - 1.0))
    rc_o, inl_o, off_o = ref.solve(a, b, region=region, temporal=1.0, threshold=10.0)
    rc_d, inl_d, off_d = dev.solve(a, b, region=region, temporal=1.0, threshold=10.0)
    assert rc_o == 0 and rc_d == 0 and np.array_equal(inl_o, inl_d)
    assert np.array_equal(off_o.reshape(-1).view(np.uint32), off_d.reshape(-1).view(np.uint32))
    ref.close(); dev.close()


def test_mesh_solver_no_estimate_cases(ctx, oracle):
    """A point in the mesh's last cell row makes the reference skip the estimate (FrameTracker.cpp:243-247 would index past the mesh):
    status 2 on the device, -1 from the oracle, and the previous solution stays untouched on both sides."""
    region = (480, 270)
    rng = np.random.default_rng(3)
    ref = oracle_lib.OracleMeshSolver(oracle, 16, 16, gen_region=region)
    dev = ctx.mesh_solver(16, 16, gen_region=region, max_points=1024)
    a, b = _pairs(rng, 600, region, 1)
    a[:, 0] = np.minimum(a[:, 0], 440.0); a[:, 1] = np.minimum(a[:, 1], 240.0)
    assert ref.solve(a, b, region=region)[0] == 0 and dev.solve(a, b, region=region)[0] == 0
    bad = a.copy(); bad[17] = (500.0, 285.0)           # beyond the last mesh vertex: cell (15, 15), whose far corner is not in the mesh
    assert ref.solve(bad, b, region=region)[0] < 0
    assert dev.solve(bad, b, region=region)[0] == 2
    # both continue from the state of the first solve
    a2, b2 = _pairs(rng, 500, region, 2)
    a2[:, 0] = np.minimum(a2[:, 0], 440.0); a2[:, 1] = np.minimum(a2[:, 1], 240.0)
    rc_o, inl_o, off_o = ref.solve(a2, b2, region=region)
    rc_d, inl_d, off_d = dev.solve(a2, b2, region=region)
    assert rc_o == 0 and rc_d == 0 and np.array_equal(inl_o, inl_d)
    assert np.array_equal(off_o.reshape(-1).view(np.uint32), off_d.reshape(-1).view(np.uint32))
    ref.close(); dev.close()
