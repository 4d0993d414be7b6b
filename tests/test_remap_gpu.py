"""GPU parity of the dense remap against the CPU oracle, through the C-ABI.
Bar: bit-exact (the kernel and the oracle implement the same binary32 op sequence)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _to_gpu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _assert_same(got, want, what):
    got = got.cpu().numpy()
    if not np.array_equal(got, want):
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        ys, xs = np.nonzero(d.max(axis=2))
        raise AssertionError(f"{what}: {len(ys)} pixels differ, max |d| = {d.max()}, first at (x={xs[0]}, y={ys[0]}): "
                             f"gpu={got[ys[0], xs[0]]} oracle={want[ys[0], xs[0]]}")


@pytest.mark.parametrize("yuv", [True, False])
@pytest.mark.parametrize("size", [(72, 96), (67, 131), (270, 480), (5, 9), (8, 8)])
def test_remap_homography_bit_exact(ctx, oracle, yuv, size):
    rows, cols = size
    rng = np.random.default_rng(rows * 1000 + cols)
    src = synth.textured_frame(rows, cols, seed=rows + cols) if rows >= 32 else rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    dsrc = _to_gpu(src)
    for t in range(3):
        H = synth.random_homography(rows, cols, rng, strength=2.5)
        want = oracle.remap_homography(src, H, bg=(3, 200, 77), yuv=yuv)
        got = ctx.remap_homography(dsrc, H, bg=(3, 200, 77), yuv=yuv)
        ctx.sync()
        _assert_same(got, want, f"homography {size} yuv={yuv} trial {t}")


def test_remap_homography_degenerate_and_far_out(ctx, oracle):
    """Background fill, the (-1, 0) truncation quirk, huge / non-finite source coordinates."""
    src = synth.textured_frame(64, 80, seed=21)
    dsrc = _to_gpu(src)
    cases = []
    H = np.eye(3, dtype=np.float32); H[0, 2] = -0.5; cases.append(H)
    H = np.eye(3, dtype=np.float32); H[0, 2] = 500.0; cases.append(H)             # everything outside
    H = np.eye(3, dtype=np.float32); H[2, 0] = -1.0 / 40.0; cases.append(H)       # denominator crosses zero -> inf/nan
    H = np.eye(3, dtype=np.float32) * np.float32(1e20); cases.append(H)
    H = np.zeros((3, 3), np.float32); cases.append(H)                             # 0/0
    for i, H in enumerate(cases):
        want = oracle.remap_homography(src, H, bg=(10, 20, 30), yuv=True)
        got = ctx.remap_homography(dsrc, H, bg=(10, 20, 30), yuv=True)
        ctx.sync()
        _assert_same(got, want, f"degenerate case {i}")


def test_remap_homography_roi_offset_and_pitch(ctx, oracle):
    """dst smaller than src with an ROI offset (Image.cpp:121-123) and padded row pitches."""
    import torch
    src = synth.textured_frame(90, 120, seed=5)
    rng = np.random.default_rng(2)
    H = synth.random_homography(90, 120, rng)
    want = oracle.remap_homography(src, H, yuv=True, dst_size=(40, 50), offset=(7, 11))
    big = torch.zeros((90, 160, 3), dtype=torch.uint8, device="cuda")
    big[:, :120] = torch.from_numpy(src).cuda()
    dsrc = big[:, :120]                                   # row pitch 480 B instead of 360 B
    outbuf = torch.zeros((40, 70, 3), dtype=torch.uint8, device="cuda")
    out = outbuf[:, :50]
    ctx.remap_homography(dsrc, H, yuv=True, out=out, dst_size=(40, 50), offset=(7, 11))
    ctx.sync()
    _assert_same(out, want, "roi/pitch")
    assert int(outbuf[:, 50:].sum()) == 0                  # nothing written past the row


@pytest.mark.parametrize("mesh_size", [(3, 3), (4, 7), (16, 16), (32, 32)])
@pytest.mark.parametrize("yuv", [True, False])
def test_remap_mesh_bit_exact(ctx, oracle, mesh_size, yuv):
    rng = np.random.default_rng(mesh_size[0] * 100 + mesh_size[1])
    for (rows, cols) in [(135, 240), (97, 203)]:
        src = synth.textured_frame(rows, cols, seed=rows)
        dsrc = _to_gpu(src)
        mesh = synth.random_mesh(*mesh_size, rng, amp=0.03)
        want = oracle.remap_mesh(src, mesh, bg=(0, 128, 128), yuv=yuv)
        got = ctx.remap_mesh(dsrc, mesh, bg=(0, 128, 128), yuv=yuv)
        ctx.sync()
        _assert_same(got, want, f"mesh {mesh_size} {rows}x{cols}")


def test_warpmesh_apply_2x2_goes_through_homography(ctx, oracle):
    rng = np.random.default_rng(9)
    src = synth.textured_frame(120, 160, seed=12)
    dsrc = _to_gpu(src)
    for t in range(4):
        mesh = synth.random_mesh(2, 2, rng, amp=0.04)
        want = oracle.warpmesh_apply(src, mesh, yuv=True)
        got = ctx.warpmesh_apply(dsrc, mesh, yuv=True)
        ctx.sync()
        _assert_same(got, want, f"warpmesh 2x2 trial {t}")


def test_many_meshes_in_flight_use_distinct_staging(ctx, oracle):
    """Back-to-back launches with different meshes must not overwrite each other's staged parameters."""
    rng = np.random.default_rng(4)
    src = synth.textured_frame(64, 96, seed=13)
    dsrc = _to_gpu(src)
    meshes = [synth.random_mesh(8, 8, rng, amp=0.03) for _ in range(40)]
    outs = [ctx.remap_mesh(dsrc, m, yuv=True) for m in meshes]
    ctx.sync()
    for i in (0, 7, 16, 17, 39):
        _assert_same(outs[i], oracle.remap_mesh(src, meshes[i], yuv=True), f"in-flight mesh {i}")


def test_full_size_4k_properties(ctx, oracle):
    """At BASELINE's full size the oracle is too slow to run everywhere; check size-independent properties:
    integer-shift equivariance over the whole frame, plus oracle equality on a band of rows."""
    import torch
    rows, cols = 2160, 3840
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    dsrc = torch.randint(0, 256, (rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g)
    I = np.eye(3, dtype=np.float32)
    S = I.copy(); S[0, 2] = 5.0; S[1, 2] = 3.0
    a = ctx.remap_homography(dsrc, I, yuv=True)
    b = ctx.remap_homography(dsrc, S, yuv=True)
    ctx.sync()
    assert torch.equal(b[1:rows - 8, 1:cols - 10], a[4:rows - 5, 6:cols - 5])
    # oracle on a 64-row band (the band is remapped as its own small frame on both sides)
    band = dsrc[1000:1064].contiguous()
    rng = np.random.default_rng(0)
    H = synth.random_homography(64, cols, rng, strength=0.5)
    want = oracle.remap_homography(band.cpu().numpy(), H, yuv=True)
    got = ctx.remap_homography(band, H, yuv=True)
    ctx.sync()
    _assert_same(got, want, "4K band")


def test_remap_strong_distortions_take_the_gather_path(ctx, oracle):
    """Zoom-out / rotation large enough that a tile's source window exceeds the LDS staging capacity: the kernel
    then gathers taps from global memory -- same bits either way."""
    src = synth.textured_frame(300, 400, seed=31)
    dsrc = _to_gpu(src)
    cases = []
    H = np.eye(3, dtype=np.float32); H[0, 0] = 3.0; H[1, 1] = 3.0; cases.append(H)                      # 3x minification
    th = 0.6; c, s_ = np.cos(th), np.sin(th)
    H = np.array([[c, -s_, 120], [s_, c, -60], [0, 0, 1]], np.float32); cases.append(H)                # 34 degree rotation
    H = np.array([[1, 0.4, 0], [0.3, 1, 0], [1e-3, 5e-4, 1]], np.float32); cases.append(H)            # shear + strong perspective
    H = np.array([[0.2, 0, 50], [0, 0.2, 40], [0, 0, 1]], np.float32); cases.append(H)                # 5x magnification (tiny window)
    for i, H in enumerate(cases):
        for yuv in (True, False):
            want = oracle.remap_homography(src, H, bg=(7, 8, 9), yuv=yuv)
            got = ctx.remap_homography(dsrc, H, bg=(7, 8, 9), yuv=yuv)
            ctx.sync()
            _assert_same(got, want, f"distortion case {i} yuv={yuv}")
    mesh = synth.random_mesh(6, 6, np.random.default_rng(1), amp=0.35)                                  # violent mesh
    want = oracle.remap_mesh(src, mesh, yuv=True)
    got = ctx.remap_mesh(dsrc, mesh, yuv=True)
    ctx.sync()
    _assert_same(got, want, "violent mesh")


@pytest.mark.parametrize("nv12", [False, True])
@pytest.mark.parametrize("size", [(72, 96), (66, 130), (270, 480), (8, 8), (6, 10)])
def test_warpmesh_apply_yuv420_equals_apply_then_egress(ctx, oracle, nv12, size):
    """remap + 4:2:0 egress in one kernel == the two-kernel chain == the oracle's chain, bit for bit."""
    rows, cols = size
    rng = np.random.default_rng(rows + cols)
    src = synth.textured_frame(rows, cols, seed=rows) if rows >= 32 else rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    dsrc = _to_gpu(src)
    for mesh in (np.zeros((2, 2, 2), np.float32), rng.uniform(-0.03, 0.03, (2, 2, 2)).astype(np.float32), synth.random_mesh(16, 16, rng, amp=0.02),
                 synth.random_mesh(3, 5, rng, amp=0.4)):
        want = oracle.egress_yuv420(oracle.warpmesh_apply(src, mesh, bg=(105, 212, 235), yuv=True), nv12=nv12)
        got = ctx.warpmesh_apply_yuv420(dsrc, mesh, bg=(105, 212, 235), nv12=nv12)
        ctx.sync()
        for a, b in zip(got, want):
            assert np.array_equal(a.cpu().numpy(), b), (size, nv12, mesh.shape)
