"""The reference pin of rows a16 / f4: the HIP kernels and the CPU oracle against the REFERENCE'S OWN OpenCL
kernels -- LiveVisionKit/Functions/OpenCL/Sources/{FSR,Drawing}.cl compiled for gfx950 by `make -C oracle ref` (oracle/_ref/*.hsaco)
and launched here with the argument lists of Functions/Image.cpp / Drawing.tpp (tests/ref_cl.py).

Bar: BIT-IDENTICAL, three ways (HIP == reference kernel == oracle), for easu_remap_homography, easu_remap, easu_scale, rcas, grid and
crosses at 360p ... 4K.  The only thing the reference leaves to the device is native_recip / the 2.5-ulp divide (v_rcp_f32 on gfx950):
the oracle models it with the committed device table tests/golden/gfx950_rcp.npz, which the first test checks against the GPU."""
import numpy as np
import pytest

from tests import oracle_lib, ref_cl, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    if not ref_cl.available():
        pytest.fail("oracle/_ref/*.hsaco missing: run `make -C oracle ref` where /root/reference exists (build() does)")
    return ref_cl.RefKernels()


def _gpu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _padded(a, pad=16):
    """A GPU copy of `a` living inside a larger allocation (the reference kernels' padding work-items touch memory past the frame)."""
    import torch
    rows, cols = a.shape[:2]
    big = torch.zeros((rows + pad, cols + pad) + tuple(a.shape[2:]), dtype=torch.from_numpy(a[:1]).dtype, device="cuda")
    view = big[:rows, :cols]
    view.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return view


def _same(a, b, what):
    a = a.cpu().numpy() if hasattr(a, "cpu") else a
    b = b.cpu().numpy() if hasattr(b, "cpu") else b
    if not np.array_equal(a, b):
        d = np.abs(a.astype(np.int32) - b.astype(np.int32))
        idx = np.argwhere(d.reshape(d.shape[0], d.shape[1], -1).max(axis=2) > 0)
        raise AssertionError(f"{what}: {len(idx)} pixels differ, max |d| = {d.max()}, first at (y, x) = {tuple(idx[0])}")


def _frame(rows, cols, seed, noise=False):
    if noise:
        return np.random.default_rng(seed).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    return synth.textured_frame(rows, cols, seed=seed)


def test_rcp_fixture_is_the_device(ctx):
    """tests/golden/gfx950_rcp.npz == v_rcp_f32 of this GPU over every mantissa; sign symmetry and exponent independence (what the
    oracle's native_rcp() assumes) on a spread of exponents."""
    import torch
    tab = oracle_lib.device_rcp_table()
    x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)
    got = ctx.native_rcp(torch.from_numpy(x).cuda()); ctx.sync()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), tab.view(np.uint32))
    for e in (-100, -20, -1, 3, 17, 90):
        xe = np.ldexp(x[::7], e).astype(np.float32)
        r = ctx.native_rcp(torch.from_numpy(xe).cuda()); ctx.sync()
        assert np.array_equal(r.cpu().numpy().view(np.uint32), np.ldexp(tab[::7], -e).astype(np.float32).view(np.uint32)), e
        rn = ctx.native_rcp(torch.from_numpy(-xe).cuda()); ctx.sync()
        assert np.array_equal(rn.cpu().numpy().view(np.uint32), (-np.ldexp(tab[::7], -e).astype(np.float32)).view(np.uint32)), e


@pytest.mark.parametrize("yuv", [True, False])
@pytest.mark.parametrize("size,noise", [((360, 640), False), ((360, 640), True), ((67, 131), True), ((1080, 1920), False), ((2160, 3840), False)])
def test_remap_homography_three_way(ctx, oracle, ref, yuv, size, noise):
    import torch
    rows, cols = size
    rng = np.random.default_rng(rows + cols + int(yuv))
    src = _frame(rows, cols, rows * 3 + cols, noise)
    d = _gpu(src)
    for t in range(2):
        H = synth.random_homography(rows, cols, rng, strength=1.0 + 1.5 * t)
        r = ref.remap_homography(d, H, bg=(3, 200, 77), yuv=yuv); torch.cuda.synchronize()
        g = ctx.remap_homography(d, H, bg=(3, 200, 77), yuv=yuv); ctx.sync()
        _same(g, r, f"HIP vs reference kernel {size} yuv={yuv} trial {t}")
        o = oracle.remap_homography(src, H, bg=(3, 200, 77), yuv=yuv, nthreads=32)
        _same(o, r, f"oracle vs reference kernel {size} yuv={yuv} trial {t}")


def test_remap_homography_roi_and_degenerate_three_way(ctx, oracle, ref):
    """dst ROI offset (Image.cpp:121-123, dst_bounds.xy) and the degenerate matrices: zero / huge / denominator crossing zero."""
    import torch
    src = _frame(90, 120, 5); d = _gpu(src)
    rng = np.random.default_rng(3)
    H = synth.random_homography(90, 120, rng, strength=2.0)
    r = ref.remap_homography(d, H, bg=(1, 2, 3), dst_size=(40, 56), offset=(17, 9)); torch.cuda.synchronize()
    g = ctx.remap_homography(d, H, bg=(1, 2, 3), dst_size=(40, 56), offset=(17, 9)); ctx.sync()
    _same(g, r, "ROI: HIP vs reference kernel")
    _same(oracle.remap_homography(src, H, bg=(1, 2, 3), dst_size=(40, 56), offset=(17, 9)), r, "ROI: oracle vs reference kernel")
    cases = []
    H = np.eye(3, dtype=np.float32); H[0, 2] = -0.5; cases.append(H)
    H = np.eye(3, dtype=np.float32); H[0, 2] = 500.0; cases.append(H)
    H = np.eye(3, dtype=np.float32); H[2, 0] = -1.0 / 40.0; cases.append(H)
    H = np.eye(3, dtype=np.float32) * np.float32(1e20); cases.append(H)
    H = np.eye(3, dtype=np.float32) * np.float32(1e-30); cases.append(H)
    # (the all-zero matrix gives NaN coordinates: convert_int2_rtz(NaN) is undefined in OpenCL C and the compiled reference does not
    #  behave like any fixed value there; tests/test_remap_gpu.py keeps that case for HIP vs oracle, both defining NaN -> 0)
    for i, H in enumerate(cases):
        r = ref.remap_homography(d, H, bg=(10, 20, 30)); torch.cuda.synchronize()
        g = ctx.remap_homography(d, H, bg=(10, 20, 30)); ctx.sync()
        _same(g, r, f"degenerate {i}: HIP vs reference kernel")
        _same(oracle.remap_homography(src, H, bg=(10, 20, 30)), r, f"degenerate {i}: oracle vs reference kernel")


@pytest.mark.parametrize("yuv", [True, False])
@pytest.mark.parametrize("size", [(360, 640), (270, 484), (1080, 1920), (2160, 3840)])
def test_remap_map_three_way(ctx, oracle, ref, yuv, size):
    """easu_remap on the W x H offset map WarpMesh::apply materialises (WarpMesh.cpp:190-191) == the HIP kernel reading the same map
    == the HIP kernel that interpolates the 16 x 16 mesh itself == the oracle."""
    import torch
    rows, cols = size
    rng = np.random.default_rng(rows)
    src = _frame(rows, cols, rows + 1); d = _gpu(src)
    mesh = synth.random_mesh(16, 16, rng, amp=0.012)
    omap = oracle.mesh_to_map(mesh, rows, cols)
    dm = _gpu(omap)
    r = ref.remap_map(d, dm, bg=(9, 8, 7), yuv=yuv); torch.cuda.synchronize()
    _same(ctx.remap_map(d, dm, bg=(9, 8, 7), yuv=yuv), r, f"map kernel vs reference kernel {size}")
    _same(ctx.remap_mesh(d, mesh, bg=(9, 8, 7), yuv=yuv), r, f"mesh kernel vs reference kernel {size}")
    ctx.sync()
    _same(oracle.remap_mesh(src, mesh, bg=(9, 8, 7), yuv=yuv, nthreads=32), r, f"oracle vs reference kernel {size}")


@pytest.mark.parametrize("yuv", [True, False])
@pytest.mark.parametrize("src_size,dst_size", [((270, 480), (540, 960)), ((61, 100), (80, 112)), ((720, 1280), (1080, 1920)), ((1080, 1920), (2160, 3840))])
def test_upscale_three_way(ctx, oracle, ref, yuv, src_size, dst_size):
    import torch
    src = _frame(src_size[0], src_size[1], 77); d = _gpu(src)
    size = (dst_size[1], dst_size[0])
    r = ref.upscale(d, size, yuv=yuv); torch.cuda.synchronize()
    g = ctx.upscale(d, size, yuv=yuv); ctx.sync()
    _same(g, r, f"HIP vs reference kernel {src_size}->{dst_size}")
    _same(oracle.upscale(src, size, yuv=yuv, nthreads=32), r, f"oracle vs reference kernel {src_size}->{dst_size}")


@pytest.mark.parametrize("size,noise,sharpness", [((360, 640), False, 0.7), ((64, 96), True, 1.0), ((61, 99), True, 0.0), ((1080, 1920), False, 0.35),
                                                  ((2160, 3840), False, 0.8)])
def test_rcas_three_way(ctx, oracle, ref, size, noise, sharpness):
    """rcas: interior + copied border.  The reference's padding work-items write (and read) past the frame (FSR.cl:478), so it runs
    inside padded allocations and the frame region is compared."""
    import torch
    rows, cols = size
    src = _frame(rows, cols, 123 + rows, noise)
    if noise:
        src[8:24, 8:40] = 0; src[30:44, 8:40] = 255          # flat rings: the 0 * inf limiter cases
    dsrc = _padded(src)
    out = _padded(np.zeros_like(src))
    ref.sharpen(dsrc, sharpness, out=out); torch.cuda.synchronize()
    g = ctx.sharpen(_gpu(src), sharpness); ctx.sync()
    _same(g, out, f"HIP vs reference kernel {size}")
    _same(oracle.sharpen(src, sharpness, nthreads=32), out, f"oracle vs reference kernel {size}")


def test_drawing_three_way(ctx, oracle, ref):
    """Drawing.cl grid / crosses (test-mode overlays, StabilizationFilter.cpp:163-188) vs k_draw_grid / k_draw_crosses vs the oracle."""
    import torch
    rows, cols = 270, 480
    base = _frame(rows, cols, 9)
    for grid, thick in (((16, 16), 1), ((5, 3), 2), ((7, 11), 3)):
        a = _padded(base)
        ref.draw_grid(a, np.float32(cols) / np.float32(grid[0]), np.float32(rows) / np.float32(grid[1]), thick, (29, 255, 107)); torch.cuda.synchronize()
        g = ctx.draw_grid(_gpu(base), grid, (29, 255, 107), thick); ctx.sync()
        _same(g, a, f"grid {grid}: HIP vs reference kernel")
        _same(oracle.draw_grid(base, grid, (29, 255, 107), thick), a, f"grid {grid}: oracle vs reference kernel")
    rng = np.random.default_rng(4)
    pts = np.concatenate([rng.uniform(-5, 245, (200, 2)), np.array([[0, 0], [239.5, 134.5], [1.5, 2.5], [2.5, 3.5]])]).astype(np.float32)
    scaling = (np.float32(cols) / np.float32(240), np.float32(rows) / np.float32(135))
    # cv::multiply(pts, scaling, CV_32S): binary32 product, round half to even
    pi = np.stack([np.rint(pts[:, 0] * scaling[0]), np.rint(pts[:, 1] * scaling[1])], axis=1).astype(np.int32)
    a = _padded(base, pad=64)
    ref.draw_crosses(a, _gpu(pi), 8, 4, (76, 84, 255)); torch.cuda.synchronize()
    g = ctx.draw_crosses(_gpu(base), pts, (76, 84, 255), 8, 4, scaling=(float(scaling[0]), float(scaling[1]))); ctx.sync()
    _same(g, a, "crosses: HIP vs reference kernel")
    _same(oracle.draw_crosses(base, pts, (76, 84, 255), 8, 4, scaling=(float(scaling[0]), float(scaling[1]))), a, "crosses: oracle vs reference kernel")
