"""CPU tests that pin the oracle's remap stage (no GPU).  The reference has no golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned by known-answer cases and by a second independent restatement."""
import numpy as np
import pytest

from tests import np_easu, synth

IDENT = np.eye(3, dtype=np.float32)


def test_identity_homography_border_band_and_flat_passthrough(oracle):
    """The 1-px / 4-px border band is a nearest copy (FSR.cl:387-397).  In a flat region with pp = 0 only the
    centre tap has weight, so the output is the source value pushed through `x * 0.00392156862f * 255.0f`
    truncation (EASU is NOT a pass-through on edges: the window stretches along them)."""
    src = synth.textured_frame(48, 64, seed=1)
    src[10:30, 10:40] = (77, 130, 201)
    out = oracle.remap_homography(src, IDENT, yuv=True)
    assert np.array_equal(out[0], src[0]) and np.array_equal(out[:, 0], src[:, 0])
    assert np.array_equal(out[-4:], src[-4:]) and np.array_equal(out[:, -4:], src[:, -4:])
    v = np.array([77, 130, 201], np.float32) * np.float32(0.00392156862)
    expect = (v * np.float32(255.0)).astype(np.int32)
    assert (out[14:26, 14:36] == expect.astype(np.uint8)).all()


def test_constant_image_is_fixed_point_up_to_truncation(oracle):
    src = np.full((40, 40, 3), 100, np.uint8)
    H = IDENT.copy(); H[0, 2] = 0.37; H[1, 2] = -0.21
    out = oracle.remap_homography(src, H, yuv=True)
    inner = out[6:-6, 6:-6]
    assert inner.min() >= 99 and inner.max() <= 100     # min/max clamp to the 2x2 centre (FSR.cl:316)


def test_out_of_frame_is_background_and_negative_fraction_quirk(oracle):
    src = synth.textured_frame(32, 32, seed=2)
    H = IDENT.copy(); H[0, 2] = -0.5                       # src x = x - 0.5
    out = oracle.remap_homography(src, H, bg=(1, 2, 3), yuv=True)
    # x = 0 -> src -0.5 truncates to pixel 0 -> nearest copy, not background
    assert np.array_equal(out[:, 0], src[:, 0])
    H[0, 2] = -1.0
    out = oracle.remap_homography(src, H, bg=(1, 2, 3), yuv=True)
    assert (out[:, 0] == np.array([1, 2, 3], np.uint8)).all()


def test_integer_shift_equivariance(oracle):
    """dst(x, y) = EASU(src, (x+3, y+2), pp = 0): an integer shift commutes with the filter."""
    src = synth.textured_frame(40, 56, seed=3)
    H = IDENT.copy(); H[0, 2] = 3.0; H[1, 2] = 2.0
    shifted = oracle.remap_homography(src, H, yuv=False)
    ident = oracle.remap_homography(src, IDENT, yuv=False)
    assert np.array_equal(shifted[1:30, 1:45], ident[3:32, 4:48])


@pytest.mark.parametrize("yuv", [True, False])
def test_oracle_matches_independent_numpy_restatement_homography(oracle, yuv):
    rng = np.random.default_rng(7)
    src = synth.textured_frame(72, 96, seed=4)
    for trial in range(3):
        H = synth.random_homography(72, 96, rng, strength=2.0)
        a = oracle.remap_homography(src, H, bg=(9, 8, 7), yuv=yuv)
        b = np_easu.remap_homography(src, H, (9, 8, 7), yuv)
        d = np.abs(a.astype(np.int32) - b.astype(np.int32))
        assert d.max() <= 1, f"trial {trial}: max diff {d.max()}"
        assert (d == 0).mean() > 0.9995


@pytest.mark.parametrize("mesh_shape,size", [((5, 7), (45, 80)), ((2, 2), (37, 53)), ((16, 16), (270, 480)), ((17, 9), (33, 200)), ((3, 31), (64, 48))])
def test_mesh_map_matches_numpy_bilinear(oracle, mesh_shape, size):
    """WarpMesh::apply's dense map (WarpMesh.cpp:190-191: cv::resize of the CV_32FC2 offsets, then * frame size): the float INTER_LINEAR machinery
    (imgproc/resize.cpp: fx = (float)((dx + 0.5) * scale - 0.5), taps (sx, sx + 1) with weights (1 - fx, fx), S[sx] * 1 past the last column,
    rows clipped individually) restated with whole-array float32 arithmetic in the same order -- bit-identical to the oracle."""
    f32 = np.float32
    rng = np.random.default_rng(11)
    mesh = synth.random_mesh(mesh_shape[0], mesh_shape[1], rng)
    rows, cols = size
    m = oracle.mesh_to_map(mesh, rows, cols)

    def axis(msize, fsize, vertical):
        scale = 1.0 / (fsize / msize)
        f = ((np.arange(fsize) + 0.5) * scale - 0.5).astype(f32)
        s = np.floor(f).astype(np.int64); f = (f - s.astype(f32)).astype(f32)
        if vertical:
            return np.clip(s, 0, msize - 1), np.clip(s + 1, 0, msize - 1), (f32(1) - f).astype(f32), f
        lo = s < 0; f[lo] = 0; s[lo] = 0
        single = s + 1 >= msize
        s = np.minimum(s, msize - 1)
        return s, np.where(single, s, s + 1), np.where(single, f32(1), f32(1) - f).astype(f32), np.where(single, f32(0), f).astype(f32), single
    x0, x1, a0, a1, single = axis(mesh_shape[1], cols, False)
    y0, y1, b0, b1 = axis(mesh_shape[0], rows, True)
    M = mesh.astype(f32)

    def hresize(rows_of):
        two = (rows_of[:, x0] * a0[None, :, None]).astype(f32) + (rows_of[:, x1] * a1[None, :, None]).astype(f32)
        one = (rows_of[:, x0] * f32(1)).astype(f32)
        return np.where(single[None, :, None], one, two.astype(f32)).astype(f32)
    h0, h1 = hresize(M[y0]), hresize(M[y1])
    v = ((h0 * b0[:, None, None]).astype(f32) + (h1 * b1[:, None, None]).astype(f32)).astype(f32)
    want = (v * np.array([cols, rows], f32)).astype(f32)
    assert np.array_equal(m, want)


def test_oracle_mesh_remap_equals_map_remap(oracle):
    """In-kernel mesh interpolation == materialised offset map pushed through easu_remap (FSR.cl:362-403)."""
    rng = np.random.default_rng(5)
    src = synth.textured_frame(64, 80, seed=6)
    mesh = synth.random_mesh(4, 4, rng, amp=0.03)
    a = oracle.remap_mesh(src, mesh, bg=(0, 128, 128), yuv=True)
    b = np_easu.remap_map(src, oracle.mesh_to_map(mesh, 64, 80), (0, 128, 128), True)
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 1 and (d == 0).mean() > 0.9995


def test_get_perspective_transform_known_affine(oracle):
    src = np.array([[0, 0], [100, 0], [0, 50], [100, 50]], np.float32)
    A = np.array([[1.1, 0.05, 3.0], [-0.02, 0.95, -4.0]])
    dst = (src @ A[:, :2].T + A[:, 2]).astype(np.float32)
    rc, M = oracle.get_perspective_transform(src, dst)
    assert rc == 0
    assert np.allclose(M[:2], A, atol=1e-5) and np.allclose(M[2], [0, 0, 1], atol=1e-7)


def test_get_perspective_transform_projective_roundtrip(oracle):
    rng = np.random.default_rng(3)
    src = np.array([[0, 0], [640, 0], [0, 360], [640, 360]], np.float32)
    dst = src + rng.uniform(-20, 20, src.shape).astype(np.float32)
    rc, M = oracle.get_perspective_transform(src, dst)
    p = np.c_[src, np.ones(4)] @ M.T
    assert np.allclose(p[:, :2] / p[:, 2:], dst, atol=1e-3)


def test_warpmesh_apply_2x2_identity_and_crop(oracle):
    src = synth.textured_frame(48, 64, seed=8)
    ident = np.zeros((2, 2, 2), np.float32)
    H = oracle.mesh2x2_to_homography(ident, 48, 64)
    assert np.allclose(H, np.eye(3), atol=1e-6)
    # scene crop of 10%: corners sample from 5% inside (WarpMesh::crop_in, WarpMesh.cpp:379-390)
    crop = np.array([[[0.05, 0.05], [-0.05, 0.05]], [[0.05, -0.05], [-0.05, -0.05]]], np.float32)
    H = oracle.mesh2x2_to_homography(crop, 48, 64)
    p = H @ np.array([0, 0, 1.0]); assert np.allclose(p[:2] / p[2], [3.2, 2.4], atol=1e-3)
    p = H @ np.array([64, 48, 1.0]); assert np.allclose(p[:2] / p[2], [60.8, 45.6], atol=1e-3)
    out = oracle.warpmesh_apply(src, crop, yuv=True)
    assert out.shape == src.shape


def test_get_perspective_transform_matches_float64_solve(oracle):
    """cv::getPerspectiveTransform (WarpMesh.cpp:214 for 2 x 2 meshes): the 8 x 8 system of the four correspondences, solved independently with
    numpy in binary64 -- the oracle's LU agrees to 1e-9 relative over random quads (incl. strongly projective ones)."""
    rng = np.random.default_rng(12)
    for trial in range(40):
        src = np.array([[0, 0], [640, 0], [0, 360], [640, 360]], np.float32) + rng.uniform(-30, 30, (4, 2)).astype(np.float32)
        dst = src + rng.uniform(-60, 60, (4, 2)).astype(np.float32) * (1 + trial % 3)
        rc, M = oracle.get_perspective_transform(src, dst)
        A = np.zeros((8, 8)); b = np.zeros(8)
        for i, ((x, y), (X, Y)) in enumerate(zip(src.astype(np.float64), dst.astype(np.float64))):
            A[i] = [x, y, 1, 0, 0, 0, -x * X, -y * X]; b[i] = X
            A[i + 4] = [0, 0, 0, x, y, 1, -x * Y, -y * Y]; b[i + 4] = Y
        want = np.append(np.linalg.solve(A, b), 1.0).reshape(3, 3)
        assert np.abs(M - want).max() <= 1e-9 * np.abs(want).max(), trial
