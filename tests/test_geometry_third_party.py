"""Third-party pins of three more author-restated OpenCV stages (round-5 VERDICT weak #7), with what this image offers (numpy / scipy):

 * cv::getPerspectiveTransform (WarpMesh.cpp:214): the oracle's 3 x 3 against numpy.linalg.solve of the 8 x 8 direct-linear-transform
   system and against the defining property (the four source points land on the four destination points);
 * the dense offset map of WarpMesh::apply (WarpMesh.cpp:190: cv::resize of the CV_32FC2 mesh to the frame size, INTER_LINEAR_EXACT -> the float
   bilinear machinery, App. A.7): sampling geometry and weights against scipy.ndimage.zoom(order=1, grid_mode=True, mode="nearest");
 * cv::cvtColor(BGR2GRAY) on 8U (VideoFrame.cpp:194): the 15-bit fixed point against the ITU-R BT.601 weights in binary64."""
import numpy as np
import pytest
from scipy import ndimage

from tests import synth


@pytest.mark.parametrize("seed", range(8))
def test_get_perspective_transform_equals_the_dlt_solution(oracle, seed):
    rng = np.random.default_rng(seed)
    w, h = 1920.0, 1080.0
    src = np.array([[0, 0], [w, 0], [0, h], [w, h]], np.float32)
    dst = (src + rng.uniform(-0.05, 0.05, (4, 2)) * (w, h)).astype(np.float32)
    rc, M = oracle.get_perspective_transform(src, dst)
    assert rc == 0
    # the 8 x 8 system of the published algorithm, solved by LAPACK in binary64
    A = np.zeros((8, 8)); b = np.zeros(8)
    for i, ((x, y), (u, v)) in enumerate(zip(src.astype(np.float64), dst.astype(np.float64))):
        A[i] = (x, y, 1, 0, 0, 0, -x * u, -y * u); b[i] = u
        A[i + 4] = (0, 0, 0, x, y, 1, -x * v, -y * v); b[i + 4] = v
    H = np.append(np.linalg.solve(A, b), 1.0).reshape(3, 3)
    assert np.abs(M - H).max() <= 1e-9 * np.abs(H).max()
    p = np.c_[src.astype(np.float64), np.ones(4)] @ M.T
    assert np.abs(p[:, :2] / p[:, 2:] - dst).max() < 1e-8


def test_get_perspective_transform_of_a_degenerate_quad_is_refused_or_finite(oracle):
    src = np.array([[0, 0], [10, 0], [0, 10], [10, 10]], np.float32)
    dst = np.array([[0, 0], [5, 5], [10, 10], [15, 15]], np.float32)           # collinear: the system is singular
    rc, M = oracle.get_perspective_transform(src, dst)
    assert rc != 0 or not np.isfinite(M).all() or np.abs(np.linalg.det(M)) < 1e-6


@pytest.mark.parametrize("mesh_shape,size", [((16, 16), (270, 480)), ((5, 7), (45, 80)), ((2, 2), (36, 64)), ((9, 17), (200, 33))])
def test_mesh_map_geometry_equals_scipy_zoom(oracle, mesh_shape, size):
    """offset map = resize(mesh offsets) * (cols, rows): pixel centres at half-integers, edge samples replicated, linear weights -- scipy's grid-mode zoom
    computes the same samples in binary64; the oracle's binary32 result sits within a few ulp of the offsets' magnitude."""
    rng = np.random.default_rng(mesh_shape[0] * 31 + size[1])
    mesh = synth.random_mesh(mesh_shape[0], mesh_shape[1], rng)
    rows, cols = size
    m = oracle.mesh_to_map(mesh, rows, cols)
    for ch, scale in ((0, cols), (1, rows)):
        z = ndimage.zoom(mesh[..., ch].astype(np.float64), (rows / mesh_shape[0], cols / mesh_shape[1]), order=1, mode="nearest", grid_mode=True) * scale
        assert z.shape == (rows, cols)
        assert np.abs(m[..., ch] - z).max() < 4e-6 * max(1.0, np.abs(z).max())


def test_bgr_to_gray_equals_bt601_within_half_a_level(oracle):
    rng = np.random.default_rng(5)
    px = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    gray = oracle.luma_area_resize(px, 64, 96, channel=-1)                       # channel -1: cvtColor(BGR2GRAY), no resampling at equal sizes
    exact = px[..., 0] * 0.114 + px[..., 1] * 0.587 + px[..., 2] * 0.299
    d = gray.astype(np.float64) - exact
    assert np.abs(d).max() <= 0.5 + 0.012                                        # round to nearest; the 15-bit coefficients are off by <= 2e-5 each (x 255 x 3)
    assert (gray == np.floor(exact + 0.5)).mean() > 0.995
