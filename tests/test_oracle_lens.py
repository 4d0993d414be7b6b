"""CPU tests pinning the oracle's lens-correction map (no GPU).  The reference holds no
golden vectors for LCFilter and OpenCV is absent from the image, so this row is pinned by known-answer cases and a second,
independently written numpy statement of the undistort-rectify model (parity vs OpenCV itself: unpinned)."""
import numpy as np

from tests import synth


def _np_map(params, rows, cols):
    """Vectorised binary64 statement of getOptimalNewCameraMatrix(alpha 0) + initUndistortRectifyMap, written from the
    published Brown-Conrady model, not from oracle/lens.cpp: returns (map_x, map_y, P)."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = [float(v) for v in params]

    def undist(u, v):
        x0 = (u - cx) / fx; y0 = (v - cy) / fy
        x, y = x0.copy(), y0.copy()
        for _ in range(5):
            r2 = x * x + y * y
            ic = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x = (x0 - dx) * ic; y = (y0 - dy) * ic
        return x, y

    gx, gy = np.meshgrid(np.arange(9) * cols / 8.0, np.arange(9) * rows / 8.0)
    ux, uy = undist(gx.astype(np.float32).astype(np.float64), gy.astype(np.float32).astype(np.float64))
    ux = ux.astype(np.float32); uy = uy.astype(np.float32)
    ix0, ix1 = ux[:, 0].max(), ux[:, 8].min(); iy0, iy1 = uy[0, :].max(), uy[8, :].min()
    nfx = (cols - 1) / float(ix1 - ix0); nfy = (rows - 1) / float(iy1 - iy0)
    ncx = -nfx * float(ix0); ncy = -nfy * float(iy0)
    jj, ii = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    x = (jj - ncx) / nfx; y = (ii - ncy) / nfy
    r2 = x * x + y * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return fx * xd + cx, fy * yd + cy, (nfx, nfy, ncx, ncy)


def _offsets_from_map(mx, my, view, rows, cols):
    """WarpMesh set_to(absolute map) -> normalise -> crop_in(view) -> pixels, in binary64."""
    jj, ii = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    vx, vy, vw, vh = [float(v) for v in view]
    ox = (mx - jj) / cols + (jj * ((vw / cols - 1) / (cols - 1)) + vx / cols)
    oy = (my - ii) / rows + (ii * ((vh / rows - 1) / (rows - 1)) + vy / rows)
    return np.stack([ox * cols, oy * rows], axis=2)


def test_zero_distortion_is_a_near_identity_map(oracle):
    rows, cols = 270, 480
    off, view = oracle.lens_offset_map((400, 400, 239.5, 134.5, 0, 0, 0, 0, 0), rows, cols)
    assert view[0] == 0 and view[1] == 0 and view[2] >= cols - 1 and view[3] >= rows - 1
    # the [0, cols] x [0, rows] sample grid of getOptimalNewCameraMatrix leaves a (cols-1)/cols zoom: <= 1 px at the far edge
    assert np.abs(off).max() <= 1.01
    assert np.abs(off[0, 0]).max() < 1e-3


def test_matches_independent_numpy_model(oracle):
    rows, cols = 135, 240
    for params in [(0.8 * cols, 0.8 * cols, cols / 2, rows / 2, -0.12, 0.03, 0, 0, 0),
                   (0.9 * cols, 0.85 * cols, cols / 2 + 3, rows / 2 - 2, -0.2, 0.05, 1e-3, -2e-3, 0.01),
                   (1.1 * cols, 1.1 * cols, cols / 2, rows / 2, 0.08, -0.01, 0, 0, 0)]:
        off, view = oracle.lens_offset_map(params, rows, cols)
        mx, my, _ = _np_map(params, rows, cols)
        want = _offsets_from_map(mx, my, view, rows, cols)
        assert np.abs(off - want).max() < 2e-3, params
        assert 0 <= view[0] and 0 <= view[1] and view[0] + view[2] <= cols and view[1] + view[3] <= rows
        assert view[2] >= cols - 2 and view[3] >= rows - 2        # alpha = 0: the whole output is valid


def test_barrel_correction_pulls_corners_inwards(oracle):
    rows, cols = 270, 480
    off, _ = oracle.lens_offset_map((0.8 * cols, 0.8 * cols, cols / 2, rows / 2, -0.12, 0.03, 0, 0, 0), rows, cols)
    assert off[0, 0, 0] > 1 and off[0, 0, 1] > 1 and off[-1, -1, 0] < -1 and off[-1, -1, 1] < -1
    assert np.abs(off[rows // 2, cols // 2]).max() < 0.1


def test_remap_map_zero_offsets_equals_identity_homography(oracle):
    src = synth.textured_frame(60, 84, seed=4)
    zero = np.zeros((60, 84, 2), np.float32)
    for yuv in (True, False):
        assert np.array_equal(oracle.remap_map(src, zero, yuv=yuv), oracle.remap_homography(src, np.eye(3, dtype=np.float32), yuv=yuv))


def test_remap_map_matches_mesh_path_on_its_materialised_map(oracle):
    """A mesh warp rendered through its per-pixel map (Image.cpp:28-81 path) equals remap_mesh when the map is the oracle's
    own mesh_to_map (same binary32 interpolation)."""
    rows, cols = 72, 96
    src = synth.textured_frame(rows, cols, seed=9)
    rng = np.random.default_rng(5)
    mesh = synth.random_mesh(5, 7, rng)
    m = oracle.mesh_to_map(mesh, rows, cols)
    assert np.array_equal(oracle.remap_map(src, m), oracle.remap_mesh(src, mesh))


# ---- fused lens mode (closed-form map composed with the stabilizing warp; this repo's design for BASELINE config 5) ----------
LENS = lambda r, c: (0.8 * c, 0.8 * c, c / 2, r / 2, -0.12, 0.03, 0, 0, 0)


def _grid(rows, cols):
    jj, ii = np.meshgrid(np.arange(cols, dtype=np.float32), np.arange(rows, dtype=np.float32))
    return np.stack([jj, ii], axis=2)


def test_undistort_points_inverts_the_offset_map(oracle):
    rows, cols = 135, 240
    for params in [LENS(rows, cols), (0.9 * cols, 0.85 * cols, cols / 2 + 3, rows / 2 - 2, -0.2, 0.05, 1e-3, -2e-3, 0.01)]:
        off, _ = oracle.lens_offset_map(params, rows, cols)
        g = _grid(rows, cols)
        raw = (g + off).reshape(-1, 2)                      # F(corrected pixel) by the reference-restated map
        back = oracle.lens_undistort_points(params, rows, cols, 1.0, 1.0, raw).reshape(rows, cols, 2)
        assert np.abs(back - g).max() < 5e-3
        # tracking-resolution points: scale, invert, scale back
        half = oracle.lens_undistort_points(params, rows, cols, 2.0, 2.0, raw / 2).reshape(rows, cols, 2)
        assert np.abs(half * 2 - g).max() < 5e-3


def test_fused_identity_equals_map_remap_up_to_coordinate_rounding(oracle):
    """Identity stabilizing warp: the closed-form binary32 coordinate and the binary64-built offset map differ by ~1e-4 px,
    so the two EASU renders agree except for isolated +-few-LSB pixels."""
    rows, cols = 135, 240
    src = synth.textured_frame(rows, cols, seed=3)
    params = LENS(rows, cols)
    off, _ = oracle.lens_offset_map(params, rows, cols)
    a = oracle.remap_map(src, off, bg=(0, 0, 0))
    b = oracle.warpmesh_apply_lens(src, np.zeros((2, 2, 2), np.float32), params, bg=(0, 0, 0))
    assert synth.psnr(a, b) > 55.0
    assert np.mean(a != b) < 0.02


def test_fused_chain_beats_two_pass_against_the_ideal_render(oracle):
    """raw = ideal seen through the lens.  Reference chain: raw -LC-> corrected -warp-> out (two EASU resamplings).
    Fused: raw -> out in one.  Both are compared with the warp applied to the ideal frame itself."""
    rows, cols = 180, 320
    ideal = synth.textured_frame(rows, cols, seed=8)
    params = LENS(rows, cols)
    corrected_of_raw = oracle.lens_undistort_points(params, rows, cols, 1.0, 1.0, _grid(rows, cols).reshape(-1, 2)).reshape(rows, cols, 2)
    raw = synth.lens_distort(ideal, corrected_of_raw)
    off, _ = oracle.lens_offset_map(params, rows, cols)
    rng = np.random.default_rng(3)
    for mesh in (rng.uniform(-0.01, 0.01, (2, 2, 2)).astype(np.float32), synth.random_mesh(9, 9, rng, amp=0.006)):
        want = oracle.warpmesh_apply(ideal, mesh, bg=(0, 0, 0))
        two = oracle.warpmesh_apply(oracle.remap_map(raw, off, bg=(0, 0, 0)), mesh, bg=(0, 0, 0))
        one = oracle.warpmesh_apply_lens(raw, mesh, params, bg=(0, 0, 0))
        m = 12                                                  # ignore the border band / background
        p_two = synth.psnr(two[m:-m, m:-m], want[m:-m, m:-m]); p_one = synth.psnr(one[m:-m, m:-m], want[m:-m, m:-m])
        assert synth.psnr(one[m:-m, m:-m], two[m:-m, m:-m]) > 28.0      # same picture
        assert p_one >= p_two - 0.1, (p_one, p_two)             # BASELINE: within 0.1 dB of the reference chain (it is better)


def test_fused_stabilizer_tracks_like_the_two_pass_chain(oracle):
    """End to end on a lens-distorted shaky clip: the fused filter's per-frame motion meshes stay close to those of the
    reference chain (LC remap, then the plain filter), and its output is the same picture."""
    from tests import oracle_lib
    rows, cols, n = 270, 480, 12        # (at 180 x 320 this test used to pass on two filters that tracked nothing: the oracle refused to enlarge the tracking frame)
    frames, _ = synth.make_clip(rows, cols, n, seed=5)
    params = LENS(rows, cols)
    corrected_of_raw = oracle.lens_undistort_points(params, rows, cols, 1.0, 1.0, _grid(rows, cols).reshape(-1, 2)).reshape(rows, cols, 2)
    raw = synth.lens_distort(frames, corrected_of_raw)
    off, _ = oracle.lens_offset_map(params, rows, cols)
    s = oracle_lib.preset("homography", predictive_samples=3)
    two = oracle_lib.OracleStabilizer(oracle, s); one = oracle_lib.OracleStabilizer(oracle, s)
    one.set_lens(params)                 # (restarts the filter: m_SceneQuality = 1, StabilizationFilter.cpp:138-143)
    two.restart()                        # the same quality-assurance state for the chain it is compared with
    produced = moving = 0
    for i in range(n):
        a, _ = two.push(oracle.remap_map(raw[i], off, bg=(0, 0, 0)), ts=i)
        b, _ = one.push(raw[i], ts=i)
        assert (a is None) == (b is None)
        ma, _ = two.meshes(); mb, _ = one.meshes()
        # (a frame on which one of the two trackers falls below min_tracking_quality has its motion zeroed by the quality assurance: compared
        #  are the frames on which both delivered one)
        both = bool(np.abs(ma).max() > 0 and np.abs(mb).max() > 0)
        moving += both
        if both:
            assert np.abs(ma - mb).max() < 2e-3, i              # normalised offsets: < 1 px at this width
        if a is not None:
            produced += 1
            assert synth.psnr(a[16:-16, 16:-16], b[16:-16, 16:-16]) > 27.0, i
    assert produced == n - 3 and moving >= 4                 # both filters really tracked (the trust factor leaves zero after ~7 frames)
    two.close(); one.close()


def test_fused_field_preset_converges(oracle):
    """Lens-corrected positions of border features leave the tracking region; they must be dropped (not handed to the mesh
    solver, which rejects the whole frame for an out-of-grid sample)."""
    from tests import oracle_lib
    rows, cols, n = 270, 480, 10
    frames, _ = synth.make_clip(rows, cols, n, seed=6)
    params = LENS(rows, cols)
    cr = oracle.lens_undistort_points(params, rows, cols, 1.0, 1.0, _grid(rows, cols).reshape(-1, 2)).reshape(rows, cols, 2)
    raw = synth.lens_distort(frames, cr)
    st = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default"))
    st.configure(oracle_lib.preset("field", predictive_samples=3))
    st.set_lens(params)
    stab = []
    for i in range(n):
        st.push(raw[i], ts=i)
        stab.append(st.stats().tracking_stability)
    st.close()
    assert min(stab[4:]) > 0.8, stab
