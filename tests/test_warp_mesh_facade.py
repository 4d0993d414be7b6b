"""lvk::Homography / lvk::WarpMesh / lvk::remap of the C++ facade (include/lvk/WarpMesh.hpp; SURVEY.md section 8 rows a11, a14, a15 at the API the plugin
links against -- LCFilter holds a WarpMesh: set_to(map) -> crop_in -> apply, LCFilter.cpp:133-192).

CPU: the host arithmetic (tests/cpp/warp_mesh_facade.cpp prints its arrays as hexadecimal floats) against an independent numpy restatement of the cited
reference lines in binary32 / binary64, exact.  GPU: apply() on a 2 x 2, a 16 x 16 and a frame-sized mesh, remap(homography, inverted / not) and
remap(offset map) against the oracle's WarpMesh::apply / lvk::remap, bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "warp_mesh_facade.cpp")
f32 = np.float32
KH = np.array([1.01, 0.012, 3.1, -0.011, 0.995, -2.2, 2e-5, -1e-5, 1.0])


def _build(tmp_path):
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    exe = str(tmp_path / "warp_mesh_facade")
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", exe, SRC,
                           "-L" + os.path.join(ROOT, "livevisionkit_amd"), "-llvk_hip", "-L" + tlib, "-l:libamdhip64.so",
                           "-Wl,-rpath," + os.path.join(ROOT, "livevisionkit_amd"), "-Wl,-rpath," + tlib])
    return exe


# ---- numpy restatement (binary32 unless stated; one rounding per operation, no contraction) -------------------------------------------------------
def from_homography(H, sw, sh, rows, cols):
    """WarpMesh::set_to(Homography, motion_scale) (WarpMesh.cpp:333-342) with Homography::transform -> cv::perspectiveTransform (binary64 inside)."""
    sw, sh = f32(sw), f32(sh)
    gx, gy = sw / f32(cols - 1), sh / f32(rows - 1)
    nx, ny = f32(1) / sw, f32(1) / sh
    out = np.zeros((rows, cols, 2), f32)
    for r in range(rows):
        for c in range(cols):
            px, py = f32(c) * gx, f32(r) * gy
            w = np.float64(px) * H[6] + np.float64(py) * H[7] + H[8]
            qx = qy = f32(0)
            if abs(w) > 1.1920928955078125e-07:
                w = 1.0 / w
                qx = f32((np.float64(px) * H[0] + np.float64(py) * H[1] + H[2]) * w)
                qy = f32((np.float64(px) * H[3] + np.float64(py) * H[4] + H[5]) * w)
            out[r, c] = ((px - qx) * nx, (py - qy) * ny)
    return out


def crop_in(m, x, y, w, h):
    rows, cols = m.shape[:2]
    kx, ky = (f32(w) - f32(1)) / f32(cols - 1), (f32(h) - f32(1)) / f32(rows - 1)
    cc, rr = np.meshgrid(np.arange(cols, dtype=f32), np.arange(rows, dtype=f32))
    out = m.copy()
    out[..., 0] = m[..., 0] + (cc * kx + f32(x))
    out[..., 1] = m[..., 1] + (rr * ky + f32(y))
    return out


def pixel_map(rows, cols):
    cc, rr = np.meshgrid(np.arange(cols, dtype=f32), np.arange(rows, dtype=f32))
    u, v = cc / f32(cols) - f32(0.5), rr / f32(rows) - f32(0.5)
    q = u * u + v * v
    return np.stack([cc + (f32(6) * u) * q, rr + (f32(4) * v) * q], -1).astype(f32)


def invert3(H):
    """cv::invert, 3 x 3 closed form (binary64)."""
    s = np.asarray(H, np.float64)
    d = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) + s[2] * (s[3] * s[7] - s[4] * s[6])
    if d == 0:
        return np.zeros(9)
    d = 1.0 / d
    return np.array([(s[4] * s[8] - s[5] * s[7]) * d, (s[2] * s[7] - s[1] * s[8]) * d, (s[1] * s[5] - s[2] * s[4]) * d,
                     (s[5] * s[6] - s[3] * s[8]) * d, (s[0] * s[8] - s[2] * s[6]) * d, (s[2] * s[3] - s[0] * s[5]) * d,
                     (s[3] * s[7] - s[4] * s[6]) * d, (s[1] * s[6] - s[0] * s[7]) * d, (s[0] * s[4] - s[1] * s[3]) * d])


def mesh16():
    cc, rr = np.meshgrid(np.arange(16), np.arange(16))
    ox = f32(0.008) * (((cc * 7 + rr * 3) % 11).astype(f32) / f32(11) - f32(0.5))
    oy = f32(0.006) * (((cc * 5 + rr * 9) % 13).astype(f32) / f32(13) - f32(0.5))
    return np.stack([ox, oy], -1).astype(f32)


def full_size_mesh(rows, cols):
    """LCFilter's flow: set_to(pixel map, as_offsets = false, normalized = false) -> crop_in(view region)."""
    cc, rr = np.meshgrid(np.arange(cols, dtype=f32), np.arange(rows, dtype=f32))
    m = pixel_map(rows, cols)
    m[..., 0] -= cc; m[..., 1] -= rr
    m[..., 0] *= f32(1) / f32(cols); m[..., 1] *= f32(1) / f32(rows)
    return crop_in(m, 0.02, 0.03, 0.95, 0.94)


def test_host_arithmetic_matches_a_numpy_restatement(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.check_output([exe]).decode()
    assert "host part done" in out
    got = {}
    for line in out.splitlines():
        parts = line.split()
        if len(parts) > 1 and parts[0] not in ("host", "flags"):
            got[parts[0]] = np.array([float.fromhex(x) for x in parts[1:]])
    rows, cols = 4, 5

    def same(name, want):
        w = np.asarray(want, np.float64).reshape(-1)
        assert got[name].shape == w.shape and np.array_equal(got[name], w), (name, np.abs(got[name] - w).max())
    A = from_homography(KH, 480, 270, rows, cols); same("set_to_H", A)
    A = crop_in(A, 0.05, 0.04, 0.9, 0.92); same("crop_in", A)
    A[..., 0] = np.clip(A[..., 0], f32(-0.06), f32(0.06)); A[..., 1] = np.clip(A[..., 1], f32(-0.05), f32(0.05)); same("clamp", A)
    cc, rr = np.meshgrid(np.arange(cols), np.arange(rows))
    B = np.stack([f32(0.001) * cc.astype(f32) - f32(0.002) * rr.astype(f32), f32(0.0005) * (cc * rr).astype(f32)], -1).astype(f32); same("write", B)
    A = B * f32(0.35) + A; same("combine", A)
    A = A + B
    A = A - np.array([0.01, -0.02], f32)
    A = A * f32(0.7)
    A = A * np.array([1.5, 0.5], f32)
    A = A / np.array([3.0, 7.0], f32)
    A = A / f32(1.3)
    A = A - B
    A = A + np.array([0.25, 0.125], f32); same("operators", A)
    kx, ky = ((f32(1) / f32(0.9)) - f32(1)) / f32(cols - 1), ((f32(1) / f32(0.8)) - f32(1)) / f32(rows - 1)
    A = A.copy(); A[..., 0] += cc.astype(f32) * kx; A[..., 1] += rr.astype(f32) * ky; same("scale", A)
    A[..., 0] = np.clip(A[..., 0], f32(-0.1), f32(0.3)); A[..., 1] = np.clip(A[..., 1], f32(0.0), f32(0.2)); same("clamp2", A)
    P = np.broadcast_to(np.array([-0.03, 0.01], f32), (rows, cols, 2)) * B; same("set_to_point_times", P)
    s0 = s1 = f32(0)
    for r in range(rows):
        for c in range(cols):
            s0 = s0 + A[r, c, 0] * f32(c + 1); s1 = s1 + A[r, c, 1] * f32(r + 1)
    same("read", [s0, s1])
    pm = pixel_map(rows, cols)
    M = pm.copy(); M[..., 0] -= cc.astype(f32); M[..., 1] -= rr.astype(f32); M[..., 0] *= f32(1) / f32(cols); M[..., 1] *= f32(1) / f32(rows); same("set_to_map", M)
    back = M.copy(); back[..., 0] += cc.astype(f32); back[..., 1] += rr.astype(f32); same("to_map", back)
    inv = invert3(KH); same("H_invert", inv)
    a, b = KH.reshape(3, 3), inv.reshape(3, 3)
    prod = np.array([[(a[i, 0] * b[0, j] + a[i, 1] * b[1, j]) + a[i, 2] * b[2, j] for j in range(3)] for i in range(3)]); same("H_product", prod)
    assert np.abs(prod - np.eye(3)).max() < 1e-12                      # and it IS the inverse
    same("H_ops", (KH + inv) * 0.5 - np.eye(3).reshape(-1) / 4.0)
    assert "flags 1 0 1 1 0" in out
    x, y = np.float64(f32(123.25)), np.float64(f32(77.5))
    w = 1.0 / (x * KH[6] + y * KH[7] + KH[8])
    same("transform_d", [(x * KH[0] + y * KH[1] + KH[2]) * w, (x * KH[3] + y * KH[4] + KH[5]) * w])
    same("transform_f", [f32((x * KH[0] + y * KH[1] + KH[2]) * w), f32((x * KH[3] + y * KH[4] + KH[5]) * w)])


@pytest.mark.gpu
def test_apply_and_remap_match_the_oracle(tmp_path, oracle):
    from tests import synth
    rows, cols = 270, 480
    src = synth.textured_frame(rows, cols, seed=11)
    (tmp_path / "in.bin").write_bytes(struct.pack("<ii", rows, cols) + src.tobytes())
    exe = _build(tmp_path)
    out = subprocess.check_output([exe, "--gpu", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], timeout=300).decode()
    assert "gpu part done: 7 frames" in out, out
    got = np.frombuffer((tmp_path / "out.bin").read_bytes(), np.uint8).reshape(7, rows, cols, 3)
    bg = (105, 212, 235)
    # (a) 2 x 2 mesh of the homography -> getPerspectiveTransform + easu_remap_homography; (b) 16 x 16 -> interpolated offsets + easu_remap
    assert np.array_equal(got[0], oracle.warpmesh_apply(src, from_homography(KH, cols, rows, 2, 2), bg=bg, yuv=True))
    assert np.array_equal(got[1], oracle.warpmesh_apply(src, mesh16(), bg=bg, yuv=True))
    # (c) a mesh of the frame's own size (LCFilter): the resize to the frame is the identity, the offsets x (W, H) are the map
    m = full_size_mesh(rows, cols)
    px = m * np.array([cols, rows], f32)
    assert np.array_equal(got[2], oracle.remap_map(src, px, bg=bg, yuv=True))
    # ... which is also what the oracle's WarpMesh::apply makes of that mesh (cv::resize INTER_LINEAR_EXACT to the same size + multiply)
    assert np.array_equal(got[2], oracle.warpmesh_apply(src, m, bg=bg, yuv=True))
    # (d) remap(homography): given inverted (dst -> src), and to be inverted by the launcher
    assert np.array_equal(got[3], oracle.remap_homography(src, KH.astype(f32).reshape(3, 3), bg=bg, yuv=True))
    assert np.array_equal(got[4], oracle.remap_homography(src, invert3(KH).astype(f32).reshape(3, 3), bg=bg, yuv=True))
    assert not np.array_equal(got[3], got[4])
    # (e) remap(offset map)
    cc, rr = np.meshgrid(np.arange(cols, dtype=f32), np.arange(rows, dtype=f32))
    offs = pixel_map(rows, cols); offs[..., 0] -= cc; offs[..., 1] -= rr
    assert np.array_equal(got[5], oracle.remap_map(src, offs, bg=bg, yuv=True))
    # (f) lvk::draw_grid + lvk::draw_crosses (Drawing.tpp:53-93,146-196) with the reference's colour constants
    want = oracle.draw_grid(src, (8, 5), (105, 212, 234), 1)
    want = oracle.draw_crosses(want, [(10.5, 7.25), (100.0, 60.0), (239.6, 134.4), (0.0, 0.0), (130.2, 20.9)], (149, 43, 21), 7, 4, scaling=(2.0, 2.0))
    assert np.array_equal(got[6], want) and not np.array_equal(got[6], src)
