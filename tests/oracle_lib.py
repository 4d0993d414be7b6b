"""ctypes wrapper around the CPU oracle (oracle/liblvk_oracle.so). TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liblvk_oracle.so")

_c = ctypes
_u8p = _c.POINTER(_c.c_uint8)
_f32p = _c.POINTER(_c.c_float)
_f64p = _c.POINTER(_c.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _p(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.lvko_remap_homography.restype = _c.c_int
        L.lvko_remap_homography.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int, _c.c_int, _c.c_int,
                                            _c.c_int, _c.c_int, _f32p, _u8p, _c.c_int, _c.c_int]
        L.lvko_remap_mesh.restype = _c.c_int
        L.lvko_remap_mesh.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int, _f32p, _c.c_int, _c.c_int,
                                      _u8p, _c.c_int, _c.c_int]
        L.lvko_mesh_to_map.restype = None
        L.lvko_mesh_to_map.argtypes = [_f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _f32p]
        L.lvko_get_perspective_transform.restype = _c.c_int
        L.lvko_get_perspective_transform.argtypes = [_f32p, _f32p, _f64p]
        L.lvko_warpmesh_apply.restype = _c.c_int
        L.lvko_warpmesh_apply.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int, _f32p, _c.c_int, _c.c_int,
                                          _u8p, _c.c_int, _c.c_int]
        L.lvko_mesh2x2_to_homography.restype = _c.c_int
        L.lvko_mesh2x2_to_homography.argtypes = [_f32p, _c.c_int, _c.c_int, _f32p]
        _i = _c.c_int
        _i16p = _c.POINTER(_c.c_int16); _i32p = _c.POINTER(_c.c_int32)
        L.lvko_luma_area_resize.restype = _i
        L.lvko_luma_area_resize.argtypes = [_u8p, _i, _i, _i, _i, _i, _u8p, _i, _i, _i]
        L.lvko_pyr_down.restype = _i
        L.lvko_pyr_down.argtypes = [_u8p, _i, _i, _i, _u8p, _i]
        L.lvko_scharr_deriv.restype = _i
        L.lvko_scharr_deriv.argtypes = [_u8p, _i, _i, _i, _i16p]
        L.lvko_fast9_16.restype = _i
        L.lvko_fast9_16.argtypes = [_u8p, _i, _i, _i, _i, _i, _i, _i32p, _i]
        L.lvko_pyrlk.restype = _i
        L.lvko_pyrlk.argtypes = [_u8p, _i, _u8p, _i, _i, _i, _f32p, _i, _f32p, _u8p, _i, _i, _i, _i, _c.c_double, _c.c_double]
        L.lvko_pyramid_levels.restype = _i
        L.lvko_pyramid_levels.argtypes = [_i, _i, _i, _i, _i, _i32p, _i32p]

    # ---- remap -----------------------------------------------------------------------------------
    def remap_homography(self, src, H, bg=(255, 0, 255), yuv=True, dst_size=None, offset=(0, 0), nthreads=8):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        rows, cols = src.shape[:2]
        drows, dcols = dst_size if dst_size is not None else (rows, cols)
        dst = np.zeros((drows, dcols, 3), np.uint8)
        H = np.ascontiguousarray(H, np.float32).reshape(9)
        bg = np.ascontiguousarray(bg, np.uint8)
        rc = self.lib.lvko_remap_homography(_p(src, _u8p), src.strides[0], rows, cols, _p(dst, _u8p), dst.strides[0],
                                            drows, dcols, offset[0], offset[1], _p(H, _f32p), _p(bg, _u8p),
                                            1 if yuv else 0, nthreads)
        assert rc == 0
        return dst

    def remap_mesh(self, src, mesh, bg=(255, 0, 255), yuv=True, nthreads=8):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        rows, cols = src.shape[:2]
        dst = np.zeros((rows, cols, 3), np.uint8)
        mesh = np.ascontiguousarray(mesh, np.float32)
        bg = np.ascontiguousarray(bg, np.uint8)
        rc = self.lib.lvko_remap_mesh(_p(src, _u8p), src.strides[0], rows, cols, _p(dst, _u8p), dst.strides[0],
                                      _p(mesh, _f32p), mesh.shape[0], mesh.shape[1], _p(bg, _u8p), 1 if yuv else 0, nthreads)
        assert rc == 0
        return dst

    def warpmesh_apply(self, src, mesh, bg=(255, 0, 255), yuv=True, nthreads=8):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        rows, cols = src.shape[:2]
        dst = np.zeros((rows, cols, 3), np.uint8)
        mesh = np.ascontiguousarray(mesh, np.float32)
        bg = np.ascontiguousarray(bg, np.uint8)
        rc = self.lib.lvko_warpmesh_apply(_p(src, _u8p), src.strides[0], rows, cols, _p(dst, _u8p), dst.strides[0],
                                          _p(mesh, _f32p), mesh.shape[0], mesh.shape[1], _p(bg, _u8p), 1 if yuv else 0, nthreads)
        assert rc == 0
        return dst

    # ---- tracker image ops ---------------------------------------------------------------------------
    def luma_area_resize(self, frame, drows, dcols, channel=0):
        """frame: [rows, cols, 3] packed or [rows, cols] planar uint8."""
        frame = np.ascontiguousarray(frame, np.uint8)
        pix = frame.shape[2] if frame.ndim == 3 else 1
        dst = np.zeros((drows, dcols), np.uint8)
        rc = self.lib.lvko_luma_area_resize(_p(frame, _u8p), frame.strides[0], pix, channel, frame.shape[0], frame.shape[1],
                                            _p(dst, _u8p), dst.strides[0], drows, dcols)
        assert rc == 0, rc
        return dst

    def pyr_down(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        dst = np.zeros(((img.shape[0] + 1) // 2, (img.shape[1] + 1) // 2), np.uint8)
        self.lib.lvko_pyr_down(_p(img, _u8p), img.strides[0], img.shape[0], img.shape[1], _p(dst, _u8p), dst.strides[0])
        return dst

    def scharr_deriv(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        dst = np.zeros(img.shape + (2,), np.int16)
        self.lib.lvko_scharr_deriv(_p(img, _u8p), img.strides[0], img.shape[0], img.shape[1], _p(dst, _c.POINTER(_c.c_int16)))
        return dst

    def pyramid_levels(self, rows, cols, max_level=3, win=(11, 11)):
        r = np.zeros(16, np.int32); c = np.zeros(16, np.int32)
        n = self.lib.lvko_pyramid_levels(rows, cols, max_level, win[0], win[1], _p(r, _c.POINTER(_c.c_int32)), _p(c, _c.POINTER(_c.c_int32)))
        return [(int(r[i]), int(c[i])) for i in range(n)]

    def fast(self, img, threshold, roi=None):
        """Returns an [n, 3] int32 array of (x, y, score), ROI-local, row-major."""
        img = np.ascontiguousarray(img, np.uint8)
        x, y, w, h = roi if roi is not None else (0, 0, img.shape[1], img.shape[0])
        cap = max(1, w * h)
        out = np.zeros((cap, 3), np.int32)
        n = self.lib.lvko_fast9_16(_p(img, _u8p), img.strides[0], x, y, w, h, threshold, _p(out, _c.POINTER(_c.c_int32)), cap)
        return out[:n].copy()

    def pyrlk(self, prev, nxt, pts, win=(11, 11), max_level=3, max_count=5, epsilon=0.01, min_eig=1e-4):
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        out = np.zeros_like(pts); st = np.zeros(len(pts), np.uint8)
        lv = self.lib.lvko_pyrlk(_p(prev, _u8p), prev.strides[0], _p(nxt, _u8p), nxt.strides[0], prev.shape[0], prev.shape[1],
                                 _p(pts, _f32p), len(pts), _p(out, _f32p), _p(st, _u8p), win[0], win[1], max_level, max_count,
                                 float(epsilon), float(min_eig))
        assert lv >= 0
        return out, st

    def mesh_to_map(self, mesh, rows, cols):
        mesh = np.ascontiguousarray(mesh, np.float32)
        out = np.zeros((rows, cols, 2), np.float32)
        self.lib.lvko_mesh_to_map(_p(mesh, _f32p), mesh.shape[0], mesh.shape[1], rows, cols, _p(out, _f32p))
        return out

    def get_perspective_transform(self, src, dst):
        src = np.ascontiguousarray(src, np.float32).reshape(8)
        dst = np.ascontiguousarray(dst, np.float32).reshape(8)
        M = np.zeros(9, np.float64)
        rc = self.lib.lvko_get_perspective_transform(_p(src, _f32p), _p(dst, _f32p), _p(M, _f64p))
        return rc, M.reshape(3, 3)

    def mesh2x2_to_homography(self, mesh, rows, cols):
        mesh = np.ascontiguousarray(mesh, np.float32).reshape(8)
        H = np.zeros(9, np.float32)
        self.lib.lvko_mesh2x2_to_homography(_p(mesh, _f32p), rows, cols, _p(H, _f32p))
        return H.reshape(3, 3)


_inst = None


def load():
    global _inst
    if _inst is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h"))]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
            build()
        _inst = Oracle(ctypes.CDLL(LIB))
    return _inst
