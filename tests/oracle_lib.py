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
        L.lvko_pyrlk_float.restype = _i
        L.lvko_pyrlk_float.argtypes = [_u8p, _i, _u8p, _i, _i, _i, _f32p, _i, _f32p, _u8p, _i, _i, _i, _i, _c.c_double, _c.c_double, _i, _i]
        L.lvko_pyramid_levels.argtypes = [_i, _i, _i, _i, _i, _i32p, _i32p]

    def set_num_threads(self, n):
        """Thread count of the oracle's row / point-parallel stages outside the stabilizer (results do not depend on it)."""
        self.lib.lvko_set_num_threads.restype = _c.c_int; self.lib.lvko_set_num_threads.argtypes = [_c.c_int]
        return self.lib.lvko_set_num_threads(int(n))

    def set_device_rcp(self, on=True):
        """native_recip of FSR.cl: the gfx950 table (default) or the correctly rounded reciprocal."""
        fn = self.lib.lvko_set_device_rcp_table
        fn.restype = _c.c_int
        fn.argtypes = [_f32p, _c.c_int]
        if on:
            tab = np.ascontiguousarray(device_rcp_table())
            assert fn(_p(tab, _f32p), len(tab)) == 0
        else:
            assert fn(None, 0) == 0

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

    def pyrlk_float(self, prev, nxt, pts, lanes, pairs, win=(11, 11), max_level=3, max_count=5, epsilon=0.01, min_eig=1e-4):
        """The tracker with OpenCV's binary32 window sums (lanes 1 / 4 / 8 / 16, pairs = v_dotprod pre-sums) instead of the specification's
        exact integer sums -- only to measure the distance between the two."""
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        out = np.zeros_like(pts); st = np.zeros(len(pts), np.uint8)
        lv = self.lib.lvko_pyrlk_float(_p(prev, _u8p), prev.strides[0], _p(nxt, _u8p), nxt.strides[0], prev.shape[0], prev.shape[1],
                                       _p(pts, _f32p), len(pts), _p(out, _f32p), _p(st, _u8p), win[0], win[1], max_level, max_count,
                                       float(epsilon), float(min_eig), int(lanes), int(pairs))
        assert lv >= 0
        return out, st

    def ingest_yuv420(self, y, u, v=None):
        """I420 (y, u, v planes) or NV12 (y, interleaved uv [r/2, c/2, 2]) -> packed [rows, cols, 3]."""
        y = np.ascontiguousarray(y, np.uint8); u = np.ascontiguousarray(u, np.uint8)
        nv12 = v is None
        vv = u if nv12 else np.ascontiguousarray(v, np.uint8)
        rows, cols = y.shape
        dst = np.zeros((rows, cols, 3), np.uint8)
        fn = self.lib.lvko_ingest_yuv420
        fn.restype = _c.c_int
        fn.argtypes = [_u8p, _c.c_int, _u8p, _c.c_int, _u8p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int]
        rc = fn(_p(y, _u8p), y.strides[0], _p(u, _u8p), u.strides[0], _p(vv, _u8p), vv.strides[0], 1 if nv12 else 0, rows, cols,
                _p(dst, _u8p), dst.strides[0])
        assert rc == 0
        return dst

    def egress_yuv420(self, frame, nv12=False):
        frame = np.ascontiguousarray(frame, np.uint8)
        rows, cols = frame.shape[:2]
        y = np.zeros((rows, cols), np.uint8)
        if nv12:
            u = np.zeros((rows // 2, cols // 2, 2), np.uint8); v = u
        else:
            u = np.zeros((rows // 2, cols // 2), np.uint8); v = np.zeros_like(u)
        fn = self.lib.lvko_egress_yuv420
        fn.restype = _c.c_int
        fn.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int, _u8p, _c.c_int, _u8p, _c.c_int, _c.c_int]
        rc = fn(_p(frame, _u8p), frame.strides[0], rows, cols, _p(y, _u8p), y.strides[0], _p(u, _u8p), u.strides[0],
                _p(v, _u8p), v.strides[0], 1 if nv12 else 0)
        assert rc == 0
        return (y, u) if nv12 else (y, u, v)

    # ---- every OBS video format of FrameIngest::Select (oracle/ingest.cpp::lvko_ingest_obs / lvko_egress_obs) ----
    VIDEO_FORMATS = {"I420": 1, "NV12": 2, "YVYU": 3, "YUY2": 4, "UYVY": 5, "RGBA": 6, "BGRA": 7, "BGRX": 8, "Y800": 9, "I444": 10, "BGR3": 11,
                     "I422": 12, "I40A": 13, "I42A": 14, "YUVA": 15, "AYUV": 16}

    @staticmethod
    def obs_plane_shapes(fmt, rows, cols):
        """shapes of the planes OBS holds for one frame of `fmt` (the alpha planes of I40A / I42A / YUVA are not touched by FrameIngest)"""
        if fmt in ("I420", "I40A"): return [(rows, cols), (rows // 2, cols // 2), (rows // 2, cols // 2)]
        if fmt == "NV12": return [(rows, cols), (rows // 2, cols // 2, 2)]
        if fmt in ("I422", "I42A"): return [(rows, cols), (rows, cols // 2), (rows, cols // 2)]
        if fmt in ("I444", "YUVA"): return [(rows, cols)] * 3
        if fmt in ("YUY2", "YVYU", "UYVY"): return [(rows, cols, 2)]
        if fmt in ("AYUV", "RGBA", "BGRA", "BGRX"): return [(rows, cols, 4)]
        if fmt == "BGR3": return [(rows, cols, 3)]
        if fmt == "Y800": return [(rows, cols)]
        raise ValueError(fmt)

    def _obs_args(self, planes):
        planes = [np.ascontiguousarray(p, np.uint8) for p in planes]
        ptrs = (_u8p * 3)(*[_p(p, _u8p) for p in planes] + [None] * (3 - len(planes)))
        steps = (_c.c_int * 3)(*[p.strides[0] for p in planes] + [0] * (3 - len(planes)))
        return planes, ptrs, steps

    def ingest_obs(self, fmt, planes):
        """OBS planes (numpy uint8, shapes of obs_plane_shapes) -> the VideoFrame FrameIngest::to_ocl makes: [rows, cols, 3] ([rows, cols] for Y800)."""
        planes, ptrs, steps = self._obs_args(planes)
        rows, cols = planes[0].shape[:2]
        dst = np.zeros((rows, cols) if fmt == "Y800" else (rows, cols, 3), np.uint8)
        fn = self.lib.lvko_ingest_obs
        fn.restype = _c.c_int
        fn.argtypes = [_c.c_int, _u8p * 3, _c.c_int * 3, _c.c_int, _c.c_int, _u8p, _c.c_int]
        rc = fn(self.VIDEO_FORMATS[fmt], ptrs, steps, rows, cols, _p(dst, _u8p), dst.strides[0])
        assert rc == 0, f"lvko_ingest_obs({fmt}) = {rc}"
        return dst

    def egress_obs(self, fmt, frame, planes=None):
        """FrameIngest::to_obs: the frame into OBS planes (`planes`: existing planes to write into -- bytes the reference leaves alone stay)."""
        frame = np.ascontiguousarray(frame, np.uint8)
        rows, cols = frame.shape[:2]
        if planes is None:
            planes = [np.zeros(sh, np.uint8) for sh in self.obs_plane_shapes(fmt, rows, cols)]
        planes, ptrs, steps = self._obs_args(planes)
        fn = self.lib.lvko_egress_obs
        fn.restype = _c.c_int
        fn.argtypes = [_c.c_int, _u8p, _c.c_int, _c.c_int, _c.c_int, _u8p * 3, _c.c_int * 3]
        rc = fn(self.VIDEO_FORMATS[fmt], _p(frame, _u8p), frame.strides[0], rows, cols, ptrs, steps)
        assert rc == 0, f"lvko_egress_obs({fmt}) = {rc}"
        return planes

    def lens_offset_map(self, params, rows, cols):
        """params = (fx, fy, cx, cy, k1, k2, p1, p2, k3). Returns (offsets [rows, cols, 2] float32 in pixels, view (x, y, w, h))."""
        pr = np.ascontiguousarray(params, np.float64).reshape(9)
        off = np.zeros((rows, cols, 2), np.float32); view = np.zeros(4, np.int32)
        fn = self.lib.lvko_lens_offset_map
        fn.restype = _c.c_int
        fn.argtypes = [_f64p, _c.c_int, _c.c_int, _f32p, _c.POINTER(_c.c_int32)]
        assert fn(_p(pr, _f64p), rows, cols, _p(off, _f32p), _p(view, _c.POINTER(_c.c_int32))) == 0
        return off, tuple(int(v) for v in view)

    def draw_grid(self, frame, grid, colour, thickness=1):
        frame = np.ascontiguousarray(frame, np.uint8).copy(); c = np.ascontiguousarray(colour, np.uint8)
        fn = self.lib.lvko_draw_grid
        fn.restype = _c.c_int; fn.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int]
        assert fn(_p(frame, _u8p), frame.strides[0], frame.shape[0], frame.shape[1], int(grid[0]), int(grid[1]), _p(c, _u8p), thickness) == 0
        return frame

    def draw_crosses(self, frame, points, colour, cross_size, thickness, scaling=(1.0, 1.0)):
        frame = np.ascontiguousarray(frame, np.uint8).copy(); c = np.ascontiguousarray(colour, np.uint8)
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        fn = self.lib.lvko_draw_crosses
        fn.restype = _c.c_int
        fn.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_int, _c.c_float, _c.c_float, _u8p, _c.c_int, _c.c_int]
        assert fn(_p(frame, _u8p), frame.strides[0], frame.shape[0], frame.shape[1], _p(p, _f32p), len(p), scaling[0], scaling[1], _p(c, _u8p),
                  cross_size, thickness) == 0
        return frame

    def lens_model(self, params, rows, cols):
        pr = np.ascontiguousarray(params, np.float64).reshape(9); m = np.zeros(17, np.float64)
        fn = self.lib.lvko_lens_model
        fn.restype = _c.c_int; fn.argtypes = [_f64p, _c.c_int, _c.c_int, _f64p]
        assert fn(_p(pr, _f64p), rows, cols, _p(m, _f64p)) == 0
        return m

    def lens_undistort_points(self, params, rows, cols, sx, sy, pts):
        m = self.lens_model(params, rows, cols)
        p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2); out = np.zeros_like(p)
        fn = self.lib.lvko_lens_undistort_points
        fn.restype = None; fn.argtypes = [_f64p, _c.c_double, _c.c_double, _f32p, _c.c_int, _f32p]
        fn(_p(m, _f64p), float(sx), float(sy), _p(p, _f32p), len(p), _p(out, _f32p))
        return out

    def warpmesh_apply_lens(self, src, mesh, params, bg=(255, 0, 255), yuv=True, nthreads=8):
        src = np.ascontiguousarray(src, np.uint8); mesh = np.ascontiguousarray(mesh, np.float32)
        rows, cols = src.shape[:2]
        m = self.lens_model(params, rows, cols)
        dst = np.zeros_like(src); bg = np.ascontiguousarray(bg, np.uint8)
        fn = self.lib.lvko_warpmesh_apply_lens
        fn.restype = _c.c_int
        fn.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int, _f32p, _c.c_int, _c.c_int, _u8p, _c.c_int, _c.c_int, _f64p]
        assert fn(_p(src, _u8p), src.strides[0], rows, cols, _p(dst, _u8p), dst.strides[0], _p(mesh, _f32p), mesh.shape[0], mesh.shape[1],
                  _p(bg, _u8p), 1 if yuv else 0, nthreads, _p(m, _f64p)) == 0
        return dst

    def remap_map(self, src, offsets, bg=(255, 0, 255), yuv=True, nthreads=8):
        src = np.ascontiguousarray(src, np.uint8); offsets = np.ascontiguousarray(offsets, np.float32)
        dst = np.zeros_like(src); bg = np.ascontiguousarray(bg, np.uint8)
        fn = self.lib.lvko_remap_map
        fn.restype = _c.c_int
        fn.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int, _f32p, _u8p, _c.c_int, _c.c_int]
        assert fn(_p(src, _u8p), src.strides[0], src.shape[0], src.shape[1], _p(dst, _u8p), dst.strides[0], _p(offsets, _f32p),
                  _p(bg, _u8p), 1 if yuv else 0, nthreads) == 0
        return dst

    def upscale(self, src, size, yuv=True, nthreads=8):
        """size = (width, height) like cv::Size."""
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros((int(size[1]), int(size[0]), 3), np.uint8)
        fn = self.lib.lvko_upscale
        fn.restype = _c.c_int
        fn.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]
        rc = fn(_p(src, _u8p), src.strides[0], src.shape[0], src.shape[1], _p(dst, _u8p), dst.strides[0], dst.shape[0], dst.shape[1],
                1 if yuv else 0, nthreads)
        assert rc == 0, rc
        return dst

    def sharpen(self, src, sharpness=0.7, nthreads=8):
        src = np.ascontiguousarray(src, np.uint8); dst = np.zeros_like(src)
        fn = self.lib.lvko_sharpen
        fn.restype = _c.c_int
        fn.argtypes = [_u8p, _c.c_int, _c.c_int, _c.c_int, _u8p, _c.c_int, _c.c_float, _c.c_int]
        rc = fn(_p(src, _u8p), src.strides[0], src.shape[0], src.shape[1], _p(dst, _u8p), dst.strides[0], float(sharpness), nthreads)
        assert rc == 0, rc
        return dst

    def find_homography(self, p1, p2, threshold, region=(480, 270), partial=False):
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        H = np.zeros(9, np.float64); mask = np.zeros(len(p1), np.uint8)
        fn = self.lib.lvko_estimate_affine_partial if partial else self.lib.lvko_find_homography
        fn.restype = _c.c_int
        fn.argtypes = [_f32p, _f32p, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _f64p, _u8p]
        rc = fn(_p(p1, _f32p), _p(p2, _f32p), len(p1), float(threshold), float(region[0]), float(region[1]), _p(H, _f64p), _p(mask, _u8p))
        return rc, H.reshape(3, 3), mask

    def usac_find_homography(self, p1, p2, threshold, max_thr=0.0, rng_state=0, final_lo=True):
        """Reference-semantics leg (oracle/usac_ref.cpp): cv::findHomography(UsacParams) as FrameTracker.cpp:337-357 configures it."""
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        H = np.zeros(9, np.float64); mask = np.zeros(len(p1), np.uint8); it = _c.c_int(0)
        fn = self.lib.lvko_usac_find_homography
        fn.restype = _c.c_int
        fn.argtypes = [_f32p, _f32p, _c.c_int, _c.c_double, _c.c_double, _c.c_uint, _c.c_int, _f64p, _u8p, _c.POINTER(_c.c_int)]
        rc = fn(_p(p1, _f32p), _p(p2, _f32p), len(p1), float(threshold), float(max_thr), int(rng_state), int(bool(final_lo)), _p(H, _f64p), _p(mask, _u8p), _c.byref(it))
        return rc, H.reshape(3, 3), mask, it.value

    def ref_estimate_affine_partial(self, p1, p2, threshold):
        """Reference-semantics leg: cv::estimateAffinePartial2D(RANSAC, thr, 50) + FromAffineMatrix (FrameTracker.cpp:364-371)."""
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        H = np.zeros(9, np.float64); mask = np.zeros(len(p1), np.uint8)
        fn = self.lib.lvko_ref_estimate_affine_partial
        fn.restype = _c.c_int
        fn.argtypes = [_f32p, _f32p, _c.c_int, _c.c_double, _f64p, _u8p]
        rc = fn(_p(p1, _f32p), _p(p2, _f32p), len(p1), float(threshold), _p(H, _f64p), _p(mask, _u8p))
        return rc, H.reshape(3, 3), mask

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


class StabSettings(_c.Structure):
    """Field-for-field mirror of lvko_stab_settings / lvk_stab_settings (flattened lvk::StabilizationFilterSettings)."""
    _fields_ = [("detection_width", _c.c_int), ("detection_height", _c.c_int),
                ("detection_regions_x", _c.c_int), ("detection_regions_y", _c.c_int), ("force_detection", _c.c_int),
                ("max_feature_density", _c.c_float), ("min_feature_density", _c.c_float), ("accumulation_rate", _c.c_float),
                ("track_local_motions", _c.c_int), ("temporal_smoothing", _c.c_float), ("local_smoothing", _c.c_float),
                ("min_motion_samples", _c.c_int), ("acceptance_threshold", _c.c_float), ("uniformity_threshold", _c.c_float),
                ("predictive_samples", _c.c_int), ("corrective_limit_x", _c.c_float), ("corrective_limit_y", _c.c_float),
                ("smoothing_steps", _c.c_float), ("response_rate", _c.c_float),
                ("motion_width", _c.c_int), ("motion_height", _c.c_int), ("background", _c.c_float * 3),
                ("crop_to_stable_region", _c.c_int), ("stabilize_output", _c.c_int),
                ("min_scene_quality", _c.c_float), ("min_tracking_quality", _c.c_float)]


class StabStats(_c.Structure):
    _fields_ = [("tracking_stability", _c.c_float), ("scene_quality", _c.c_float), ("trust", _c.c_float), ("distribution", _c.c_float),
                ("n_detected", _c.c_int), ("n_matched", _c.c_int), ("n_tracked", _c.c_int), ("frame_delay", _c.c_int),
                ("smoothing_factor", _c.c_double), ("homography", _c.c_double * 9)]


def preset(name="homography", **over):
    """The two OBS presets (Modules/OBS-Plugin/Sources/Stabilisation/VSFilter.cpp:255-293) on top of the library defaults."""
    s = StabSettings()
    s.detection_width, s.detection_height = 256, 256
    s.detection_regions_x, s.detection_regions_y, s.force_detection = 2, 2, 0
    s.max_feature_density, s.min_feature_density, s.accumulation_rate = 0.20, 0.05, 2.0
    s.track_local_motions, s.temporal_smoothing, s.local_smoothing = 1, 1.0, 20.0
    s.min_motion_samples, s.acceptance_threshold, s.uniformity_threshold = 75, 8.0, 0.20
    s.predictive_samples, s.corrective_limit_x, s.corrective_limit_y, s.smoothing_steps, s.response_rate = 10, 0.1, 0.1, 20.0, 0.04
    s.motion_width, s.motion_height = 2, 2
    s.background[0], s.background[1], s.background[2] = 255, 0, 255
    s.crop_to_stable_region, s.stabilize_output, s.min_scene_quality, s.min_tracking_quality = 0, 1, 0.8, 0.3
    if name == "homography":
        s.detection_width, s.detection_height = 480, 270
        s.acceptance_threshold, s.track_local_motions = 3.0, 0
        s.motion_width, s.motion_height = 2, 2
        s.detection_regions_x, s.detection_regions_y = 2, 1
        s.max_feature_density, s.min_feature_density, s.accumulation_rate = 0.12, 0.04, 3.0
    elif name == "field":
        s.detection_width, s.detection_height = 480, 270
        s.acceptance_threshold, s.track_local_motions = 10.0, 1
        s.motion_width, s.motion_height = 16, 16
        s.detection_regions_x, s.detection_regions_y = 2, 2
        s.max_feature_density, s.min_feature_density, s.accumulation_rate = 0.12, 0.06, 3.0
    elif name != "default":
        raise ValueError(name)
    if name in ("homography", "field"):
        s.min_scene_quality, s.min_tracking_quality = 0.95, 0.35           # "strict" QA
        s.corrective_limit_x = s.corrective_limit_y = 0.05                  # UI crop default 5 %
        s.crop_to_stable_region = 1
        s.background[0], s.background[1], s.background[2] = 105, 212, 235   # rgb2yuv of the default magenta (approx.)
    for k, v in over.items():
        setattr(s, k, v)
    return s


class OracleMeshSolver:
    def __init__(self, oracle, cols, rows, gen_region=(480, 270), temporal=1.0, local=20.0):
        L = self.L = oracle.lib
        L.lvko_mesh_solver_create.restype = _c.c_void_p
        L.lvko_mesh_solver_create.argtypes = [_c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_float, _c.c_float]
        L.lvko_mesh_solver_destroy.argtypes = [_c.c_void_p]
        L.lvko_mesh_solver_reset.argtypes = [_c.c_void_p]
        L.lvko_mesh_solver_static_rows.argtypes = [_c.c_void_p]
        L.lvko_mesh_solver_static_triplets.argtypes = [_c.c_void_p]
        L.lvko_mesh_solver_mesh.restype = _f32p
        L.lvko_mesh_solver_mesh.argtypes = [_c.c_void_p]
        L.lvko_mesh_solver_solve.restype = _c.c_int
        L.lvko_mesh_solver_solve.argtypes = [_c.c_void_p, _f32p, _f32p, _c.c_int, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _u8p, _f32p]
        self.cols, self.rows = cols, rows
        self.h = L.lvko_mesh_solver_create(cols, rows, gen_region[0], gen_region[1], temporal, local)

    def static_counts(self):
        return self.L.lvko_mesh_solver_static_rows(self.h), self.L.lvko_mesh_solver_static_triplets(self.h)

    def solve(self, tracked, matched, region=(480, 270), temporal=1.0, threshold=10.0):
        t = np.ascontiguousarray(tracked, np.float32).reshape(-1, 2); m = np.ascontiguousarray(matched, np.float32).reshape(-1, 2)
        inl = np.zeros(len(t), np.uint8); off = np.zeros((self.rows, self.cols, 2), np.float32)
        rc = self.L.lvko_mesh_solver_solve(self.h, _p(t, _f32p), _p(m, _f32p), len(t), region[0], region[1], temporal, threshold,
                                           _p(inl, _u8p), _p(off, _f32p))
        return rc, inl, off

    def reset(self):
        self.L.lvko_mesh_solver_reset(self.h)

    def mesh(self):
        ptr = self.L.lvko_mesh_solver_mesh(self.h)
        return np.ctypeslib.as_array(ptr, shape=(self.rows, self.cols, 2)).copy()

    def close(self):
        if self.h:
            self.L.lvko_mesh_solver_destroy(self.h); self.h = None


class OracleStabilizer:
    def __init__(self, oracle, settings):
        self.L = oracle.lib
        L = self.L
        L.lvko_stab_create.restype = _c.c_void_p
        L.lvko_stab_create.argtypes = [_c.POINTER(StabSettings)]
        L.lvko_stab_destroy.argtypes = [_c.c_void_p]
        L.lvko_stab_configure.argtypes = [_c.c_void_p, _c.POINTER(StabSettings)]
        L.lvko_stab_restart.argtypes = [_c.c_void_p]
        L.lvko_stab_push.restype = _c.c_int
        L.lvko_stab_push.argtypes = [_c.c_void_p, _u8p, _c.c_int, _c.c_int, _c.c_int, _c.c_uint64, _u8p, _c.c_int,
                                     _c.POINTER(_c.c_uint64), _c.c_int]
        L.lvko_stab_get_stats.argtypes = [_c.c_void_p, _c.POINTER(StabStats)]
        L.lvko_stab_get_meshes.restype = _c.c_int
        L.lvko_stab_get_meshes.argtypes = [_c.c_void_p, _f32p, _f32p, _c.c_int]
        L.lvko_stab_get_features.restype = _c.c_int
        L.lvko_stab_get_features.argtypes = [_c.c_void_p, _f32p, _c.c_int]
        self.settings = settings
        self.h = L.lvko_stab_create(_c.byref(settings))

    def close(self):
        if self.h:
            self.L.lvko_stab_destroy(self.h)
            self.h = None

    def configure(self, settings):
        self.settings = settings
        self.L.lvko_stab_configure(self.h, _c.byref(settings))

    def restart(self):
        self.L.lvko_stab_restart(self.h)

    def draw_trackers(self):
        self.L.lvko_stab_draw_trackers.argtypes = [_c.c_void_p]; self.L.lvko_stab_draw_trackers.restype = None
        self.L.lvko_stab_draw_trackers(self.h)

    def draw_motion_mesh(self):
        self.L.lvko_stab_draw_motion_mesh.argtypes = [_c.c_void_p]; self.L.lvko_stab_draw_motion_mesh.restype = None
        self.L.lvko_stab_draw_motion_mesh(self.h)

    def set_lens(self, params):
        self.L.lvko_stab_set_lens.argtypes = [_c.c_void_p, _f64p]
        self.L.lvko_stab_set_lens.restype = None
        if params is None:
            self.L.lvko_stab_set_lens(self.h, None)
        else:
            pr = np.ascontiguousarray(params, np.float64).reshape(9)
            self.L.lvko_stab_set_lens(self.h, _p(pr, _f64p))

    def push(self, frame, ts=0, nthreads=8, fmt=4, out=None):
        """out: a buffer for the emitted frame when it may be LARGER than the pushed one (a stream whose frame size changes: the queue
        holds whole frames; the caller crops the buffer to the emitted frame's size, which it knows from the timestamp)."""
        frame = np.ascontiguousarray(frame, np.uint8)
        out = np.zeros_like(frame) if out is None else out
        ots = _c.c_uint64(0)
        self.L.lvko_stab_push_fmt.restype = _c.c_int
        self.L.lvko_stab_push_fmt.argtypes = [_c.c_void_p, _u8p, _c.c_int, _c.c_int, _c.c_int, _c.c_uint64, _c.c_int, _u8p, _c.c_int,
                                              _c.POINTER(_c.c_uint64), _c.c_int]
        rc = self.L.lvko_stab_push_fmt(self.h, _p(frame, _u8p), frame.strides[0], frame.shape[0], frame.shape[1], ts, fmt,
                                       _p(out, _u8p), out.strides[0], _c.byref(ots), nthreads)
        assert rc >= 0
        return (out, ots.value) if rc == 1 else (None, None)

    def stats(self):
        st = StabStats()
        self.L.lvko_stab_get_stats(self.h, _c.byref(st))
        return st

    STAGES = ("downscale", "detect", "lk", "estimate", "smooth", "remap")

    def stage_ms(self, reset=False):
        """Accumulated wall time per stage (ms) of the pushes so far: {downscale, detect, lk, estimate, smooth, remap}."""
        a = (_c.c_double * 6)()
        self.L.lvko_stab_get_stage_ms.restype = None
        self.L.lvko_stab_get_stage_ms.argtypes = [_c.c_void_p, _c.POINTER(_c.c_double), _c.c_int]
        self.L.lvko_stab_get_stage_ms(self.h, a, 1 if reset else 0)
        return dict(zip(self.STAGES, [float(v) for v in a]))

    def meshes(self):
        n = self.settings.motion_width * self.settings.motion_height * 2
        a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
        self.L.lvko_stab_get_meshes(self.h, _p(a, _f32p), _p(b, _f32p), n)
        shp = (self.settings.motion_height, self.settings.motion_width, 2)
        return a.reshape(shp), b.reshape(shp)

    def features(self, cap=4096):
        a = np.zeros((cap, 4), np.float32)
        n = self.L.lvko_stab_get_features(self.h, _p(a, _f32p), cap)
        return a[:n].copy()

    def matches(self, cap=8192):
        """(tracked, matched, estimator) of the last frame's motion estimate; estimator 0 = none ran, 1 homography, 2 affine fallback, 3 mesh."""
        a = np.zeros((cap, 2), np.float32); b = np.zeros((cap, 2), np.float32); e = _c.c_int(0)
        self.L.lvko_stab_get_matches.restype = _c.c_int
        self.L.lvko_stab_get_matches.argtypes = [_c.c_void_p, _f32p, _f32p, _c.c_int, _c.POINTER(_c.c_int)]
        n = self.L.lvko_stab_get_matches(self.h, _p(a, _f32p), _p(b, _f32p), cap, _c.byref(e))
        return a[:max(n, 0)].copy(), b[:max(n, 0)].copy(), e.value


def require_live_warp(ost, what="", min_trust=0.1):
    """Every full-size parity test calls this on its ORACLE stabilizer after the last push: the comparison only says something about the
    stabilizing warp when the quality-assurance trust factor (Filters/StabilizationFilter.cpp:101-115) has left zero -- with trust 0 the
    correction is scaled to nothing and every compared frame is crop (+ lens) only, the tracker's estimate never reaches a pixel.  Fails
    (does not skip) so that a clip or preset change cannot silently turn a warp test back into a crop test."""
    st = ost.stats()
    if not st.trust > min_trust:
        raise AssertionError(f"{what}: the oracle's trust factor ended at {st.trust:.2f} (<= {min_trust}): the compared frames carry no "
                             f"stabilizing warp (scene quality {st.scene_quality:.3f}, tracking stability {st.tracking_stability:.3f}); "
                             f"relax min_scene_quality / min_tracking_quality or push more frames")
    motion, corr = ost.meshes()
    if st.n_matched < 50 or not np.abs(motion).max() > 0:
        raise AssertionError(f"{what}: the last frame's motion estimate is empty ({st.n_matched} matches, zero motion mesh): nothing was tracked")
    if not np.abs(corr).max() > 0:
        raise AssertionError(f"{what}: the correction mesh is all zero")
    return st.trust


_inst = None
_fast = None


def load_fast():
    """The TIMING build of the same sources (`make -C oracle fast`: -O3 -march=native) for bench.py's cpu_baseline, built
    on the machine that runs it into oracle/_fast/<cpu>/ (never shipped: a -march=native library of another CPU must not be loaded).  Not the
    parity checker -- bench.py holds its frames to load()'s before it reports a number from it.  Returns None when it cannot be built."""
    global _fast
    if _fast is None:
        import hashlib
        try:
            model = [ln for ln in open("/proc/cpuinfo") if ln.startswith(("model name", "flags"))][:2]
        except OSError:
            model = []
        tag = hashlib.sha1("".join(model).encode()).hexdigest()[:12]
        out_dir = os.path.join(ORACLE_DIR, "_fast", tag)
        lib = os.path.join(out_dir, "liblvk_oracle_fast.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h"))]
        try:
            if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
                os.makedirs(out_dir, exist_ok=True)
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "fast", "FASTDIR=" + out_dir])
            _fast = Oracle(ctypes.CDLL(lib))
            _fast.set_device_rcp(True)
        except (OSError, subprocess.CalledProcessError):
            return None
    return _fast


RCP_FIXTURE = os.path.join(ROOT, "tests", "golden", "gfx950_rcp.npz")


def device_rcp_table(path=RCP_FIXTURE):
    """v_rcp_f32 of gfx950 over the 2^23 mantissas of [1, 2) (float32 array), decoded from the committed device dump
    (scripts/dump_rcp_table.py: 2-bit deltas to the correctly rounded reciprocal; tests/test_ref_pin_gpu.py checks it against the GPU)."""
    packed = np.load(path)["packed"]
    code = np.empty(1 << 23, np.uint8)
    for k in range(4):
        code[k::4] = (packed >> (2 * k)) & 3
    delta = np.where(code == 3, -1, code.astype(np.int64))
    x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)
    cr = (1.0 / x.astype(np.float64)).astype(np.float32)
    return (cr.view(np.uint32).astype(np.int64) + delta).astype(np.uint32).view(np.float32)


def load(device_rcp=True):
    """The oracle with native_recip = gfx950's v_rcp_f32 (what the reference's OpenCL kernels compile to for this device, oracle/_ref);
    device_rcp=False on the returned object's set_device_rcp() gives the correctly rounded 1/x instead."""
    global _inst
    if _inst is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h"))]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
            build()
        _inst = Oracle(ctypes.CDLL(LIB))
        _inst.set_device_rcp(device_rcp)
    return _inst
