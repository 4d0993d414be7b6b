"""Python mirror of the C-ABI context: device buffers are torch uint8 tensors, work runs on a HIP stream."""
import ctypes

import numpy as np

from . import _native


class LvkHipError(RuntimeError):
    pass


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _u8x3(bg):
    a = np.ascontiguousarray(bg, dtype=np.uint8).reshape(3)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


class Context:
    """One HIP stream + staging on one GPU (reference analogue: the implicit cv::ocl queue)."""

    def __init__(self, device=0, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise LvkHipError("no GPU visible: the lvk HIP path has no CPU fallback")
        self.lib = _native.load()
        self.device = device
        torch.cuda.set_device(device)
        self._torch_stream = stream if stream is not None else torch.cuda.current_stream(device)
        handle = ctypes.c_void_p()
        # enqueue on torch's stream so kernels are ordered with the tensors' producers/consumers
        rc = self.lib.lvk_hip_ctx_create_on_stream(device, ctypes.c_void_p(self._torch_stream.cuda_stream), ctypes.byref(handle))
        if rc != 0:
            raise LvkHipError(f"lvk_hip_ctx_create failed ({rc}): {self.lib.lvk_hip_last_error(None).decode()}")
        self.handle = handle

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lvk_hip_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise LvkHipError(f"lvk_hip call failed ({rc}): {self.lib.lvk_hip_last_error(self.handle).decode()}")

    def sync(self):
        self._check(self.lib.lvk_hip_sync(self.handle))

    # ---- a15/a16 -------------------------------------------------------------------------------
    def remap_homography(self, src, H, bg=(255, 0, 255), yuv=True, out=None, dst_size=None, offset=(0, 0)):
        """lvk::remap(src, dst, homography, background, inverted=true); src/out: torch uint8 [rows, cols, 3] on the GPU."""
        import torch
        rows, cols = src.shape[0], src.shape[1]
        drows, dcols = dst_size if dst_size is not None else (rows, cols)
        if out is None:
            out = torch.empty((drows, dcols, 3), dtype=torch.uint8, device=src.device)
        Ha, Hp = _f32(np.asarray(H, dtype=np.float32).reshape(9))
        bga, bgp = _u8x3(bg)
        self._check(self.lib.lvk_hip_remap_homography(
            self.handle, src.data_ptr(), src.stride(0), rows, cols, out.data_ptr(), out.stride(0), drows, dcols,
            offset[0], offset[1], Hp, bgp, 1 if yuv else 0))
        return out

    def remap_mesh(self, src, mesh, bg=(255, 0, 255), yuv=True, out=None):
        """lvk::remap(src, dst, offset_map, background) with the map interpolated in-kernel from `mesh` [mr, mc, 2]."""
        import torch
        rows, cols = src.shape[0], src.shape[1]
        if out is None:
            out = torch.empty((rows, cols, 3), dtype=torch.uint8, device=src.device)
        m = np.ascontiguousarray(mesh, dtype=np.float32)
        ma, mp = _f32(m)
        bga, bgp = _u8x3(bg)
        self._check(self.lib.lvk_hip_remap_mesh(
            self.handle, src.data_ptr(), src.stride(0), rows, cols, out.data_ptr(), out.stride(0),
            mp, m.shape[0], m.shape[1], bgp, 1 if yuv else 0))
        return out

    def warpmesh_apply(self, src, mesh, bg=(255, 0, 255), yuv=True, out=None):
        """WarpMesh::apply(src, dst, background)."""
        import torch
        rows, cols = src.shape[0], src.shape[1]
        if out is None:
            out = torch.empty((rows, cols, 3), dtype=torch.uint8, device=src.device)
        m = np.ascontiguousarray(mesh, dtype=np.float32)
        ma, mp = _f32(m)
        bga, bgp = _u8x3(bg)
        self._check(self.lib.lvk_hip_warpmesh_apply(
            self.handle, src.data_ptr(), src.stride(0), rows, cols, out.data_ptr(), out.stride(0),
            mp, m.shape[0], m.shape[1], bgp, 1 if yuv else 0))
        return out

    # ---- a3/a4/a7 image ops --------------------------------------------------------------------------
    def luma_area_resize(self, frame, drows, dcols, channel=0):
        """frame: torch uint8 [rows, cols, 3] (packed) or [rows, cols] (planar) -> [drows, dcols] uint8."""
        import torch
        pix = frame.shape[2] if frame.dim() == 3 else 1
        out = torch.empty((drows, dcols), dtype=torch.uint8, device=frame.device)
        self._check(self.lib.lvk_hip_luma_area_resize(self.handle, frame.data_ptr(), frame.stride(0), pix, channel,
                                                      frame.shape[0], frame.shape[1], out.data_ptr(), out.stride(0), drows, dcols))
        return out

    def pyr_down(self, img):
        import torch
        out = torch.empty(((img.shape[0] + 1) // 2, (img.shape[1] + 1) // 2), dtype=torch.uint8, device=img.device)
        self._check(self.lib.lvk_hip_pyr_down(self.handle, img.data_ptr(), img.stride(0), img.shape[0], img.shape[1],
                                              out.data_ptr(), out.stride(0)))
        return out

    def scharr(self, img):
        import torch
        out = torch.empty((img.shape[0], img.shape[1], 2), dtype=torch.int16, device=img.device)
        self._check(self.lib.lvk_hip_scharr(self.handle, img.data_ptr(), img.stride(0), img.shape[0], img.shape[1], out.data_ptr()))
        return out

    def build_pyramid(self, img, max_level=3, win=(11, 11)):
        """Returns [(level uint8 [r, c], deriv int16 [r, c, 2]), ...] as the LK tracker sees them."""
        rows, cols = img.shape
        lv = np.zeros(rows * cols * 2, np.uint8); dv = np.zeros(rows * cols * 4, np.int16)
        lr = np.zeros(8, np.int32); lc = np.zeros(8, np.int32)
        n = self.lib.lvk_hip_build_pyramid(self.handle, img.data_ptr(), img.stride(0), rows, cols, max_level, win[0], win[1],
                                           lv.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), dv.ctypes.data_as(ctypes.POINTER(ctypes.c_int16)),
                                           lr.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), lc.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        self._check(min(n, 0))
        out, lo, do = [], 0, 0
        for i in range(n):
            r, c = int(lr[i]), int(lc[i])
            out.append((lv[lo:lo + r * c].reshape(r, c).copy(), dv[do:do + r * c * 2].reshape(r, c, 2).copy()))
            lo += r * c; do += r * c * 2
        return out

    # ---- a5 / a7 ----------------------------------------------------------------------------------------
    def fast_detect(self, img, regions, cap=None):
        """regions: list of (x, y, w, h, threshold, active). Returns a list of [n, 3] int32 (x, y, score) arrays, region-local."""
        rg = np.ascontiguousarray(regions, dtype=np.int32).reshape(-1, 6)
        n = rg.shape[0]
        cap = cap or int(max(1, (rg[:, 2] * rg[:, 3]).max()))
        out = np.zeros((n, cap), np.uint32)
        counts = np.zeros(n, np.int32)
        self._check(self.lib.lvk_hip_fast_detect(self.handle, img.data_ptr(), img.stride(0), img.shape[0], img.shape[1],
                                                 rg.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n,
                                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), cap,
                                                 counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        res = []
        for i in range(n):
            e = out[i, :min(cap, counts[i])]
            res.append(np.stack([e & 0xFFF, (e >> 12) & 0xFFF, e >> 24], axis=1).astype(np.int32))
        return res, counts

    def pyrlk(self, prev, nxt, pts, win=(11, 11), max_level=3, max_count=5, epsilon=0.01, min_eig=1e-4):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        out = np.zeros_like(pts)
        st = np.zeros(len(pts), np.uint8)
        self._check(self.lib.lvk_hip_pyrlk(self.handle, prev.data_ptr(), prev.stride(0), nxt.data_ptr(), nxt.stride(0),
                                           prev.shape[0], prev.shape[1],
                                           pts.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(pts),
                                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                           st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                           win[0], win[1], max_level, max_count, float(epsilon), float(min_eig)))
        return out, st

    def estimate_global_motion(self, p1, p2, threshold, region=(480, 270), full_homography=True):
        """cv::findHomography(UsacParams) / cv::estimateAffinePartial2D stand-in; returns (n_inliers, H 3x3 float64, mask)."""
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        H = np.zeros(9, np.float64); mask = np.zeros(len(p1), np.uint8)
        rc = self.lib.lvk_hip_estimate_global_motion(self.handle, p1.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                     p2.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(p1), float(threshold),
                                                     float(region[0]), float(region[1]), 1 if full_homography else 0,
                                                     H.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                     mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
        if rc in (-1, -2, -3):
            self._check(rc)
        return rc, H.reshape(3, 3), mask

    # ---- YUV420 <-> packed 444 ----------------------------------------------------------
    def ingest_yuv420(self, y, u, v=None, out=None):
        """I420 (y, u, v) or NV12 (y, uv[r/2, c/2, 2]) torch uint8 planes -> packed [rows, cols, 3]."""
        import torch
        rows, cols = y.shape
        if out is None:
            out = torch.empty((rows, cols, 3), dtype=torch.uint8, device=y.device)
        nv12 = v is None
        vv = u if nv12 else v
        self._check(self.lib.lvk_hip_ingest_yuv420(self.handle, y.data_ptr(), y.stride(0), u.data_ptr(), u.stride(0), vv.data_ptr(), vv.stride(0),
                                                   1 if nv12 else 0, rows, cols, out.data_ptr(), out.stride(0)))
        return out

    def egress_yuv420(self, frame, nv12=False, out=None):
        import torch
        rows, cols = frame.shape[:2]
        if out is None:
            y = torch.empty((rows, cols), dtype=torch.uint8, device=frame.device)
            if nv12:
                u = torch.empty((rows // 2, cols // 2, 2), dtype=torch.uint8, device=frame.device); v = u
            else:
                u = torch.empty((rows // 2, cols // 2), dtype=torch.uint8, device=frame.device); v = torch.empty_like(u)
        else:
            y, u, v = out if not nv12 else (out[0], out[1], out[1])
        self._check(self.lib.lvk_hip_egress_yuv420(self.handle, frame.data_ptr(), frame.stride(0), rows, cols, y.data_ptr(), y.stride(0),
                                                   u.data_ptr(), u.stride(0), v.data_ptr(), v.stride(0), 1 if nv12 else 0))
        return (y, u) if nv12 else (y, u, v)

    # ---- every OBS video format of FrameIngest::Select (Modules/OBS-Plugin/Interop/FrameIngest.cpp:36-75) ----------------------
    VIDEO_FORMATS = {"I420": 1, "NV12": 2, "YVYU": 3, "YUY2": 4, "UYVY": 5, "RGBA": 6, "BGRA": 7, "BGRX": 8, "Y800": 9, "I444": 10, "BGR3": 11,
                     "I422": 12, "I40A": 13, "I42A": 14, "YUVA": 15, "AYUV": 16}

    def _obs_args(self, planes):
        import ctypes as c
        ptrs = (c.c_void_p * 3)(*[p.data_ptr() for p in planes] + [None] * (3 - len(planes)))
        steps = (c.c_int * 3)(*[p.stride(0) for p in planes] + [0] * (3 - len(planes)))
        return ptrs, steps

    def ingest_obs(self, fmt, planes, out=None):
        """FrameIngest::to_ocl: the planes OBS holds for one frame of `fmt` (torch uint8, on the GPU) -> the packed frame [rows, cols, 3] ([rows, cols] for Y800)."""
        import torch
        rows, cols = planes[0].shape[:2]
        if out is None:
            out = torch.empty((rows, cols) if fmt == "Y800" else (rows, cols, 3), dtype=torch.uint8, device=planes[0].device)
        ptrs, steps = self._obs_args(planes)
        self._check(self.lib.lvk_hip_ingest_obs(self.handle, self.VIDEO_FORMATS[fmt], ptrs, steps, rows, cols, out.data_ptr(), out.stride(0)))
        return out

    def egress_obs(self, fmt, frame, planes):
        """FrameIngest::to_obs: the frame into the given OBS planes (bytes the reference leaves alone stay what they are)."""
        rows, cols = frame.shape[:2]
        ptrs, steps = self._obs_args(planes)
        self._check(self.lib.lvk_hip_egress_obs(self.handle, self.VIDEO_FORMATS[fmt], frame.data_ptr(), frame.stride(0), rows, cols, ptrs, steps))
        return planes

    def obs_frame_format(self, fmt):
        return self.lib.lvk_hip_obs_frame_format(self.VIDEO_FORMATS[fmt] if isinstance(fmt, str) else int(fmt))

    # ---- lens correction --------------------------------------------------------------------
    def remap_map(self, src, offsets, bg=(255, 0, 255), yuv=True, out=None):
        """lvk::remap(src, dst, offset_map): offsets = torch float32 [rows, cols, 2] on the GPU (pixels)."""
        import torch
        rows, cols = src.shape[:2]
        if out is None:
            out = torch.empty_like(src)
        bga, bgp = _u8x3(bg)
        self._check(self.lib.lvk_hip_remap_map(self.handle, src.data_ptr(), src.stride(0), rows, cols, out.data_ptr(), out.stride(0),
                                               offsets.data_ptr(), offsets.stride(0) * 4, bgp, 1 if yuv else 0))
        return out

    def upscale(self, src, size, yuv=True, out=None):
        """lvk::upscale(src, dst, size, yuv): EASU upsampling to size = (width, height) >= the source size."""
        import torch
        rows, cols = src.shape[:2]
        if out is None:
            out = torch.empty((int(size[1]), int(size[0]), 3), dtype=torch.uint8, device=src.device)
        self._check(self.lib.lvk_hip_upscale(self.handle, src.data_ptr(), src.stride(0), rows, cols,
                                             out.data_ptr(), out.stride(0), out.shape[0], out.shape[1], 1 if yuv else 0))
        return out

    def sharpen(self, src, sharpness=0.7, out=None):
        """lvk::sharpen(src, dst, sharpness): RCAS, out of place."""
        import torch
        rows, cols = src.shape[:2]
        if out is None:
            out = torch.empty_like(src)
        self._check(self.lib.lvk_hip_sharpen(self.handle, src.data_ptr(), src.stride(0), rows, cols, out.data_ptr(), out.stride(0),
                                             float(sharpness)))
        return out

    def mesh_solver(self, cols, rows, gen_region=(480, 270), temporal=1.0, local=20.0, max_points=4096):
        """FrameTracker's local-motion solver (stage a10) as an object with .solve(tracked, matched, ...), .reset(), .close()."""
        return MeshSolver(self, cols, rows, gen_region, temporal, local, max_points)

    def native_rcp(self, x):
        """native_recip of FSR.cl as this device defines it (v_rcp_f32), elementwise; x: torch float32 on the GPU."""
        import torch
        x = x.contiguous()
        out = torch.empty_like(x)
        self._check(self.lib.lvk_hip_native_rcp(self.handle, x.data_ptr(), out.data_ptr(), x.numel()))
        return out

    def lens_map(self, params, rows, cols):
        """Device offset map of LCFilter for camera params (fx, fy, cx, cy, k1, k2, p1, p2, k3): (torch view [rows, cols, 2], view_xywh)."""
        import torch
        arr = (ctypes.c_double * 9)(*[float(v) for v in params])
        d = ctypes.c_void_p(); view = (ctypes.c_int * 4)()
        self._check(self.lib.lvk_hip_lens_map_create(self.handle, arr, rows, cols, ctypes.byref(d), view))
        # hand ownership to torch: stage through host once (the map is static per profile / frame size)
        tmp = np.zeros((rows, cols, 2), np.float32)
        self._check(self.lib.lvk_hip_download(self.handle, tmp.ctypes.data_as(ctypes.c_void_p), d, tmp.nbytes))
        self.sync()
        t = torch.from_numpy(tmp).to("cuda")
        self._check(self.lib.lvk_hip_lens_map_destroy(self.handle, d))
        return t, tuple(view)

    def warpmesh_apply_lens(self, src, mesh, params, bg=(255, 0, 255), yuv=True, out=None):
        """WarpMesh::apply on the lens-corrected frame, sampled from the RAW frame `src` in one pass (fused lens mode)."""
        import torch
        rows, cols = src.shape[0], src.shape[1]
        if out is None:
            out = torch.empty((rows, cols, 3), dtype=torch.uint8, device=src.device)
        m = np.ascontiguousarray(mesh, dtype=np.float32)
        ma, mp = _f32(m)
        bga, bgp = _u8x3(bg)
        arr = (ctypes.c_double * 9)(*[float(v) for v in params])
        self._check(self.lib.lvk_hip_warpmesh_apply_lens(self.handle, src.data_ptr(), src.stride(0), rows, cols, out.data_ptr(), out.stride(0),
                                                         mp, m.shape[0], m.shape[1], bgp, 1 if yuv else 0, arr))
        return out

    def lens_undistort_points(self, params, rows, cols, sx, sy, pts):
        """Raw tracking-frame points -> lens-corrected positions (binary64 on the GPU)."""
        p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        out = np.zeros_like(p)
        arr = (ctypes.c_double * 9)(*[float(v) for v in params])
        fp = ctypes.POINTER(ctypes.c_float)
        self._check(self.lib.lvk_hip_lens_undistort_points(self.handle, arr, rows, cols, float(sx), float(sy), p.ctypes.data_as(fp), len(p), out.ctypes.data_as(fp)))
        return out

    # ---- debug overlays ---------------------------------------------------------------------
    def draw_grid(self, frame, grid, colour, thickness=1):
        """lvk::draw_grid(dst, grid = (w, h) cells, colour, thickness): draws into `frame` in place."""
        ca, cp = _u8x3(colour)
        self._check(self.lib.lvk_hip_draw_grid(self.handle, frame.data_ptr(), frame.stride(0), frame.shape[0], frame.shape[1], int(grid[0]), int(grid[1]), cp, thickness))
        return frame

    def draw_crosses(self, frame, points, colour, cross_size, thickness, scaling=(1.0, 1.0)):
        """lvk::draw_crosses(dst, points, colour, size, thickness, coord_scaling): draws into `frame` in place."""
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        ca, cp = _u8x3(colour)
        self._check(self.lib.lvk_hip_draw_crosses(self.handle, frame.data_ptr(), frame.stride(0), frame.shape[0], frame.shape[1],
                                                  p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(p), float(scaling[0]), float(scaling[1]), cp,
                                                  int(cross_size), int(thickness)))
        return frame

    def fast_filter(self, prev, matched, status):
        """GPU-side fast_filter of the flow result: (kept prev, kept matched) in the reference's swap-erase order."""
        a = np.ascontiguousarray(prev, np.float32).reshape(-1, 2); b = np.ascontiguousarray(matched, np.float32).reshape(-1, 2)
        st = np.ascontiguousarray(status, np.uint8).reshape(-1)
        oa = np.zeros_like(a); ob = np.zeros_like(b)
        fp = ctypes.POINTER(ctypes.c_float)
        m = self.lib.lvk_hip_fast_filter(self.handle, a.ctypes.data_as(fp), b.ctypes.data_as(fp), st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), len(a),
                                         oa.ctypes.data_as(fp), ob.ctypes.data_as(fp))
        if m < 0:
            self._check(m)
        return oa[:m], ob[:m]

    def warpmesh_apply_yuv420(self, src, mesh, bg=(255, 0, 255), nv12=False):
        """WarpMesh::apply + 4:2:0 egress in one kernel: packed YUV [rows, cols, 3] -> (y, u, v) I420 or (y, uv) NV12 planes."""
        import torch
        rows, cols = src.shape[0], src.shape[1]
        y = torch.empty((rows, cols), dtype=torch.uint8, device=src.device)
        if nv12:
            u = torch.empty((rows // 2, cols // 2, 2), dtype=torch.uint8, device=src.device); v = u
        else:
            u = torch.empty((rows // 2, cols // 2), dtype=torch.uint8, device=src.device); v = torch.empty_like(u)
        m = np.ascontiguousarray(mesh, dtype=np.float32)
        ma, mp = _f32(m)
        bga, bgp = _u8x3(bg)
        self._check(self.lib.lvk_hip_warpmesh_apply_yuv420(self.handle, src.data_ptr(), src.stride(0), rows, cols, y.data_ptr(), y.stride(0),
                                                           u.data_ptr(), u.stride(0), v.data_ptr(), v.stride(0), 1 if nv12 else 0,
                                                           mp, m.shape[0], m.shape[1], bgp))
        return (y, u) if nv12 else (y, u, v)


class MeshSolver:
    """lvk_hip_mesh_solver_*: FrameTracker::estimate_local_motions on the device; keeps the previous solution between calls."""

    def __init__(self, ctx, cols, rows, gen_region, temporal, local, max_points):
        self.ctx, self.cols, self.rows = ctx, cols, rows
        h = ctypes.c_void_p()
        ctx._check(ctx.lib.lvk_hip_mesh_solver_create(ctx.handle, cols, rows, float(gen_region[0]), float(gen_region[1]), float(temporal), float(local),
                                                      int(max_points), ctypes.byref(h)))
        self.handle = h

    def solve(self, tracked, matched, region=(480, 270), temporal=1.0, threshold=10.0):
        """Returns (status, inliers uint8 [n], offsets float32 [rows, cols, 2]); status 0 = estimate, 2 / 3 = none."""
        t = np.ascontiguousarray(tracked, np.float32).reshape(-1, 2); m = np.ascontiguousarray(matched, np.float32).reshape(-1, 2)
        inl = np.zeros(len(t), np.uint8); off = np.zeros((self.rows, self.cols, 2), np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        rc = self.ctx.lib.lvk_hip_mesh_solver_solve(self.handle, t.ctypes.data_as(fp), m.ctypes.data_as(fp), len(t), float(region[0]), float(region[1]),
                                                    float(temporal), float(threshold), inl.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), off.ctypes.data_as(fp))
        if rc < 0:
            self.ctx._check(rc)
        return rc, inl, off

    def reset(self):
        self.ctx._check(self.ctx.lib.lvk_hip_mesh_solver_reset(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):          # (never hand the library a solver whose context is gone)
                self.ctx.lib.lvk_hip_mesh_solver_destroy(self.handle)
            self.handle = None
