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
