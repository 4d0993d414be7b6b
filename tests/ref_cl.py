"""Launches the REFERENCE's own OpenCL kernels (oracle/_ref/*.hsaco, compiled by `make -C oracle ref` from
LiveVisionKit/Functions/OpenCL/Sources/{FSR,Drawing}.cl with the ROCm clang) on the GPU.  TEST INFRASTRUCTURE ONLY.

The code objects are loaded with hipModuleLoad and launched with the argument lists the reference's host code builds:
  lvk::remap(map)        Functions/Image.cpp:65-76    cv::ocl::KernelArg::ReadOnly(src) = (ptr, step, offset, rows, cols),
                                                      WriteOnlyNoSize(dst) = (ptr, step, offset), Vec4i bounds, ReadOnlyNoSize(map), Vec4b
  lvk::remap(homography) Functions/Image.cpp:133-146  ... Vec4f x3 (rows of the double matrix cast to float), Vec4b
  lvk::upscale           Functions/Image.cpp:189-196  ReadOnly(src), WriteOnly(dst) = (ptr, step, offset, rows, cols), Vec2f
  lvk::sharpen           Functions/Image.cpp:225-229  ReadOnly(src), WriteOnlyNoSize(dst), float
  lvk::draw_grid/crosses Functions/Drawing.cpp        (see the launchers below)
and the work sizes of ocl::optimal_groups (Functions/OpenCL/Kernels.cpp:49-71): 8 x 8 groups over ceil(size / 8) * 8.
Tensors are torch uint8 / float32 CUDA tensors; launches go to torch's current stream.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

_c = ctypes


def available():
    return all(os.path.exists(os.path.join(REF_DIR, f)) for f in ("fsr_yuv.hsaco", "fsr_bgr.hsaco", "drawing.hsaco"))


class _I4(_c.Structure):
    _fields_ = [("v", _c.c_int * 4)]


class _F4(_c.Structure):
    _fields_ = [("v", _c.c_float * 4)]


class _F2(_c.Structure):
    _fields_ = [("v", _c.c_float * 2)]


class _B4(_c.Structure):
    _fields_ = [("v", _c.c_uint8 * 4)]


def _hip():
    import torch  # noqa: F401  (loads the HIP runtime the process shares)
    for name in ("libamdhip64.so.7", "libamdhip64.so"):
        try:
            return _c.CDLL(name)
        except OSError:
            continue
    import torch as t
    return _c.CDLL(os.path.join(os.path.dirname(t.__file__), "lib", "libamdhip64.so"))


class RefKernels:
    """The reference's compiled OpenCL programs: fsr (YUV_INPUT / plain) and drawing."""

    def __init__(self):
        import torch
        assert torch.cuda.is_available()
        torch.cuda.init()
        torch.zeros(1, device="cuda")            # make sure the primary context exists
        self.hip = _hip()
        self.hip.hipModuleLoad.argtypes = [_c.POINTER(_c.c_void_p), _c.c_char_p]
        self.hip.hipModuleGetFunction.argtypes = [_c.POINTER(_c.c_void_p), _c.c_void_p, _c.c_char_p]
        self.hip.hipModuleLaunchKernel.argtypes = [_c.c_void_p] + [_c.c_uint] * 6 + [_c.c_uint, _c.c_void_p, _c.POINTER(_c.c_void_p), _c.c_void_p]
        self.mods, self.funcs = {}, {}
        for key, f in (("yuv", "fsr_yuv.hsaco"), ("bgr", "fsr_bgr.hsaco"), ("draw", "drawing.hsaco")):
            m = _c.c_void_p()
            rc = self.hip.hipModuleLoad(_c.byref(m), os.path.join(REF_DIR, f).encode())
            assert rc == 0, f"hipModuleLoad({f}) = {rc}"
            self.mods[key] = m

    def _fn(self, mod, name):
        k = (mod, name)
        if k not in self.funcs:
            f = _c.c_void_p()
            rc = self.hip.hipModuleGetFunction(_c.byref(f), self.mods[mod], name.encode())
            assert rc == 0, f"hipModuleGetFunction({name}) = {rc}"
            self.funcs[k] = f
        return self.funcs[k]

    def _launch(self, fn, gx, gy, bx, by, args):
        self._prepare(fn, gx, gy, bx, by, args)()

    def _prepare(self, fn, gx, gy, bx, by, args):
        """The launch with its argument block marshalled ONCE: returns a callable that only calls hipModuleLaunchKernel on torch's current
        stream (a timing loop then measures the kernel, not the ctypes marshalling of its arguments)."""
        import torch
        holders = []
        for a in args:
            if isinstance(a, int):
                holders.append(_c.c_int(a))
            elif isinstance(a, float):
                holders.append(_c.c_float(a))
            else:
                holders.append(a)
        ptrs = (_c.c_void_p * len(holders))(*[_c.cast(_c.pointer(h), _c.c_void_p) for h in holders])
        launch = self.hip.hipModuleLaunchKernel

        def go(_keep=(holders, ptrs)):
            rc = launch(fn, gx, gy, 1, bx, by, 1, 0, _c.c_void_p(torch.cuda.current_stream().cuda_stream), ptrs, None)
            assert rc == 0, f"hipModuleLaunchKernel = {rc}"
        return go

    @staticmethod
    def _groups(cols, rows):
        return (cols + 7) // 8, (rows + 7) // 8

    @staticmethod
    def _ptr(t):
        return _c.c_void_p(t.data_ptr())

    @staticmethod
    def _bg(bg):
        return _B4((_c.c_uint8 * 4)(int(bg[0]), int(bg[1]), int(bg[2]), 0))

    # ---- FSR.cl ---------------------------------------------------------------------------------------------------
    def remap_homography(self, src, H, bg=(255, 0, 255), yuv=True, dst_size=None, offset=(0, 0), out=None, prepared=False):
        """easu_remap_homography as lvk::remap(src, dst, H, bg, inverted = true) launches it; H = dst -> src, row major.
        prepared=True: returns (launch callable, out) instead of launching."""
        import torch
        rows, cols = src.shape[:2]
        drows, dcols = dst_size if dst_size is not None else (rows, cols)
        if out is None:
            out = torch.zeros((drows, dcols, 3), dtype=torch.uint8, device=src.device)
        Hd = np.asarray(H, np.float64).reshape(3, 3)
        r = [_F4((_c.c_float * 4)(float(np.float32(Hd[i, 0])), float(np.float32(Hd[i, 1])), float(np.float32(Hd[i, 2])), 0.0)) for i in range(3)]
        gx, gy = self._groups(dcols, drows)
        go = self._prepare(self._fn("yuv" if yuv else "bgr", "easu_remap_homography"), gx, gy, 8, 8, [
            self._ptr(src), src.stride(0), 0, rows, cols,
            self._ptr(out), out.stride(0), 0, _I4((_c.c_int * 4)(offset[0], offset[1], dcols, drows)),
            r[0], r[1], r[2], self._bg(bg)])
        if prepared:
            return go, out
        go()
        return out

    def remap_map(self, src, offsets, bg=(255, 0, 255), yuv=True, out=None):
        """easu_remap as lvk::remap(src, dst, offset_map, bg) launches it; offsets: float32 [rows, cols, 2] on the GPU (pixels)."""
        import torch
        rows, cols = src.shape[:2]
        drows, dcols = offsets.shape[:2]
        if out is None:
            out = torch.zeros((drows, dcols, 3), dtype=torch.uint8, device=src.device)
        gx, gy = self._groups(dcols, drows)
        self._launch(self._fn("yuv" if yuv else "bgr", "easu_remap"), gx, gy, 8, 8, [
            self._ptr(src), src.stride(0), 0, rows, cols,
            self._ptr(out), out.stride(0), 0, _I4((_c.c_int * 4)(0, 0, dcols, drows)),
            self._ptr(offsets), offsets.stride(0) * 4, 0, self._bg(bg)])
        return out

    def upscale(self, src, size, yuv=True, out=None):
        """easu_scale as lvk::upscale(src, dst, size = (width, height), yuv) launches it."""
        import torch
        rows, cols = src.shape[:2]
        dcols, drows = int(size[0]), int(size[1])
        if out is None:
            out = torch.zeros((drows, dcols, 3), dtype=torch.uint8, device=src.device)
        rs = _F2((_c.c_float * 2)(float(np.float32(cols) / np.float32(dcols)), float(np.float32(rows) / np.float32(drows))))
        gx, gy = self._groups(dcols, drows)
        self._launch(self._fn("yuv" if yuv else "bgr", "easu_scale"), gx, gy, 8, 8, [
            self._ptr(src), src.stride(0), 0, rows, cols, self._ptr(out), out.stride(0), 0, drows, dcols, rs])
        return out

    def sharpen(self, src, sharpness, out=None):
        """rcas as lvk::sharpen(src, dst, sharpness) launches it (plain program, Image.cpp:214).  `out` should be padded by the caller
        when the size is not a multiple of 8: the padding work-items pass the kernel's `<=` guard and write (FSR.cl:478)."""
        import torch
        rows, cols = src.shape[:2]
        if out is None:
            out = torch.zeros_like(src)
        gx, gy = self._groups(cols, rows)
        # Image.cpp:227: std::exp2(-2.0f * (1.0f - sharpness)) in binary32 = glibc's exp2f (numpy's SIMD exp2 can differ in the last bit)
        libm = _c.CDLL("libm.so.6"); libm.exp2f.restype = _c.c_float; libm.exp2f.argtypes = [_c.c_float]
        s = float(libm.exp2f(_c.c_float(float(np.float32(-2.0) * (np.float32(1.0) - np.float32(sharpness))))))
        self._launch(self._fn("bgr", "rcas"), gx, gy, 8, 8, [
            self._ptr(src), src.stride(0), 0, rows, cols, self._ptr(out), out.stride(0), 0, s])
        return out

    # ---- Drawing.cl -----------------------------------------------------------------------------------------------
    def draw_grid(self, dst, cell_w, cell_h, thickness, colour):
        rows, cols = dst.shape[:2]
        gx, gy = self._groups(cols, rows)
        self._launch(self._fn("draw", "grid"), gx, gy, 8, 8, [
            self._ptr(dst), dst.stride(0), 0, rows, cols, float(cell_w), float(cell_h), int(thickness), self._bg(colour)])
        return dst

    def draw_crosses(self, dst, points_i32, cross_size, thickness, colour):
        """points_i32: torch int32 [n, 2] on the GPU (already scaled and rounded, as cv::multiply(.., CV_32S) leaves them);
        cross_size is the caller's value: the kernel receives (cross_size + 1) / 2 (Drawing.tpp:183)."""
        rows, cols = dst.shape[:2]
        n = points_i32.shape[0]
        self._launch(self._fn("draw", "crosses"), (n + 63) // 64, 1, 64, 1, [
            self._ptr(points_i32), 8, 0, n, 1, self._ptr(dst), dst.stride(0), 0, rows, cols,
            (int(cross_size) + 1) // 2, int(thickness), self._bg(colour)])
        return dst
