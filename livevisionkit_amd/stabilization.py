"""Host-side mirror of lvk::StabilizationFilter (reference: LiveVisionKit/Filters/StabilizationFilter.hpp:42-78,
LiveVisionKit/Filters/VideoFilter.hpp:32-61) over the C-ABI.  Same method names and argument meaning:
configure / reconfigure / settings / apply / restart / ready / reset_context / frame_delay / stable_region / timings.
Frames are torch uint8 tensors [rows, cols, 3] resident on the GPU (packed YUV, the reference's 8UC3 VideoFrame)."""
import ctypes
import time

import numpy as np

from . import _native
from .context import Context

_c = ctypes

FORMAT_BGR, FORMAT_BGRA, FORMAT_RGB, FORMAT_RGBA, FORMAT_YUV, FORMAT_GRAY = range(6)


class StabilizationFilterSettings(_c.Structure):
    """lvk::StabilizationFilterSettings flattened (field-for-field lvk_stab_settings of include/lvk_hip.h)."""
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

    def __init__(self, **over):
        super().__init__()
        _native.load().lvk_stab_default_settings(_c.byref(self))
        for k, v in over.items():
            setattr(self, k, v)

    def copy(self):
        other = StabilizationFilterSettings()
        _c.memmove(_c.byref(other), _c.byref(self), _c.sizeof(self))
        return other

    @classmethod
    def obs_preset(cls, subsystem="homography", strict=True, crop=0.05, predictive_samples=10, apply_crop=True, **over):
        """The OBS plugin's presets (Modules/OBS-Plugin/Sources/Stabilisation/VSFilter.cpp:235-294)."""
        s = cls()
        s.detection_width, s.detection_height = 480, 270
        if subsystem == "field":
            s.acceptance_threshold, s.track_local_motions = 10.0, 1
            s.motion_width, s.motion_height = 16, 16
            s.detection_regions_x, s.detection_regions_y = 2, 2
            s.max_feature_density, s.min_feature_density, s.accumulation_rate = 0.12, 0.06, 3.0
        else:
            s.acceptance_threshold, s.track_local_motions = 3.0, 0
            s.motion_width, s.motion_height = 2, 2
            s.detection_regions_x, s.detection_regions_y = 2, 1
            s.max_feature_density, s.min_feature_density, s.accumulation_rate = 0.12, 0.04, 3.0
        s.min_scene_quality, s.min_tracking_quality = (0.95, 0.35) if strict else (0.40, 0.20)
        s.corrective_limit_x = s.corrective_limit_y = crop
        s.predictive_samples = predictive_samples
        s.crop_to_stable_region = 1 if apply_crop else 0
        s.background[0], s.background[1], s.background[2] = 105, 212, 235
        for k, v in over.items():
            setattr(s, k, v)
        return s


class StabStats(_c.Structure):
    _fields_ = [("tracking_stability", _c.c_float), ("scene_quality", _c.c_float), ("trust", _c.c_float), ("distribution", _c.c_float),
                ("n_detected", _c.c_int), ("n_matched", _c.c_int), ("n_tracked", _c.c_int), ("frame_delay", _c.c_int),
                ("smoothing_factor", _c.c_double), ("homography", _c.c_double * 9)]


class FrameInfo(_c.Structure):
    """lvk_frame_info: geometry and format of an emitted frame."""
    _fields_ = [("rows", _c.c_int), ("cols", _c.c_int), ("format", _c.c_int)]


class Stopwatch:
    """Subset of lvk::Stopwatch the plugin reads: timings().average() / deviation() in milliseconds (Timing/Stopwatch.hpp)."""

    def __init__(self, history=1):
        self.set_history_size(history)

    def set_history_size(self, n):
        self._n = max(1, int(n)); self._hist = []

    def _add(self, seconds):
        self._hist.append(seconds); self._hist = self._hist[-self._n:]

    def average_ms(self):
        return 1e3 * float(np.mean(self._hist)) if self._hist else 0.0

    def deviation_ms(self):
        return 1e3 * float(np.std(self._hist)) if self._hist else 0.0


class StabilizationFilter:
    def __init__(self, settings=None, context=None, device=0):
        self.ctx = context if context is not None else Context(device)
        self.lib = self.ctx.lib
        self._settings = (settings or StabilizationFilterSettings()).copy()
        self._timer = Stopwatch()
        self._borrowed = {}
        h = _c.c_void_p()
        self.ctx._check(self.lib.lvk_hip_stab_create(self.ctx.handle, _c.byref(self._settings), _c.byref(h)))
        self.handle = h
        self._produced = _c.c_int(0); self._ots = _c.c_uint64(0)
        self._produced_ref = _c.byref(self._produced); self._ots_ref = _c.byref(self._ots)

    # ---- Configurable<StabilizationFilterSettings> (Utility/Configurable.hpp:26-44)
    def configure(self, settings):
        self.ctx._check(self.lib.lvk_hip_stab_configure(self.handle, _c.byref(settings)))
        self._settings = settings.copy()

    def reconfigure(self, updater):
        s = self._settings.copy()
        updater(s)
        self.configure(s)

    def settings(self):
        return self._settings

    # ---- VideoFilter (Filters/VideoFilter.hpp:32-61)
    def alias(self):
        return "Stabilization Filter"

    def set_timing_samples(self, n):
        self._timer.set_history_size(n)

    def timings(self):
        return self._timer

    def next_output(self, rows, cols, fmt=FORMAT_YUV):
        """lvk_hip_stab_next_output: (rows, cols, format) of the frame the push of a rows x cols frame will emit -- the DELAYED frame's own
        size (the queue holds whole frames, StabilizationFilter.cpp:118-131) --, or None when that push emits nothing."""
        info = FrameInfo()
        rc = self.lib.lvk_hip_stab_next_output(self.handle, rows, cols, fmt, _c.byref(info))
        if rc < 0:
            self.ctx._check(rc)
        return (info.rows, info.cols, info.format) if rc == 1 else None

    def apply(self, frame, timestamp=0, out=None, profile=False, fmt=FORMAT_YUV):
        """apply(std::move(input), output, profile): returns (output tensor, its timestamp) or (None, None) while the delay builds.
        `frame` is borrowed (not copied) until it has been emitted; do not modify it in the meantime.  The output has the size of the
        DELAYED frame (a stream whose frame size changes emits the queued frames at their own size): `out`, when given, must hold it
        (next_output()), and the returned tensor is its top-left rows x cols view; self.last_format = that frame's format."""
        import torch
        if profile:
            self.ctx.sync()
        t0 = time.perf_counter()
        if out is None:
            due = self.next_output(frame.shape[0], frame.shape[1], fmt)
            out = torch.empty((due[0], due[1], 3), dtype=torch.uint8, device=frame.device) if due else None
        produced = _c.c_int(0); ots = _c.c_uint64(0); released = _c.c_void_p(); info = FrameInfo()
        rc = self.lib.lvk_hip_stab_push(self.handle, frame.data_ptr(), frame.stride(0), frame.shape[0], frame.shape[1],
                                        int(timestamp), fmt, out.data_ptr() if out is not None else None, out.stride(0) if out is not None else 0,
                                        out.shape[0] if out is not None else 0,
                                        _c.byref(produced), _c.byref(ots), _c.byref(released), _c.byref(info))
        self.ctx._check(rc)                                      # (a refused push has queued nothing: the frame is still the caller's)
        self._borrowed[frame.data_ptr()] = frame
        if released.value:
            self._borrowed.pop(released.value, None)
        if profile:
            self.ctx.sync()
        self._timer._add(time.perf_counter() - t0)
        if not produced.value:
            return None, None
        self.last_format = info.format
        return out[:info.rows, :info.cols], ots.value

    def apply_obs(self, fmt, planes, timestamp=0, out=None):
        """The plugin's asynchronous path for any OBS video format (lvk_hip_stab_push_obs): `planes` as Context.ingest_obs takes them, `out` = planes for
        the emitted frame (same format; they must hold the DELAYED frame's size, next_output()).  Returns (out planes, timestamp) or (None, None)."""
        import torch
        t0 = time.perf_counter()
        vf = self.ctx.VIDEO_FORMATS[fmt]
        rows, cols = planes[0].shape[:2]
        if out is None:
            due = self.next_output(rows, cols, self.ctx.obs_frame_format(fmt))
            r, c = (due[0], due[1]) if due else (rows, cols)
            out = [torch.empty((p.shape[0] * r // rows, p.shape[1] * c // cols) + tuple(p.shape[2:]), dtype=torch.uint8, device=p.device) for p in planes]
        iptr, istep = self.ctx._obs_args(planes)
        optr, ostep = self.ctx._obs_args(out)
        produced = _c.c_int(0); ots = _c.c_uint64(0); info = FrameInfo()
        rc = self.lib.lvk_hip_stab_push_obs(self.handle, vf, iptr, istep, rows, cols, int(timestamp), optr, ostep, out[0].shape[0],
                                            _c.byref(produced), _c.byref(ots), _c.byref(info))
        self.ctx._check(rc)
        self._timer._add(time.perf_counter() - t0)
        if not produced.value:
            return None, None
        self.last_format = info.format
        if (info.rows, info.cols) != (out[0].shape[0], out[0].shape[1]):     # a larger buffer was given: the emitted frame is its top-left part
            out = [p[:p.shape[0] * info.rows // out[0].shape[0], :p.shape[1] * info.cols // out[0].shape[1]] for p in out]
        return out, ots.value

    def apply_yuv420(self, planes, timestamp=0, out=None):
        """The OBS async path in one call: planes = (y, u, v) I420 or (y, uv) NV12 torch uint8 tensors on the GPU.
        Returns (output planes, timestamp) or (None, None) while the delay builds.  The output planes have the DELAYED frame's size (a stream
        whose frame size changes still emits its queued frames at their own size): `out`, when given, must hold it (next_output())."""
        import torch
        t0 = time.perf_counter()
        nv12 = len(planes) == 2
        y, u = planes[0], planes[1]
        v = u if nv12 else planes[2]
        if out is None:
            due = self.next_output(y.shape[0], y.shape[1], FORMAT_YUV)
            r, c = (due[0], due[1]) if due else (y.shape[0], y.shape[1])
            if nv12:
                out = (torch.empty((r, c), dtype=torch.uint8, device=y.device), torch.empty((r // 2, c // 2, 2), dtype=torch.uint8, device=y.device))
            else:
                out = (torch.empty((r, c), dtype=torch.uint8, device=y.device),) + tuple(torch.empty((r // 2, c // 2), dtype=torch.uint8, device=y.device) for _ in range(2))
        oy, ou = out[0], out[1]
        ov = ou if nv12 else out[2]
        produced = _c.c_int(0); ots = _c.c_uint64(0); info = FrameInfo()
        rc = self.lib.lvk_hip_stab_push_yuv420(self.handle, y.data_ptr(), y.stride(0), u.data_ptr(), u.stride(0), v.data_ptr(), v.stride(0),
                                               1 if nv12 else 0, y.shape[0], y.shape[1], int(timestamp),
                                               oy.data_ptr(), oy.stride(0), ou.data_ptr(), ou.stride(0), ov.data_ptr(), ov.stride(0), oy.shape[0],
                                               _c.byref(produced), _c.byref(ots), _c.byref(info))
        self.ctx._check(rc)
        self._timer._add(time.perf_counter() - t0)
        if not produced.value:
            return None, None
        if (info.rows, info.cols) != tuple(oy.shape[:2]):                # a larger buffer was given: the emitted frame is its top-left part
            out = (oy[:info.rows, :info.cols],) + tuple(p[:info.rows // 2, :info.cols // 2] for p in out[1:])
        return out, ots.value

    # ---- pre-marshalled arguments: a streaming caller that cycles through a fixed set of buffers converts the tensor addresses
    #      and pitches to ctypes objects once instead of on every frame (about half of the per-call Python cost)
    def prepare_yuv420(self, planes):
        """ctypes argument block of an I420 (y, u, v) / NV12 (y, uv) plane set, for apply_yuv420_prepared (keeps the tensors alive)."""
        nv12 = len(planes) == 2
        y, u = planes[0], planes[1]
        v = u if nv12 else planes[2]
        args = (_c.c_void_p(y.data_ptr()), _c.c_int(y.stride(0)), _c.c_void_p(u.data_ptr()), _c.c_int(u.stride(0)),
                _c.c_void_p(v.data_ptr()), _c.c_int(v.stride(0)))
        return {"args": args, "nv12": _c.c_int(1 if nv12 else 0), "rows": _c.c_int(y.shape[0]), "cols": _c.c_int(y.shape[1]), "planes": planes}

    def apply_yuv420_prepared(self, src, timestamp, dst):
        """apply_yuv420 with argument blocks from prepare_yuv420 (src: input planes, dst: output planes)."""
        produced = self._produced; ots = self._ots
        rc = self.lib.lvk_hip_stab_push_yuv420(self.handle, *src["args"], src["nv12"], src["rows"], src["cols"], timestamp,
                                               *dst["args"], dst["rows"], self._produced_ref, self._ots_ref, None)
        if rc != 0:
            self.ctx._check(rc)
        return (dst["planes"], ots.value) if produced.value else (None, None)

    def prepare_obs(self, fmt, planes):
        """ctypes argument block of one OBS frame's planes for apply_obs_prepared (keeps the tensors alive)."""
        ptrs, steps = self.ctx._obs_args(planes)
        return {"vf": _c.c_int(self.ctx.VIDEO_FORMATS[fmt]), "ptrs": ptrs, "steps": steps, "rows": _c.c_int(planes[0].shape[0]), "cols": _c.c_int(planes[0].shape[1]),
                "planes": planes}

    def apply_obs_prepared(self, src, timestamp, dst):
        """apply_obs with argument blocks from prepare_obs (src: the incoming frame's planes, dst: planes for the emitted frame)."""
        produced = self._produced; ots = self._ots
        rc = self.lib.lvk_hip_stab_push_obs(self.handle, src["vf"], src["ptrs"], src["steps"], src["rows"], src["cols"], timestamp,
                                            dst["ptrs"], dst["steps"], dst["rows"], self._produced_ref, self._ots_ref, None)
        if rc != 0:
            self.ctx._check(rc)
        return (dst["planes"], ots.value) if produced.value else (None, None)

    # ---- host-resident frames (FrameIngest::upload_planes ... download_planes, FrameIngest.cpp:415-474,567-602)
    def host_planes(self, rows, cols, nv12=False):
        """One pinned, CONTIGUOUS 4:2:0 frame (the OBS layout): returns numpy views (y, u, v) / (y, uv) of one lvk_hip_host_malloc block."""
        import numpy as np
        n = rows * cols * 3 // 2
        p = _c.c_void_p()
        self.ctx._check(self.lib.lvk_hip_host_malloc(self.ctx.handle, n, _c.byref(p)))
        self._host_blocks = getattr(self, "_host_blocks", []); self._host_blocks.append(p)
        buf = np.ctypeslib.as_array((_c.c_uint8 * n).from_address(p.value))
        y = buf[:rows * cols].reshape(rows, cols)
        if nv12:
            return y, buf[rows * cols:].reshape(rows // 2, cols // 2, 2)
        q = rows * cols // 4
        return y, buf[rows * cols:rows * cols + q].reshape(rows // 2, cols // 2), buf[rows * cols + q:].reshape(rows // 2, cols // 2)

    def prepare_yuv420_host(self, planes):
        """ctypes argument block of host planes (numpy uint8 arrays in pinned memory) for apply_yuv420_host_prepared."""
        nv12 = len(planes) == 2
        y, u = planes[0], planes[1]
        v = u if nv12 else planes[2]
        args = (_c.c_void_p(y.ctypes.data), _c.c_int(y.strides[0]), _c.c_void_p(u.ctypes.data), _c.c_int(u.strides[0]),
                _c.c_void_p(v.ctypes.data), _c.c_int(v.strides[0]))
        return {"args": args, "nv12": _c.c_int(1 if nv12 else 0), "rows": _c.c_int(y.shape[0]), "cols": _c.c_int(y.shape[1]), "planes": planes}

    def prefetch_yuv420_host_prepared(self, src):
        """lvk_hip_stab_prefetch_yuv420_host: start the upload of the planes the NEXT apply_yuv420_host_prepared call will push."""
        rc = self.lib.lvk_hip_stab_prefetch_yuv420_host(self.handle, *src["args"], src["nv12"], src["rows"], src["cols"])
        if rc != 0:
            self.ctx._check(rc)

    def prefetch(self, frame, fmt=FORMAT_YUV):
        """lvk_hip_stab_prefetch: announce the packed device frame the NEXT apply() will carry (call it before applying the current one)."""
        self.ctx._check(self.lib.lvk_hip_stab_prefetch(self.handle, frame.data_ptr(), frame.stride(0), frame.shape[0], frame.shape[1], fmt))

    def prefetch_yuv420_prepared(self, src):
        """lvk_hip_stab_prefetch_yuv420: announce the device planes the NEXT apply_yuv420_prepared call will carry."""
        rc = self.lib.lvk_hip_stab_prefetch_yuv420(self.handle, *src["args"], src["nv12"], src["rows"], src["cols"])
        if rc != 0:
            self.ctx._check(rc)

    def lookahead_frames(self):
        """lvk_hip_stab_lookahead_frames: pushes so far that found their downscale + pyramid built ahead."""
        return int(self.lib.lvk_hip_stab_lookahead_frames(self.handle))

    def prefetch_cancel(self):
        """lvk_hip_stab_prefetch_cancel: forget the announced frames that have not been pushed."""
        self.ctx._check(self.lib.lvk_hip_stab_prefetch_cancel(self.handle))

    def apply_yuv420_host_prepared(self, src, timestamp, dst):
        """lvk_hip_stab_push_yuv420_host: pinned host planes in, pinned host planes out (complete after Context.sync())."""
        produced = self._produced; ots = self._ots
        rc = self.lib.lvk_hip_stab_push_yuv420_host(self.handle, *src["args"], src["nv12"], src["rows"], src["cols"], timestamp,
                                                    *dst["args"], dst["rows"], self._produced_ref, self._ots_ref, None)
        if rc != 0:
            self.ctx._check(rc)
        return (dst["planes"], ots.value) if produced.value else (None, None)

    # ---- StabilizationFilter (Filters/StabilizationFilter.hpp:46-62)
    def restart(self):
        self.ctx._check(self.lib.lvk_hip_stab_restart(self.handle)); self._borrowed.clear()

    def reset_context(self):
        self.ctx._check(self.lib.lvk_hip_stab_reset_context(self.handle))

    def ready(self):
        return bool(self.lib.lvk_hip_stab_ready(self.handle))

    def frame_delay(self):
        return int(self.lib.lvk_hip_stab_frame_delay(self.handle))

    def stable_region(self, rows, cols):
        r = (_c.c_int * 4)()
        self.ctx._check(self.lib.lvk_hip_stab_stable_region(self.handle, rows, cols, r))
        return tuple(r)

    # ---- taps (tests / HUD)
    def stats(self):
        st = StabStats()
        self.ctx._check(self.lib.lvk_hip_stab_get_stats(self.handle, _c.byref(st)))
        return st

    def detector_frames(self):
        """(frames whose corners went through the suppression grid on the device, frames that took the host loop)."""
        a = _c.c_longlong(0); b = _c.c_longlong(0)
        self.ctx._check(self.lib.lvk_hip_stab_detector_frames(self.handle, _c.byref(a), _c.byref(b)))
        return a.value, b.value

    SCHEDULE = ("push_free_running", "push_synchronised", "ingest_on_tracker", "ingest_on_bulk", "ingest_inline", "remap_persistent", "remap_full",
                "wait_signal_word", "wait_event", "wait_word_timeout")

    def schedule_counters(self, reset=False):
        """lvk_hip_stab_schedule_counters: which schedule the pushes so far took (the library picks per push from what it sees the caller doing:
        free-running -> persistent remap grid + event wait; a caller that waits for every frame -> full grid + host signal word)."""
        a = (_c.c_longlong * len(self.SCHEDULE))()
        self.ctx._check(self.lib.lvk_hip_stab_schedule_counters(self.handle, a, 1 if reset else 0))
        return {k: int(a[i]) for i, k in enumerate(self.SCHEDULE)}

    def meshes(self):
        n = self._settings.motion_width * self._settings.motion_height * 2
        a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
        self.lib.lvk_hip_stab_get_meshes(self.handle, a.ctypes.data_as(_c.POINTER(_c.c_float)), b.ctypes.data_as(_c.POINTER(_c.c_float)), n)
        shp = (self._settings.motion_height, self._settings.motion_width, 2)
        return a.reshape(shp), b.reshape(shp)

    def features(self, cap=8192):
        a = np.zeros((cap, 4), np.float32)
        n = self.lib.lvk_hip_stab_get_features(self.handle, a.ctypes.data_as(_c.POINTER(_c.c_float)), cap)
        return a[:max(n, 0)].copy()

    def draw_trackers(self):
        """StabilizationFilter::draw_trackers: crosses at the tracked features, into the frame of the last apply() (in place)."""
        self.ctx._check(self.lib.lvk_hip_stab_draw_trackers(self.handle))

    def draw_motion_mesh(self):
        """StabilizationFilter::draw_motion_mesh: the motion-mesh grid, into the frame of the last apply() (in place)."""
        self.ctx._check(self.lib.lvk_hip_stab_draw_motion_mesh(self.handle))

    def set_lens(self, params):
        """Fused lens pre-warp: params = (fx, fy, cx, cy, k1, k2, p1, p2, k3) of the plugin's camera profile
        (Modules/OBS-Plugin/Sources/Tools/CCTool.cpp:120-153) or None.  Frames are then pushed RAW; restarts the filter."""
        arr = (ctypes.c_double * 9)(*[float(v) for v in params]) if params is not None else None
        self.ctx._check(self.lib.lvk_hip_stab_set_lens(self.handle, arr))

    def output_stream(self):
        """torch stream the outputs of the following pushes are produced on (chain D2H copies / consumers behind it instead of sync())."""
        import torch
        return torch.cuda.ExternalStream(self.lib.lvk_hip_stab_output_stream(self.handle), device=torch.device("cuda", self.ctx.device))

    def set_overlap(self, enable=True):
        """Run the output remap on a second stream, overlapping the next frame's tracking (output valid after ctx.sync())."""
        self.ctx._check(self.lib.lvk_hip_stab_set_overlap(self.handle, 1 if enable else 0))

    STAGES = ("downscale", "pyramid", "fast", "pyrlk", "motion", "remap", "ingest", "egress")

    def set_profiling(self, enable=True, stages=None, every=1):
        """stages: optional subset of STAGES to time (every timed stage costs two event records per frame on the host);
        every = N: time one push in N."""
        flag = 1 if enable else 0
        if enable and stages is not None:
            flag = 0
            for name in stages:
                flag |= 1 << (self.STAGES.index(name) + 1)
        if enable and every > 1:
            if flag == 1:
                flag = 0
                for i in range(len(self.STAGES)):
                    flag |= 1 << (i + 1)
            flag |= (min(int(every), 255) & 0xff) << 16
        self.ctx._check(self.lib.lvk_hip_stab_set_profiling(self.handle, flag))

    def profile(self):
        """{stage: (total_ms, launches)} measured with HIP events on the launch stream since set_profiling(True)."""
        ms = (_c.c_double * 8)(); n = (_c.c_longlong * 8)()
        self.ctx._check(self.lib.lvk_hip_stab_get_profile(self.handle, ms, n))
        return {k: (ms[i], int(n[i])) for i, k in enumerate(self.STAGES)}

    def close(self):
        # (a filter that outlives its context -- a failed test whose traceback keeps it alive past the session's Context -- must not hand a dangling
        #  context to the library: the handle is dropped, the process is ending anyway)
        alive = getattr(self.ctx, "handle", None)
        if getattr(self, "handle", None):
            if alive:
                self.lib.lvk_hip_stab_destroy(self.handle)
            self.handle = None
        for p in getattr(self, "_host_blocks", []):
            if alive:
                self.lib.lvk_hip_host_free(self.ctx.handle, p)
        self._host_blocks = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
