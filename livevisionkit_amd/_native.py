"""ctypes loader for liblvk_hip.so (the HIP kernels + C-ABI declared in include/lvk_hip.h).

There is deliberately NO fallback: if the extension is missing or cannot run, importing callers fail
loudly.  torch is imported first so that liblvk_hip.so binds to the HIP runtime torch already loaded
(both export the soname libamdhip64.so.7) -- two HIP runtimes in one process cannot share pointers.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LVK_HIP_LIB") or os.path.join(_HERE, "liblvk_hip.so")      # override: kernel-variant experiments only

# every symbol include/lvk_hip.h declares: (name, restype, argtypes)
_c = ctypes
_P = _c.c_void_p
_SIG = {
    "lvk_hip_ctx_create": (_c.c_int, [_c.c_int, _c.POINTER(_P)]),
    "lvk_hip_ctx_create_on_stream": (_c.c_int, [_c.c_int, _P, _c.POINTER(_P)]),
    "lvk_hip_ctx_destroy": (None, [_P]),
    "lvk_hip_sync": (_c.c_int, [_P]),
    "lvk_hip_stream": (_P, [_P]),
    "lvk_hip_last_error": (_c.c_char_p, [_P]),
    "lvk_hip_version": (_c.c_char_p, []),
    "lvk_hip_abi_version": (_c.c_int, []),
    "lvk_hip_device_count": (_c.c_int, []),
    "lvk_hip_device_usable": (_c.c_int, [_c.c_int]),
    "lvk_hip_malloc": (_c.c_int, [_P, _c.c_size_t, _c.POINTER(_P)]),
    "lvk_hip_free": (_c.c_int, [_P, _P]),
    "lvk_hip_trim": (_c.c_int, [_P]),
    "lvk_hip_ctx_wait": (_c.c_int, [_P, _P]),
    "lvk_hip_upload": (_c.c_int, [_P, _P, _P, _c.c_size_t]),
    "lvk_hip_download": (_c.c_int, [_P, _P, _P, _c.c_size_t]),
    "lvk_hip_remap_homography": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int,
                                            _c.c_int, _c.c_int, _c.POINTER(_c.c_float), _c.POINTER(_c.c_uint8), _c.c_int]),
    "lvk_hip_remap_mesh": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int,
                                      _c.POINTER(_c.c_float), _c.c_int, _c.c_int, _c.POINTER(_c.c_uint8), _c.c_int]),
    "lvk_hip_remap_map": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.POINTER(_c.c_uint8), _c.c_int]),
    "lvk_hip_upscale": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "lvk_hip_sharpen": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _c.c_float]),
    "lvk_hip_native_rcp": (_c.c_int, [_P, _P, _P, _c.c_size_t]),
    "lvk_hip_mesh_solver_create": (_c.c_int, [_P, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _c.c_int, _c.POINTER(_P)]),
    "lvk_hip_mesh_solver_destroy": (None, [_P]),
    "lvk_hip_mesh_solver_reset": (_c.c_int, [_P]),
    "lvk_hip_mesh_solver_solve": (_c.c_int, [_P, _c.POINTER(_c.c_float), _c.POINTER(_c.c_float), _c.c_int, _c.c_float, _c.c_float, _c.c_float, _c.c_float,
                                             _c.POINTER(_c.c_uint8), _c.POINTER(_c.c_float)]),
    "lvk_hip_lens_map_create": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.POINTER(_P), _c.POINTER(_c.c_int)]),
    "lvk_hip_lens_map_destroy": (_c.c_int, [_P, _P]),
    "lvk_hip_warpmesh_apply_lens": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _c.POINTER(_c.c_float), _c.c_int, _c.c_int, _c.POINTER(_c.c_uint8), _c.c_int, _P]),
    "lvk_hip_lens_undistort_points": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.POINTER(_c.c_float), _c.c_int, _c.POINTER(_c.c_float)]),
    "lvk_hip_stab_set_lens": (_c.c_int, [_P, _P]),
    "lvk_hip_stab_output_stream": (_P, [_P]),
    "lvk_hip_fast_filter": (_c.c_int, [_P, _c.POINTER(_c.c_float), _c.POINTER(_c.c_float), _c.POINTER(_c.c_uint8), _c.c_int, _c.POINTER(_c.c_float), _c.POINTER(_c.c_float)]),
    "lvk_hip_stab_draw_trackers": (_c.c_int, [_P]),
    "lvk_hip_stab_draw_motion_mesh": (_c.c_int, [_P]),
    "lvk_hip_draw_grid": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_uint8), _c.c_int]),
    "lvk_hip_draw_crosses": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_float), _c.c_int, _c.c_float, _c.c_float,
                                        _c.POINTER(_c.c_uint8), _c.c_int, _c.c_int]),
    "lvk_hip_warpmesh_apply_yuv420": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int,
                                                 _c.POINTER(_c.c_float), _c.c_int, _c.c_int, _c.POINTER(_c.c_uint8)]),
    "lvk_hip_warpmesh_apply": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int,
                                          _c.POINTER(_c.c_float), _c.c_int, _c.c_int, _c.POINTER(_c.c_uint8), _c.c_int]),
    "lvk_hip_luma_area_resize": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int]),
    "lvk_hip_pyr_down": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int]),
    "lvk_hip_scharr": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P]),
    "lvk_hip_build_pyramid": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                         _c.POINTER(_c.c_uint8), _c.POINTER(_c.c_int16), _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "lvk_hip_fast_detect": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_int), _c.c_int,
                                       _c.POINTER(_c.c_uint32), _c.c_int, _c.POINTER(_c.c_int)]),
    "lvk_hip_pyrlk": (_c.c_int, [_P, _P, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_float), _c.c_int,
                                 _c.POINTER(_c.c_float), _c.POINTER(_c.c_uint8), _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                 _c.c_double, _c.c_double]),
    "lvk_hip_ingest_yuv420": (_c.c_int, [_P, _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int]),
    "lvk_hip_egress_yuv420": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int]),
    "lvk_hip_ingest_obs": (_c.c_int, [_P, _c.c_int, _P * 3, _c.c_int * 3, _c.c_int, _c.c_int, _P, _c.c_int]),
    "lvk_hip_egress_obs": (_c.c_int, [_P, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _P * 3, _c.c_int * 3]),
    "lvk_hip_obs_frame_format": (_c.c_int, [_c.c_int]),
    "lvk_hip_estimate_global_motion": (_c.c_int, [_P, _c.POINTER(_c.c_float), _c.POINTER(_c.c_float), _c.c_int, _c.c_double,
                                                  _c.c_double, _c.c_double, _c.c_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_uint8)]),
    "lvk_stab_default_settings": (None, [_P]),
    "lvk_hip_stab_create": (_c.c_int, [_P, _P, _c.POINTER(_P)]),
    "lvk_hip_stab_destroy": (None, [_P]),
    "lvk_hip_stab_configure": (_c.c_int, [_P, _P]),
    "lvk_hip_stab_restart": (_c.c_int, [_P]),
    "lvk_hip_stab_reset_context": (_c.c_int, [_P]),
    "lvk_hip_stab_ready": (_c.c_int, [_P]),
    "lvk_hip_stab_frame_delay": (_c.c_int, [_P]),
    "lvk_hip_stab_stable_region": (_c.c_int, [_P, _c.c_int, _c.c_int, _c.POINTER(_c.c_int)]),
    "lvk_hip_stab_next_output": (_c.c_int, [_P, _c.c_int, _c.c_int, _c.c_int, _P]),
    "lvk_hip_stab_push": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_uint64, _c.c_int, _P, _c.c_int, _c.c_int,
                                     _c.POINTER(_c.c_int), _c.POINTER(_c.c_uint64), _c.POINTER(_P), _P]),
    "lvk_hip_stab_push_yuv420": (_c.c_int, [_P, _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_uint64,
                                            _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int, _c.POINTER(_c.c_int), _c.POINTER(_c.c_uint64), _P]),
    "lvk_hip_stab_push_obs": (_c.c_int, [_P, _c.c_int, _P * 3, _c.c_int * 3, _c.c_int, _c.c_int, _c.c_uint64, _P * 3, _c.c_int * 3, _c.c_int,
                                         _c.POINTER(_c.c_int), _c.POINTER(_c.c_uint64), _P]),
    "lvk_hip_stab_push_yuv420_host": (_c.c_int, [_P, _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_uint64,
                                            _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int, _c.POINTER(_c.c_int), _c.POINTER(_c.c_uint64), _P]),
    "lvk_hip_stab_prefetch_yuv420_host": (_c.c_int, [_P, _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "lvk_hip_stab_prefetch_cancel": (_c.c_int, [_P]),
    "lvk_hip_stab_prefetch": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "lvk_hip_stab_prefetch_yuv420": (_c.c_int, [_P, _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "lvk_hip_stab_lookahead_frames": (_c.c_longlong, [_P]),
    "lvk_hip_stab_schedule_counters": (_c.c_int, [_P, _c.POINTER(_c.c_longlong), _c.c_int]),
    "lvk_hip_stab_detector_frames": (_c.c_int, [_P, _c.POINTER(_c.c_longlong), _c.POINTER(_c.c_longlong)]),
    "lvk_hip_host_malloc": (_c.c_int, [_P, _c.c_size_t, _c.POINTER(_P)]),
    "lvk_hip_host_free": (_c.c_int, [_P, _P]),
    "lvk_hip_stab_get_stats": (_c.c_int, [_P, _P]),
    "lvk_hip_stab_get_meshes": (_c.c_int, [_P, _c.POINTER(_c.c_float), _c.POINTER(_c.c_float), _c.c_int]),
    "lvk_hip_stab_get_features": (_c.c_int, [_P, _c.POINTER(_c.c_float), _c.c_int]),
    "lvk_hip_stab_set_overlap": (_c.c_int, [_P, _c.c_int]),
    "lvk_hip_stab_set_bulk_context": (_c.c_int, [_P, _P]),
    "lvk_hip_stab_set_profiling": (_c.c_int, [_P, _c.c_int]),
    "lvk_hip_stab_get_profile": (_c.c_int, [_P, _c.POINTER(_c.c_double), _c.POINTER(_c.c_longlong)]),
}

_lib = None


def symbols():
    return sorted(_SIG.keys())


def load():
    """Load liblvk_hip.so and bind every declared symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C livevisionkit_amd/csrc`). There is no CPU fallback.")
    import torch  # noqa: F401  (loads the process-wide HIP runtime first)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in _SIG.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
