"""livevisionkit_amd -- MI355X-native implementation of LiveVisionKit's video stabilization hot path.

The product is liblvk_hip.so (hand-written HIP kernels for gfx950 behind the C-ABI of include/lvk_hip.h) plus
the C++ facade in include/lvk/.  This Python package is the thin host-side mirror used by the tests and the
bench driver: it calls the C-ABI through ctypes and uses torch only for device memory and streams.
"""
from . import _native
from .context import Context, LvkHipError
from .stabilization import StabilizationFilter, StabilizationFilterSettings
from . import shard

__all__ = ["Context", "LvkHipError", "StabilizationFilter", "StabilizationFilterSettings", "_native"]
