"""ctypes binding of libhwy_engine.so (the C-ABI of include/hwy_engine.h).

The library is the product: if it is missing this module raises -- there is no
Python/CPU fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _abi
from .build import LIB_PATH

_lib = None

# every symbol include/hwy_engine.h declares
EXPORTS = [
    "hwy_abi_version", "hwy_config_size", "hwy_device_count", "hwy_status_string", "hwy_create",
    "hwy_destroy", "hwy_last_error", "hwy_set_state", "hwy_get_state", "hwy_reset", "hwy_step",
    "hwy_step_device", "hwy_rollout_device", "hwy_rollout", "hwy_step_frames", "hwy_observe", "hwy_set_autoreset", "hwy_sync",
    "hwy_profile_enable", "hwy_profile_read", "hwy_get_prio_turn", "hwy_debug_math", "hwy_get_counters", "hwy_set_block_order",
    "hwy_comm_unique_id", "hwy_comm_init", "hwy_gather", "hwy_comm_destroy",
]


class EngineLibraryMissing(RuntimeError):
    pass


def _share_the_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (same SONAME as /opt/rocm's).  If the engine library is loaded
    first it binds to the system copy, torch later loads its own, and the SECOND runtime in the process finds no GPU ("No HIP
    GPUs are available").  Loading torch's copy first -- when torch is installed at all -- makes both bind to one runtime,
    whatever the import order; engine streams and torch tensors (bench.py, dist.py, vector.py's output="torch") then mix."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # no torch / an unusual layout: the system runtime is used
        pass


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("HWY_ENGINE_LIB", LIB_PATH)  # developer knob: alternative build of the same library
    _share_the_hip_runtime_with_torch()
    if not os.path.exists(path):
        raise EngineLibraryMissing(
            f"{path} not found. Build it with `python -m highwayenv_amd.build` (needs hipcc, "
            "cross-compiles for gfx950). The MI355X engine has no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32, u8p, f64 = C.c_void_p, C.c_int32, C.POINTER(C.c_uint8), C.c_double
    lib.hwy_abi_version.restype = C.c_int
    lib.hwy_config_size.restype = C.c_size_t
    lib.hwy_device_count.restype = C.c_int
    lib.hwy_status_string.restype = C.c_char_p
    lib.hwy_status_string.argtypes = [C.c_int]
    lib.hwy_last_error.restype = C.c_char_p
    lib.hwy_last_error.argtypes = [vp]
    lib.hwy_create.argtypes = [C.POINTER(_abi.HwyConfig), C.c_int, vp, C.POINTER(vp)]
    lib.hwy_destroy.argtypes = [vp]
    lib.hwy_set_state.argtypes = [vp, C.POINTER(_abi.HwyState)]
    lib.hwy_get_state.argtypes = [vp, C.POINTER(_abi.HwyState)]
    lib.hwy_reset.argtypes = [vp, vp, vp, f64, f64, i32, vp]
    lib.hwy_step.argtypes = [vp] + [vp] * 7
    lib.hwy_step_device.argtypes = [vp] + [vp] * 7
    lib.hwy_rollout_device.argtypes = [vp, i32] + [vp] * 7
    lib.hwy_rollout.argtypes = [vp, i32] + [vp] * 7
    lib.hwy_step_frames.argtypes = [vp, vp, i32]
    lib.hwy_observe.argtypes = [vp, vp]
    lib.hwy_set_autoreset.argtypes = [vp, i32, C.c_uint64, f64, f64, i32]
    lib.hwy_sync.argtypes = [vp]
    lib.hwy_debug_math.argtypes = [vp, i32, vp, vp, C.c_int64]
    lib.hwy_get_counters.argtypes = [vp, C.POINTER(C.c_uint64), i32, i32]
    lib.hwy_get_counters.restype = C.c_int
    lib.hwy_set_block_order.argtypes = [vp, vp]
    lib.hwy_set_block_order.restype = C.c_int
    lib.hwy_comm_unique_id.argtypes = [vp]
    lib.hwy_comm_init.argtypes = [vp, vp, i32, i32]
    lib.hwy_gather.argtypes = [vp, vp, vp, C.c_size_t, i32]
    lib.hwy_comm_destroy.argtypes = [vp]
    lib.hwy_profile_enable.argtypes = [vp, i32]
    lib.hwy_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.hwy_get_prio_turn.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.hwy_get_prio_turn.restype = C.c_int
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int or name.startswith(("hwy_create", "hwy_destroy", "hwy_set", "hwy_get", "hwy_reset",
                                                      "hwy_step", "hwy_rollout", "hwy_observe", "hwy_sync", "hwy_profile", "hwy_debug", "hwy_comm", "hwy_gather")):
            if name not in ("hwy_status_string", "hwy_last_error", "hwy_config_size"):
                fn.restype = C.c_int
    if lib.hwy_abi_version() != _abi.HWY_ABI_VERSION:
        raise RuntimeError("libhwy_engine.so ABI version mismatch; rebuild with python -m highwayenv_amd.build")
    if lib.hwy_config_size() != C.sizeof(_abi.HwyConfig):
        raise RuntimeError("hwy_config layout mismatch between include/hwy_engine.h and highwayenv_amd/_abi.py")
    _lib = lib
    return lib
