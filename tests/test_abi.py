"""The C-ABI shared library: loads, exports every symbol include/hwy_engine.h declares, agrees with the
ctypes mirror on struct layout, and refuses to run without a GPU (no CPU fallback).  No compute calls."""
import ctypes as C
import os
import re

import pytest

from highwayenv_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "hwy_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # strip comments
    return sorted(set(re.findall(r"\b(hwy_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hwy_engine.h but not exported"
    assert sorted(_lib.EXPORTS) == names


def test_struct_layout_and_version_agree():
    lib = _lib.load()
    assert lib.hwy_abi_version() == _abi.HWY_ABI_VERSION
    assert lib.hwy_config_size() == C.sizeof(_abi.HwyConfig)
    assert lib.hwy_status_string(_abi.HWY_ERR_NO_DEVICE).decode().startswith("no MI355X")


def test_constants_match_header():
    text = open(os.path.join(ROOT, "include", "hwy_engine.h")).read()
    for name, val in [("HWY_MAX_AGENTS", _abi.HWY_MAX_AGENTS), ("HWY_MAX_FEATURES", _abi.HWY_MAX_FEATURES),
                      ("HWY_MAX_TARGET_SPEEDS", _abi.HWY_MAX_TARGET_SPEEDS), ("HWY_MAX_LANES", _abi.HWY_MAX_LANES),
                      ("HWY_MAX_VEHICLES", _abi.HWY_MAX_VEHICLES), ("HWY_ABI_VERSION", _abi.HWY_ABI_VERSION)]:
        assert re.search(rf"#define {name} {val}\b", text), name
    for name, val in [("HWY_F_CRASHED", 1), ("HWY_F_HAS_IMPACT", 2), ("HWY_F_CHECK_COLLISIONS", 4), ("HWY_F_CONTROLLED", 8),
                      ("HWY_C_NORMALIZE_REWARD", 1), ("HWY_C_OFFROAD_TERMINAL", 2), ("HWY_C_OBS_ABSOLUTE", 4),
                      ("HWY_C_OBS_NORMALIZE", 8), ("HWY_C_OBS_CLIP", 16), ("HWY_C_OBS_SEE_BEHIND", 32),
                      ("HWY_C_EGO_ONLY_COLLISIONS", 64)]:
        assert re.search(rf"{name} = {val}\b", text), name
    feats = re.search(r"HWY_FEAT_PRESENCE = 0,(.*?)HWY_FEAT_COUNT", text, re.S).group(1)
    order = ["presence"] + [m.lower() for m in re.findall(r"HWY_FEAT_([A-Z_]+)", feats)]
    assert order == [k for k, _ in sorted(_abi.FEATURE_IDS.items(), key=lambda kv: kv[1])]


def test_create_without_gpu_fails_loudly():
    lib = _lib.load()
    if lib.hwy_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = _abi.make_config(_abi.highway_fast_default_config(), 2, fast=True)
    h = C.c_void_p()
    rc = lib.hwy_create(C.byref(cfg), 0, None, C.byref(h))
    assert rc == _abi.HWY_ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in lib.hwy_last_error(None)


def test_invalid_config_is_rejected_before_touching_the_device():
    lib = _lib.load()
    cfg = _abi.make_config(_abi.highway_fast_default_config(), 2, fast=True)
    cfg.abi_version = 99
    h = C.c_void_p()
    assert lib.hwy_create(C.byref(cfg), 0, None, C.byref(h)) == _abi.HWY_ERR_INVALID_ARG
    assert b"abi_version" in lib.hwy_last_error(None)
