"""TEST INFRASTRUCTURE: ctypes wrapper around oracle/_build/libhwy_oracle.so.

The oracle is the checker for the HIP engine (and bench.py's timed CPU baseline);
it is never imported by the product package ``highwayenv_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from highwayenv_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhwy_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (no GPU needed)."""
    src = os.path.join(_HERE, "hwy_oracle.c")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "hwy_engine.h")
    src_net = os.path.join(_HERE, "hwy_oracle_net.c")
    src_ix = os.path.join(_HERE, "hwy_oracle_ix.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in (src, src_net, src_ix, hdr)))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_config_size.restype = C.c_size_t
        assert _lib.orc_config_size() == C.sizeof(_abi.HwyConfig), "hwy_config layout mismatch"
        for fn in (_lib.orc_frames, _lib.orc_observe, _lib.orc_step):
            fn.restype = C.c_int
    return _lib


def _p(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def frames(cfg: _abi.HwyConfig, st: dict, actions, n_frames: int) -> None:
    """n_frames x {[meta-action]; Road.act(); Road.step(dt)} in place on the SoA dict."""
    acts = None if actions is None else np.ascontiguousarray(actions, np.int32)
    s = _abi.state_struct(st)
    rc = lib().orc_frames(C.byref(cfg), C.byref(s), _p(acts, C.c_int32), C.c_int32(n_frames))
    assert rc == 0, rc


def net_neighbours(cfg: _abi.HwyConfig, st: dict, e: int, slot: int, lane) -> tuple:
    """Road.neighbour_vehicles(vehicle, lane_index) on a road-network scenario: (front slot | None, rear slot | None).
    ``lane=None`` is the reference's ``lane_index=None``: the vehicle's own lane, or (None, None) when it has none (< 0)."""
    lane = -1 if lane is None else lane
    f, b = C.c_int32(-1), C.c_int32(-1)
    s = _abi.state_struct(st)
    rc = lib().orc_net_neighbours(C.byref(cfg), C.byref(s), C.c_int32(e), C.c_int32(slot), C.c_int32(lane),
                                  C.byref(f), C.byref(b))
    assert rc == 0, rc
    return (None if f.value < 0 else f.value), (None if b.value < 0 else b.value)


def observe(cfg: _abi.HwyConfig, st: dict) -> np.ndarray:
    obs = np.zeros((cfg.num_envs, cfg.num_agents, *_abi.obs_shape(cfg)), np.float32)
    s = _abi.state_struct(st)
    rc = lib().orc_observe(C.byref(cfg), C.byref(s), _p(obs, C.c_float))
    assert rc == 0, rc
    return obs


def step(cfg: _abi.HwyConfig, st: dict, actions) -> tuple:
    """AbstractEnv.step for every env; returns (obs, reward, terminated, truncated, info)."""
    E, A = cfg.num_envs, cfg.num_agents
    acts = np.ascontiguousarray(np.asarray(actions, np.int32).reshape(E, A))
    obs = np.zeros((E, A, *_abi.obs_shape(cfg)), np.float32)
    reward = np.zeros((E, A), np.float64)
    term = np.zeros(E, np.uint8)
    trunc = np.zeros(E, np.uint8)
    speed = np.zeros((E, A), np.float64)
    crashed = np.zeros((E, A), np.uint8)
    s = _abi.state_struct(st)
    rc = lib().orc_step(C.byref(cfg), C.byref(s), _p(acts, C.c_int32), _p(obs, C.c_float),
                        _p(reward, C.c_double), _p(term, C.c_uint8), _p(trunc, C.c_uint8),
                        _p(speed, C.c_double), _p(crashed, C.c_uint8))
    if rc == _abi.HWY_ERR_ACTION:
        raise KeyError("invalid meta-action")
    assert rc == 0, rc
    return obs, reward, term.astype(bool), trunc.astype(bool), {"speed": speed, "crashed": crashed.astype(bool)}


class impact_margins:
    """Context manager (test diagnostics): while active, every ``frames`` / ``step`` call of the oracle (all three
    families: straight road, merge networks, and -- with an ``oracle_ix.IxConfig`` -- the intersection, indexed by list position) fills ``self.margin`` [E, N] with the smallest ``|d . normal|`` (utils.py:232-236) among the impacts assigned to each
    vehicle during that call (+inf where none was).  A margin at rounding-noise level marks a collision whose push
    direction is decided by the last bit of the libm in use (two cars on one lane centre, lateral axis)."""

    def __init__(self, cfg: _abi.HwyConfig):
        n = cfg.num_vehicles if hasattr(cfg, "num_vehicles") else cfg.n_slots
        self.margin = np.full((cfg.num_envs, n), np.inf)
        # smallest |interval distance| behind an `intersecting` / `will_intersect` decision (utils.py:222-229) of any SAT the
        # vehicle took part in: ~0 once a wreck, pushed back by its impact, rests exactly touching what it hit
        self.flag_margin = np.full((cfg.num_envs, n), np.inf)

    def __enter__(self):
        lib().orc_set_margin_buffer(self.margin.ctypes.data_as(C.POINTER(C.c_double)))
        lib().orc_set_flag_margin_buffer(self.flag_margin.ctypes.data_as(C.POINTER(C.c_double)))
        return self

    def __exit__(self, *exc):
        lib().orc_set_margin_buffer(None)
        lib().orc_set_flag_margin_buffer(None)
        return False

    def well(self, knife: float = 1e-9) -> np.ndarray:
        """[E]: every collision decision of the call is well conditioned (push direction AND flags)."""
        return (self.margin.min(1) >= knife) & (self.flag_margin.min(1) >= knife)


class knife_bias:
    """Context manager (test diagnostics, road-network oracle): the SAT's two `distance > 0` decisions (utils.py:222-229) taken as
    `distance > bias` -- the answer the reference gives when a distance at rounding level (|d| < |bias|) rounds to the other side."""

    def __init__(self, bias: float):
        self.bias = float(bias)

    def __enter__(self):
        lib().orc_set_knife_bias(C.c_double(self.bias))
        return self

    def __exit__(self, *exc):
        lib().orc_set_knife_bias(C.c_double(0.0))
        return False
