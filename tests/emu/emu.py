"""TEST INFRASTRUCTURE ONLY: run the product's kernel source on the CPU (tests/emu/hip_emu.h).

``EmuEngine`` has the same Python surface as ``highwayenv_amd.engine.Engine`` so that the
parity tests can be written once and run against the CPU emulation here (no GPU in the build
container) and against the real HIP engine on the MI355X (``-m gpu``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from highwayenv_amd import _abi, build as build_flags

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
# HWY_EMU_FLAGS: extra g++ flags for the emulator build (its own library file), e.g. "-ffp-contract=fast -mfma" -- how the test
# suite's tolerances were tried against fused multiply-adds before the kernel build's -ffp-contract=off was put up for an A/B
# (profiles/r03_history.md).  The default build rounds every a*b+c twice, like the kernel build and like numpy.
_EXTRA = os.environ.get("HWY_EMU_FLAGS", "").split()
_LIB = os.path.join(_HERE, "_build", "libhwy_emu.so" if not _EXTRA else
                    "libhwy_emu_%08x.so" % (__import__("zlib").crc32(" ".join(_EXTRA).encode()) & 0xffffffff))
_lib = None


def compile_emulator(src_cpp: str, out_lib: str, extra=()) -> None:
    """One emulator library from `src_cpp` (emu_engine.cpp of this tree, or of a mutated copy: tests/test_mutations.py)."""
    # like the kernel build (build.FP_CONTRACT).  "on" = contraction by source expression is a FRONT-END rule: the emulator
    # is compiled by the clang the kernels are compiled by (ROCm's), so the same a*b+c fuse here and on the GPU; g++ 11
    # treats -ffp-contract=on as off and is only the fallback (then: fuse where it likes, same tolerances).
    clang = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang++")
    cxx = "g++"
    if build_flags.FP_CONTRACT == "off":
        contract = ["-ffp-contract=off"]
    elif build_flags.FP_CONTRACT == "on" and os.path.exists(clang):
        cxx, contract = clang, ["-ffp-contract=on", "-mfma"]
    else:
        contract = ["-ffp-contract=fast", "-mfma"]
    tmp = f"{out_lib}.{os.getpid()}.tmp"   # pytest-xdist workers may build at once: never expose a half-written library
    subprocess.run([cxx, "-std=c++20", "-O1", "-pthread", "-fPIC", "-shared", *contract, *extra, "-o", tmp, src_cpp],
                   check=True, capture_output=True)
    os.replace(tmp, out_lib)


def build(force: bool = False) -> str:
    if os.environ.get("HWY_EMU_LIB"):  # a prebuilt (mutated) emulator: tests/test_mutations.py
        return os.environ["HWY_EMU_LIB"]
    srcs = [os.path.join(_HERE, "emu_engine.cpp"), os.path.join(_HERE, "hip_emu.h"),
            os.path.join(_ROOT, "highwayenv_amd", "csrc", "hwy_device.h"),
            os.path.join(_ROOT, "highwayenv_amd", "csrc", "hwy_wave.h"),
            os.path.join(_ROOT, "highwayenv_amd", "csrc", "hwy_wave2.h"),
            os.path.join(_ROOT, "highwayenv_amd", "csrc", "hwy_net.h"),
            os.path.join(_ROOT, "highwayenv_amd", "csrc", "hwy_ix.h"),
            os.path.join(_ROOT, "highwayenv_amd", "csrc", "hwy_math.h"),
            os.path.join(_ROOT, "highwayenv_amd", "csrc", "hwy_params.h"),
            os.path.join(_ROOT, "include", "hwy_engine.h")]
    stale = not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs)
    if force or stale:
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        compile_emulator(srcs[0], _LIB, _EXTRA)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.emu_config_size.restype = C.c_size_t
        assert _lib.emu_config_size() == C.sizeof(_abi.HwyConfig)
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class EmuEngine:
    def __init__(self, cfg: _abi.HwyConfig):
        self.cfg = cfg
        self.E, self.N, self.A = cfg.num_envs, cfg.num_vehicles, cfg.num_agents
        self.ix = cfg.scenario == _abi.SCENARIO_INTERSECTION
        self.st = _abi.alloc_state_ix(self.E, self.N) if self.ix else _abi.alloc_state(self.E, self.N)
        self.done = np.zeros(self.E, np.uint8)
        self.episode = np.zeros(self.E, np.uint32)
        self.autoreset = (0, 0, 2.0, 1.0, -1)
        if self.ix:  # next-episode pre-warming buffers (hwy_engine.hip allocates the same on the device)
            self.shadow = (np.zeros((9, self.E, self.N)), np.zeros((self.E, self.N), np.int32),
                           np.zeros((self.E, self.N), np.int64), np.full((self.E, 4), -1, np.int32))

    def _bind_shadow(self):
        if self.ix:
            f, pk, rt, meta = self.shadow
            lib().emu_set_shadow(_p(f, C.c_double), _p(pk, C.c_int32), _p(rt, C.c_int64), _p(meta, C.c_int32))
        else:
            lib().emu_set_shadow(None, None, None, None)

    def close(self):
        pass

    def set_state(self, st):
        self.st = _abi.copy_state(st)
        self.done[:] = 0

    def get_state(self):
        st = _abi.copy_state(self.st)
        if self.ix:  # like hwy_get_state: the planes of an empty slot (and an impact without its flag) are unspecified: zeros
            absent = (st["flags"] & _abi.F_ABSENT) != 0
            for k in _abi.STATE_F64 + ["lane", "target_lane", "speed_index", "route"]:
                st[k][absent] = 0
            st["flags"][absent] = _abi.F_ABSENT
            no_impact = (st["flags"] & _abi.F_HAS_IMPACT) == 0
            st["impact_x"][no_impact] = 0
            st["impact_y"][no_impact] = 0
        return st

    def set_autoreset(self, enabled, base_seed=0, ego_spacing=2.0, vehicles_density=1.0, initial_lane_id=-1):
        if self.ix and int(base_seed) != self.autoreset[1]:
            self.shadow[3][...] = -1
        self.autoreset = (int(enabled), int(base_seed), float(ego_spacing), float(vehicles_density), int(initial_lane_id))

    def set_block_order(self, env_of_block=None):
        self._block_env = None if env_of_block is None else np.ascontiguousarray(env_of_block, np.uint16)

    def _run(self, mode, n_frames, actions):
        E, A = self.E, self.A
        acts = None if actions is None else np.ascontiguousarray(np.asarray(actions, np.int32).reshape(E, A))
        obs = np.zeros((E, A, *_abi.obs_shape(self.cfg)), np.float32)
        reward = np.zeros((E, A))
        term = np.zeros(E, np.uint8)
        trunc = np.zeros(E, np.uint8)
        speed = np.zeros((E, A))
        crashed = np.zeros((E, A), np.uint8)
        s = _abi.state_struct(self.st)
        ar = self.autoreset
        self._bind_shadow()
        be = getattr(self, "_block_env", None)
        lib().emu_set_block_order(None if be is None else be.ctypes.data_as(C.c_void_p))
        rc = lib().emu_run(C.byref(self.cfg), C.byref(s), _p(self.done, C.c_uint8), _p(self.episode, C.c_uint32),
                           C.c_int(mode), C.c_int(n_frames), _p(acts, C.c_int32), _p(obs, C.c_float),
                           _p(reward, C.c_double), _p(term, C.c_uint8), _p(trunc, C.c_uint8), _p(speed, C.c_double),
                           _p(crashed, C.c_uint8), C.c_int(ar[0]), C.c_uint64(ar[1]), C.c_double(ar[2]),
                           C.c_double(ar[3]), C.c_int(ar[4]))
        assert rc == 0
        return obs, reward, term.astype(bool), trunc.astype(bool), {"speed": speed, "crashed": (crashed & 1).astype(bool),
                                                                    "arrived": (crashed & 2).astype(bool)}

    def step_frames(self, actions, n_frames):
        self._run(0, n_frames, actions)

    def step(self, actions):
        a = np.asarray(actions)
        if ((a < 0) | (a > (2 if self.ix else _abi.num_actions(self.cfg) - 1))).any():
            raise KeyError("invalid meta-action")
        return self._run(1, self.cfg.frames_per_step, actions)

    def observe(self):
        return self._run(2, 0, None)[0]

    def rollout(self, actions):
        """hwy_rollout_device: actions [K, E, A] -> (obs [K, E, A, ...], reward [K, E, A], terminated [K, E], truncated [K, E],
        info) -- ONE multi-step launch where the engine has one (the one-wavefront kernel), else K steps."""
        acts = np.ascontiguousarray(np.asarray(actions, np.int32).reshape(-1, self.E, self.A))
        K = acts.shape[0]
        if not lib().emu_has_rollout_kernel(C.byref(self.cfg)):
            outs = [self.step(acts[k]) for k in range(K)]
            info = {key: np.stack([o[4][key] for o in outs]) for key in outs[0][4]}
            return tuple(np.stack([o[j] for o in outs]) for j in range(4)) + (info,)
        E, A = self.E, self.A
        obs = np.zeros((K, E, A, *_abi.obs_shape(self.cfg)), np.float32)
        reward = np.zeros((K, E, A))
        term, trunc = np.zeros((K, E), np.uint8), np.zeros((K, E), np.uint8)
        speed, crashed = np.zeros((K, E, A)), np.zeros((K, E, A), np.uint8)
        s = _abi.state_struct(self.st)
        ar = self.autoreset
        self._bind_shadow()
        lib().emu_set_rollout(C.c_int(K))
        try:
            rc = lib().emu_run(C.byref(self.cfg), C.byref(s), _p(self.done, C.c_uint8), _p(self.episode, C.c_uint32),
                               C.c_int(1), C.c_int(self.cfg.frames_per_step), _p(acts, C.c_int32), _p(obs, C.c_float),
                               _p(reward, C.c_double), _p(term, C.c_uint8), _p(trunc, C.c_uint8), _p(speed, C.c_double),
                               _p(crashed, C.c_uint8), C.c_int(ar[0]), C.c_uint64(ar[1]), C.c_double(ar[2]),
                               C.c_double(ar[3]), C.c_int(ar[4]))
        finally:
            lib().emu_set_rollout(C.c_int(0))
        assert rc == 0
        return obs, reward, term.astype(bool), trunc.astype(bool), {"speed": speed, "crashed": (crashed & 1).astype(bool),
                                                                    "arrived": (crashed & 2).astype(bool)}

    def debug_math(self, op, x):
        xin = np.ascontiguousarray(x, np.float64).ravel()
        out = np.empty_like(xin)
        lib().emu_debug_math(C.c_int(op), _p(xin, C.c_double), _p(out, C.c_double), C.c_longlong(xin.size))
        return out.reshape(np.shape(x))

    def reset(self, seeds=None, mask=None, ego_spacing=2.0, vehicles_density=1.0, initial_lane_id=-1, base_seed=0):
        E, A = self.E, self.A
        obs = np.zeros((E, A, *_abi.obs_shape(self.cfg)), np.float32)
        sd = None if seeds is None else np.ascontiguousarray(seeds, np.uint64)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        s = _abi.state_struct(self.st)
        self._bind_shadow()
        rc = lib().emu_reset(C.byref(self.cfg), C.byref(s), _p(self.done, C.c_uint8), _p(self.episode, C.c_uint32),
                             _p(mk, C.c_uint8), _p(sd, C.c_uint64), C.c_uint64(base_seed), C.c_double(ego_spacing),
                             C.c_double(vehicles_density), C.c_int(initial_lane_id), _p(obs, C.c_float))
        assert rc == 0
        return obs


def force_block_kernel(on: bool):
    """Use the generic workgroup kernel even for N <= 64 (the engine's HWY_STEP_KERNEL=block)."""
    lib().emu_force_block_kernel(C.c_int(int(on)))


def philox_uniform2(seed, vehicle, episode, draw):
    u0, u1 = C.c_double(), C.c_double()
    lib().emu_philox_uniform2(C.c_uint64(seed), C.c_uint32(vehicle), C.c_uint32(episode), C.c_uint32(draw),
                              C.byref(u0), C.byref(u1))
    return u0.value, u1.value
