"""Thin object wrapper over the C-ABI (include/hwy_engine.h): one ``Engine`` == one GPU's
batch of E environments.  Host numpy in / numpy out for the gymnasium-style API, raw device
pointers for on-GPU policies, RCCL gathers and the benchmark.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi, _lib


class EngineError(RuntimeError):
    pass


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Engine:
    def __init__(self, cfg: _abi.HwyConfig, device: int = 0, stream: int | None = None):
        self._lib = _lib.load()
        self.cfg = cfg
        self.E, self.N, self.A = cfg.num_envs, cfg.num_vehicles, cfg.num_agents
        self.V, self.F = cfg.obs_vehicles, cfg.obs_features
        h = C.c_void_p()
        rc = self._lib.hwy_create(C.byref(cfg), int(device), C.c_void_p(stream or 0), C.byref(h))
        if rc != 0:
            msg = self._lib.hwy_last_error(None).decode()
            raise EngineError(f"hwy_create failed ({self._lib.hwy_status_string(rc).decode()}): {msg}")
        self._h = h
        self.device = device

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.hwy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc == _abi.HWY_ERR_ACTION:
            # the reference raises KeyError from self.actions[int(action)] (envs/common/action.py:260)
            raise KeyError(self._lib.hwy_last_error(self._h).decode())
        if rc != 0:
            raise EngineError(f"{self._lib.hwy_status_string(rc).decode()}: {self._lib.hwy_last_error(self._h).decode()}")

    # -- state --------------------------------------------------------------------------------
    def set_state(self, st: dict):
        """hwy_set_state reads E*N elements per vehicle field and E per env field through raw pointers: every array is
        coerced to its dtype and its shape is checked here, so a dict built for another batch raises instead of
        reading out of bounds."""
        E, N = self.E, self.N
        ix = self.cfg.scenario == _abi.SCENARIO_INTERSECTION
        want = {k: (np.float64, (E, N)) for k in _abi.STATE_F64}
        want.update({k: (np.int32, (E, N)) for k in _abi.STATE_I32})
        want["time"] = (np.float64, (E,))
        if ix:
            want["route"], want["road_steps"] = (np.int64, (E, N)), (np.int32, (E,))
        out = {}
        for k, (dt, shape) in want.items():
            if k not in st:
                raise ValueError(f"set_state: missing field {k!r}")
            a = np.ascontiguousarray(st[k], dtype=dt)
            if a.shape != shape:
                raise ValueError(f"set_state: field {k!r} has shape {a.shape}, this engine needs {shape} "
                                 f"(num_envs={E}, slots per env={N})")
            out[k] = a
        st = out
        s = _abi.state_struct(st)
        self._check(self._lib.hwy_set_state(self._h, C.byref(s)))

    def get_state(self) -> dict:
        ix = self.cfg.scenario == _abi.SCENARIO_INTERSECTION
        st = _abi.alloc_state_ix(self.E, self.N) if ix else _abi.alloc_state(self.E, self.N)
        s = _abi.state_struct(st)
        self._check(self._lib.hwy_get_state(self._h, C.byref(s)))
        return st

    # -- stepping -----------------------------------------------------------------------------
    def step(self, actions):
        E, A = self.E, self.A
        acts = np.ascontiguousarray(np.asarray(actions, np.int32).reshape(E, A))
        obs = np.empty((E, A, *_abi.obs_shape(self.cfg)), np.float32)
        reward = np.empty((E, A), np.float64)
        term = np.empty(E, np.uint8)
        trunc = np.empty(E, np.uint8)
        speed = np.empty((E, A), np.float64)
        crashed = np.empty((E, A), np.uint8)
        self._check(self._lib.hwy_step(self._h, _ptr(acts), _ptr(obs), _ptr(reward), _ptr(term), _ptr(trunc),
                                       _ptr(speed), _ptr(crashed)))
        # info_crashed: bit 0 = vehicle.crashed; the intersection scenario adds bit 1 = has_arrived(vehicle) (hwy_engine.h)
        return obs, reward, term.astype(bool), trunc.astype(bool), {"speed": speed, "crashed": (crashed & 1).astype(bool),
                                                                    "arrived": (crashed & 2).astype(bool)}

    def step_device(self, d_actions: int, d_obs: int, d_reward: int, d_terminated: int, d_truncated: int,
                    d_info_speed: int = 0, d_info_crashed: int = 0):
        """Enqueue one batched policy step on raw device pointers; does not synchronise."""
        vp = C.c_void_p
        self._check(self._lib.hwy_step_device(self._h, vp(d_actions), vp(d_obs), vp(d_reward), vp(d_terminated),
                                              vp(d_truncated), vp(d_info_speed or None), vp(d_info_crashed or None)))

    def rollout_device(self, k_steps: int, d_actions: int, d_obs: int, d_reward: int, d_terminated: int, d_truncated: int,
                       d_info_speed: int = 0, d_info_crashed: int = 0):
        """Enqueue ``k_steps`` consecutive policy steps with pre-staged actions (hwy_rollout_device: block k of every plane
        belongs to step k; ONE launch on the one-wavefront kernel); does not synchronise.  Same results as ``k_steps`` calls
        of ``step_device``, bit for bit."""
        vp = C.c_void_p
        self._check(self._lib.hwy_rollout_device(self._h, int(k_steps), vp(d_actions), vp(d_obs), vp(d_reward),
                                                 vp(d_terminated), vp(d_truncated), vp(d_info_speed or None),
                                                 vp(d_info_crashed or None)))

    def rollout(self, actions):
        """hwy_rollout (host arrays): actions [K, E, A] -> (obs [K, E, A, ...], reward [K, E, A], terminated [K, E],
        truncated [K, E], info).  Raises KeyError for an action id outside the table, like ``step``."""
        acts = np.ascontiguousarray(np.asarray(actions, np.int32).reshape(-1, self.E, self.A))
        K, E, A = acts.shape[0], self.E, self.A
        obs = np.empty((K, E, A, *_abi.obs_shape(self.cfg)), np.float32)
        reward = np.empty((K, E, A), np.float64)
        term, trunc = np.empty((K, E), np.uint8), np.empty((K, E), np.uint8)
        speed, crashed = np.empty((K, E, A), np.float64), np.empty((K, E, A), np.uint8)
        self._check(self._lib.hwy_rollout(self._h, K, _ptr(acts), _ptr(obs), _ptr(reward), _ptr(term), _ptr(trunc),
                                          _ptr(speed), _ptr(crashed)))
        return obs, reward, term.astype(bool), trunc.astype(bool), {"speed": speed, "crashed": (crashed & 1).astype(bool),
                                                                    "arrived": (crashed & 2).astype(bool)}

    def step_frames(self, actions, n_frames: int):
        acts = None if actions is None else np.ascontiguousarray(np.asarray(actions, np.int32).reshape(self.E, self.A))
        self._check(self._lib.hwy_step_frames(self._h, _ptr(acts), int(n_frames)))

    def observe(self) -> np.ndarray:
        obs = np.empty((self.E, self.A, *_abi.obs_shape(self.cfg)), np.float32)
        self._check(self._lib.hwy_observe(self._h, _ptr(obs)))
        return obs

    # -- reset --------------------------------------------------------------------------------
    def reset(self, seeds=None, mask=None, ego_spacing=2.0, vehicles_density=1.0, initial_lane_id=-1, base_seed=0):
        """Device-side spawn (counter-based RNG; NOT numpy's stream -- see spawn.py for that)."""
        if seeds is None:
            seeds = np.uint64(base_seed) + np.arange(self.E, dtype=np.uint64)
        sd = np.ascontiguousarray(seeds, np.uint64)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        obs = np.zeros((self.E, self.A, *_abi.obs_shape(self.cfg)), np.float32)
        self._check(self._lib.hwy_reset(self._h, _ptr(mk), _ptr(sd), float(ego_spacing), float(vehicles_density),
                                        int(initial_lane_id), _ptr(obs)))
        return obs

    def set_autoreset(self, enabled: bool, base_seed: int = 0, ego_spacing=2.0, vehicles_density=1.0,
                      initial_lane_id=-1):
        self._check(self._lib.hwy_set_autoreset(self._h, int(bool(enabled)), C.c_uint64(base_seed), float(ego_spacing),
                                                float(vehicles_density), int(initial_lane_id)))

    def debug_math(self, op: int, x) -> np.ndarray:
        """Evaluate one of the kernel's math routines (csrc/hwy_math.h) on the device (self-test hook)."""
        xin = np.ascontiguousarray(x, np.float64).ravel()
        out = np.empty_like(xin)
        self._check(self._lib.hwy_debug_math(self._h, int(op), _ptr(xin), _ptr(out), xin.size))
        return out.reshape(np.shape(x))

    # -- misc ---------------------------------------------------------------------------------
    COUNTERS = ("ix_spawns", "ix_spawns_dropped", "nonfinite_stores")

    def counters(self, reset: bool = False) -> dict:
        """Event counters (hwy_get_counters): intersection scenario, device traffic -- spawns performed, and spawns
        DROPPED because all ``max_vehicles`` slots were taken (the reference's vehicle list is unbounded)."""
        out = (C.c_uint64 * 8)()
        self._check(self._lib.hwy_get_counters(self._h, out, 8, int(bool(reset))))
        return {k: int(out[i]) for i, k in enumerate(self.COUNTERS)}

    def set_block_order(self, env_of_block=None):
        """hwy_set_block_order: workgroup b of the following step launches steps environment ``env_of_block[b]`` (a permutation;
        None = identity).  A placement of the environments on the SIMDs -- never changes a result."""
        if env_of_block is None:
            self._check(self._lib.hwy_set_block_order(self._h, None))
            return
        a = np.ascontiguousarray(env_of_block, np.int32)
        if a.shape != (self.E,):
            raise ValueError(f"env_of_block must have shape ({self.E},)")
        self._check(self._lib.hwy_set_block_order(self._h, _ptr(a)))

    # -- multi-GPU: RCCL gather behind the ABI (one process per GPU) -----------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: the 128-byte RCCL id every rank passes to ``comm_init`` (ship it by any means)."""
        lib = _lib.load()
        buf = (C.c_uint8 * 128)()
        rc = lib.hwy_comm_unique_id(buf)
        if rc != 0:
            raise EngineError(f"hwy_comm_unique_id: {lib.hwy_last_error(None).decode()}")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.hwy_comm_init(self._h, buf, int(rank), int(world)))

    def gather(self, d_send: int, d_recv: int, nbytes: int, root: int = 0):
        """Enqueue ncclGather of ``nbytes`` bytes per rank (device pointers) on the engine's stream."""
        self._check(self._lib.hwy_gather(self._h, C.c_void_p(d_send), C.c_void_p(d_recv or None), C.c_size_t(nbytes), int(root)))

    def comm_destroy(self):
        self._check(self._lib.hwy_comm_destroy(self._h))

    def sync(self):
        self._check(self._lib.hwy_sync(self._h))

    def profile_enable(self, every: int = 1):
        """HIP-event timing of every `every`-th step-kernel launch (0 / False = off)."""
        self._check(self._lib.hwy_profile_enable(self._h, int(every)))

    def profile_read(self):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.hwy_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def prio_turn(self):
        """(turn, state) of hwy_get_prio_turn: the issue-priority turn in use (tune_prio_shift's encoding) and whether the engine
        chose it itself (0 = no selection, 1 = still sampling its first launches, 2 = chosen)."""
        turn, state = C.c_int32(), C.c_int32()
        self._check(self._lib.hwy_get_prio_turn(self._h, C.byref(turn), C.byref(state)))
        return turn.value, state.value


def device_count() -> int:
    return _lib.load().hwy_device_count()
