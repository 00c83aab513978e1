"""Host-side, stream-identical reset: ``HighwayEnv._create_vehicles``
(highway_env/envs/highway_env.py:72-98) replayed with numpy's PCG64 stream so that
``reset(seed=s)`` yields exactly the reference's initial traffic.

Two layers:

* :func:`draw_reference_stream` consumes ``np.random.default_rng(seed)`` (what
  ``gymnasium.Env.reset(seed=seed)`` installs as ``env.np_random``) in the reference's order:
  per vehicle ``choice(from)``, ``choice(to)`` (single-element lists: no state consumed),
  ``choice(lanes)`` unless ``lane_id`` is given, ``uniform(0.7*limit, 0.8*limit)`` unless ``speed``
  is given, ``uniform(0.9, 1.1)`` (``Vehicle.create_random``, vehicle/kinematics.py:50-104), then
  ``uniform(3.5, 4.5)`` for ``IDMVehicle.randomize_behavior`` (vehicle/behavior.py:66-69).
* :func:`spawn_from_draws` is the spawn RULE as array arithmetic (lane ids + raw uniforms in,
  struct-of-arrays out).  The device-side reset kernel (hwy_device.h: spawn_env) implements the
  same rule on Philox uniforms, and is tested against this function.
"""
from __future__ import annotations

import numpy as np

from . import _abi


def controlled_mask(cfg: _abi.HwyConfig) -> np.ndarray:
    m = np.zeros(cfg.num_vehicles, bool)
    for a in range(cfg.num_agents):
        m[cfg.agent_index[a]] = True
    return m


def spawn_from_draws(cfg: _abi.HwyConfig, lane_ids, speed_u, pos_u, delta_u, ego_spacing: float,
                     vehicles_density: float) -> dict:
    """Initial SoA for E envs from per-vehicle draws, all arrays [E, N].

    ``lane_ids`` int lane per vehicle; ``speed_u``/``pos_u``/``delta_u`` raw uniforms in [0,1)
    (``speed_u``/``delta_u`` ignored for controlled vehicles).
    """
    lane_ids = np.asarray(lane_ids)
    E, N = lane_ids.shape
    L = cfg.lanes_count
    ctrl = controlled_mask(cfg)[None, :]
    lo, hi = 0.7 * cfg.speed_limit, 0.8 * cfg.speed_limit
    speed = np.where(ctrl, 25.0, lo + (hi - lo) * np.asarray(speed_u, np.float64))
    spacing = np.where(ctrl, float(ego_spacing), 1 / vehicles_density)
    default_spacing = 12 + 1.0 * speed
    offset = spacing * default_spacing * np.exp(-5 / 40 * L)
    step = offset * (0.9 + (1.1 - 0.9) * np.asarray(pos_u, np.float64))
    # x_k = max_x(existing) + step_k: sequential accumulation in creation order, first from 3*offset_0
    x = np.empty((E, N))
    acc = 3 * offset[:, 0]
    for k in range(N):
        acc = acc + step[:, k]
        x[:, k] = acc
    st = _abi.alloc_state(E, N)
    st["x"][...] = x
    st["y"][...] = lane_ids * cfg.lane_width
    st["speed"][...] = speed
    st["lane"][...] = lane_ids
    st["target_lane"][...] = lane_ids
    # MDPVehicle ladder snap (controller.py:287-293, 326-344)
    ts = np.array([cfg.target_speeds[k] for k in range(cfg.num_target_speeds)])
    xs = (speed - ts[0]) / (ts[-1] - ts[0])
    sidx = np.clip(np.round(xs * (ts.size - 1)), 0, ts.size - 1).astype(np.int32)
    st["speed_index"][...] = np.where(ctrl, sidx, 0)
    st["target_speed"][...] = np.where(ctrl, ts[sidx], speed)
    # IDMVehicle ctor (behavior.py:64): timer = (sum(position) * pi) % LANE_CHANGE_DELAY
    st["timer"][...] = np.where(ctrl, 0.0, ((st["x"] + st["y"]) * np.pi) % 1.0)
    st["delta"][...] = np.where(ctrl, 0.0, 3.5 + (4.5 - 3.5) * np.asarray(delta_u, np.float64))
    fast = bool(cfg.flags & _abi.C_EGO_ONLY_COLLISIONS)
    st["flags"][...] = np.where(ctrl, _abi.F_CONTROLLED | _abi.F_CHECK_COLLISIONS,
                                0 if fast else _abi.F_CHECK_COLLISIONS)
    return st


def draw_reference_stream(cfg: _abi.HwyConfig, seeds, initial_lane_id=None):
    """Consume numpy's PCG64 stream exactly like the reference's reset(seed=...)."""
    E, N, L = len(seeds), cfg.num_vehicles, cfg.lanes_count
    ctrl = controlled_mask(cfg)
    lane_ids = np.zeros((E, N), np.int64)
    speed_u = np.zeros((E, N))
    pos_u = np.zeros((E, N))
    delta_u = np.zeros((E, N))
    for e, seed in enumerate(seeds):
        rng = seed if isinstance(seed, np.random.Generator) else np.random.default_rng(int(seed))
        for k in range(N):
            if ctrl[k]:
                lane_ids[e, k] = initial_lane_id if initial_lane_id is not None else rng.choice(L)
                pos_u[e, k] = rng.random()
            else:
                lane_ids[e, k] = rng.choice(L)
                speed_u[e, k] = rng.random()
                pos_u[e, k] = rng.random()
                delta_u[e, k] = rng.random()
    return lane_ids, speed_u, pos_u, delta_u


def spawn_reference_stream(cfg: _abi.HwyConfig, seeds, ego_spacing: float, vehicles_density: float,
                           initial_lane_id=None) -> dict:
    draws = draw_reference_stream(cfg, seeds, initial_lane_id)
    return spawn_from_draws(cfg, *draws, ego_spacing=ego_spacing, vehicles_density=vehicles_density)
