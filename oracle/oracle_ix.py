"""TEST INFRASTRUCTURE: ctypes wrapper around the intersection oracle (oracle/hwy_oracle_ix.c, in libhwy_oracle.so).

The checker for the next hot-path row (intersection dynamics); nothing in the product
package imports it.  The structs mirror the oracle's private ``ix_config`` / ``ix_state``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import oracle as _base

IX_MAX_LANES, IX_MAX_ROUTE, IX_MAX_FEATURES = 32, 8, 16
FEATURE_IDS = {"presence": 0, "x": 1, "y": 2, "vx": 3, "vy": 4, "heading": 5, "cos_h": 6, "sin_h": 7, "cos_d": 8, "sin_d": 9, "long_off": 10, "lat_off": 11, "ang_off": 12, "on_road": 13}
LANE_F64 = ["sx", "sy", "ex", "ey", "heading", "dirx", "diry", "cx", "cy", "radius", "start_phase", "end_phase",
            "length", "width", "speed_limit"]
LANE_I32 = ["kind", "direction", "priority", "forbidden", "from_node", "to_node", "id"]
STATE_F64 = ["x", "y", "heading", "speed", "timer", "target_speed", "delta", "impact_x", "impact_y"]
STATE_I32 = ["present", "lane", "target_lane", "speed_index", "crashed", "has_impact", "controlled", "is_yielding",
             "yield_timer", "route_len"]
STATE_ROUTE = ["route_from", "route_to", "route_id"]


class IxLane(C.Structure):
    _fields_ = ([(k, C.c_int32) for k in ["kind", "direction", "priority", "forbidden", "from_node", "to_node", "id",
                                          "exit_lane"]]
                + [(k, C.c_double) for k in ["sx", "sy", "ex", "ey", "heading", "dirx", "diry", "cx", "cy", "radius",
                                             "start_phase", "end_phase", "length", "width", "speed_limit"]])


class IxConfig(C.Structure):
    _fields_ = ([(k, C.c_int32) for k in ["num_envs", "n_slots", "n_lanes", "n_route", "frames_per_step",
                                          "num_target_speeds", "obs_vehicles", "obs_features"]]
                + [("obs_feature_ids", C.c_int32 * IX_MAX_FEATURES)]
                + [(k, C.c_int32) for k in ["obs_absolute", "obs_normalize", "obs_clip", "obs_see_behind",
                                            "normalize_reward", "offroad_terminal", "connected_lanes", "obs_unsorted"]]
                + [(k, C.c_double) for k in ["dt", "policy_dt", "duration", "perception_distance", "distance_wanted",
                                             "time_wanted", "comfort_acc_max", "comfort_acc_min"]]
                + [("target_speeds", C.c_double * 8)]
                + [(k, C.c_double) for k in ["collision_reward", "high_speed_reward", "arrived_reward"]]
                + [("reward_speed_range", C.c_double * 2)]
                + [(k, C.c_double * 2) for k in ["obs_range_x", "obs_range_y", "obs_range_vx", "obs_range_vy"]]
                + [("spawn_probability", C.c_double), ("access_lane", C.c_int32 * 4), ("outer_node", C.c_int32 * 4),
                   ("obs_type", C.c_int32), ("grid_align", C.c_int32), ("grid_shape", C.c_int32 * 2),
                   ("grid_min", C.c_double * 2), ("grid_step", C.c_double * 2),
                   ("num_agents", C.c_int32), ("obs_intentions", C.c_int32), ("grid_image", C.c_int32), ("pad_", C.c_int32),
                   ("lanes", IxLane * IX_MAX_LANES)])


_DP, _IP = C.POINTER(C.c_double), C.POINTER(C.c_int32)


class IxState(C.Structure):
    _fields_ = ([(k, _DP) for k in STATE_F64] + [(k, _IP) for k in STATE_I32] + [(k, _IP) for k in STATE_ROUTE]
                + [("road_steps", _IP), ("time", _DP)])


def make_config(config: dict, lane_tab: dict, node_names, num_envs: int, n_slots: int, n_route: int = 4) -> IxConfig:
    """Flatten IntersectionEnv's config dict (intersection_env.py:17-58) + the recorded lane table."""
    c = IxConfig()
    c.num_envs, c.n_slots, c.n_route = int(num_envs), int(n_slots), int(n_route)
    c.num_agents = int(config.get("controlled_vehicles", 1))
    # MultiAgentAction / MultiAgentObservation wrap the per-agent types (action.py:331-350, observation.py:715-731)
    action_cfg = config["action"].get("action_config", config["action"])
    obs_cfg = config["observation"].get("observation_config", config["observation"])
    n = len(lane_tab["kind"])
    c.n_lanes = n
    names = [str(s) for s in node_names]
    for k in range(n):
        for f in LANE_F64:
            setattr(c.lanes[k], f, float(lane_tab[f][k]))
        for f in LANE_I32:
            setattr(c.lanes[k], f, int(lane_tab[f][k]))
        c.lanes[k].exit_lane = int("il" in names[lane_tab["from_node"][k]] and "o" in names[lane_tab["to_node"][k]])
    for q in range(4):
        c.outer_node[q] = names.index(f"o{q}")
        c.access_lane[q] = next(k for k in range(n) if names[lane_tab["from_node"][k]] == f"o{q}"
                                and names[lane_tab["to_node"][k]] == f"ir{q}" and lane_tab["id"][k] == 0)
    c.frames_per_step = int(config["simulation_frequency"] // config["policy_frequency"])
    c.dt, c.policy_dt = 1 / config["simulation_frequency"], 1 / config["policy_frequency"]
    c.duration = float(config["duration"])
    c.perception_distance = 5.0 * 40.0  # AbstractEnv.PERCEPTION_DISTANCE (abstract.py:58)
    # IntersectionEnv._make_vehicles sets these on the vehicle class (intersection_env.py:243-247)
    c.distance_wanted, c.time_wanted, c.comfort_acc_max, c.comfort_acc_min = 7.0, 1.5, 6.0, -3.0
    ts = np.asarray(action_cfg["target_speeds"], np.float64)
    c.num_target_speeds = ts.size
    for k, v in enumerate(ts):
        c.target_speeds[k] = float(v)
    obs = obs_cfg
    grid = obs["type"] == "OccupancyGrid"
    if grid:  # OccupancyGridObservation.__init__ defaults (observation.py:286-327, 347-351)
        feats = obs.get("features") or ["presence", "vx", "vy", "on_road"]
        gs = np.array(obs.get("grid_size") or [[-27.5, 27.5], [-27.5, 27.5]], np.float64)
        step = np.array(obs.get("grid_step") or [5, 5], np.float64)
        shape = np.asarray(np.floor((gs[:, 1] - gs[:, 0]) / step), dtype=np.intp)
        c.obs_type, c.grid_align = 1, int(obs.get("align_to_vehicle_axes", False))
        c.grid_image = int(bool(obs.get("as_image", False)))
        c.grid_shape[0], c.grid_shape[1] = int(shape[0]), int(shape[1])
        c.grid_min[0], c.grid_min[1] = float(gs[0, 0]), float(gs[1, 0])
        c.grid_step[0], c.grid_step[1] = float(step[0]), float(step[1])
        c.obs_vehicles = 1
        fr = obs.get("features_range") or {"vx": [-80.0, 80.0], "vy": [-80.0, 80.0]}
    else:
        feats = obs["features"]
        c.obs_vehicles = int(obs["vehicles_count"])
        fr = obs["features_range"]
    c.obs_features = len(feats)
    for k, name in enumerate(feats):
        c.obs_feature_ids[k] = FEATURE_IDS[name]
    c.obs_absolute, c.obs_normalize = int(obs.get("absolute", False)), int(obs.get("normalize", True))
    c.obs_clip, c.obs_see_behind = int(obs.get("clip", True)), int(obs.get("see_behind", False))
    c.obs_unsorted = int(obs.get("order", "sorted") == "shuffled")
    c.obs_intentions = int(bool(obs.get("observe_intentions", False)))
    inf = float("inf")
    for name, field in (("x", c.obs_range_x), ("y", c.obs_range_y), ("vx", c.obs_range_vx), ("vy", c.obs_range_vy)):
        field[0], field[1] = (float(fr[name][0]), float(fr[name][1])) if name in fr else (-inf, inf)
    c.collision_reward, c.high_speed_reward = float(config["collision_reward"]), float(config["high_speed_reward"])
    c.arrived_reward = float(config["arrived_reward"])
    c.reward_speed_range[0], c.reward_speed_range[1] = map(float, config["reward_speed_range"])
    c.normalize_reward, c.offroad_terminal = int(config["normalize_reward"]), int(config["offroad_terminal"])
    c.connected_lanes = int(bool(config.get("neighbour_vehicles_connected_lanes", False)))
    c.spawn_probability = float(config["spawn_probability"])
    return c


def alloc_state(E: int, C_: int, R: int = 4) -> dict:
    st = {k: np.zeros((E, C_), np.float64) for k in STATE_F64}
    st.update({k: np.zeros((E, C_), np.int32) for k in STATE_I32})
    st.update({k: np.full((E, C_, R), -1, np.int32) for k in STATE_ROUTE})
    st["road_steps"] = np.zeros(E, np.int32)
    st["time"] = np.zeros(E, np.float64)
    return st


def _struct(st: dict) -> IxState:
    s = IxState()
    for k in STATE_F64 + ["time"]:
        assert st[k].dtype == np.float64 and st[k].flags.c_contiguous, k
        setattr(s, k, st[k].ctypes.data_as(_DP))
    for k in STATE_I32 + STATE_ROUTE + ["road_steps"]:
        assert st[k].dtype == np.int32 and st[k].flags.c_contiguous, k
        setattr(s, k, st[k].ctypes.data_as(_IP))
    return s


def _lib():
    lib = _base.lib()
    lib.orc_ix_config_size.restype = C.c_size_t
    assert lib.orc_ix_config_size() == C.sizeof(IxConfig), "ix_config layout mismatch"
    return lib


def frames(cfg: IxConfig, st: dict, actions, n_frames: int) -> None:
    acts = None if actions is None else np.ascontiguousarray(np.asarray(actions, np.int32).reshape(cfg.num_envs * agents(cfg)))
    s = _struct(st)
    rc = _lib().orc_ix_frames(C.byref(cfg), C.byref(s), None if acts is None else acts.ctypes.data_as(_IP),
                              C.c_int32(n_frames))
    assert rc == 0, rc


def neighbours(cfg: IxConfig, st: dict, e: int, slot: int, lane: int) -> tuple:
    """Road.neighbour_vehicles(vehicle, lane_index): (front | None, rear | None) as positions in the vehicle list."""
    f, b = C.c_int32(-1), C.c_int32(-1)
    s = _struct(st)
    rc = _lib().orc_ix_neighbours(C.byref(cfg), C.byref(s), C.c_int32(e), C.c_int32(slot), C.c_int32(lane),
                                  C.byref(f), C.byref(b))
    assert rc == 0, rc
    return (None if f.value < 0 else f.value), (None if b.value < 0 else b.value)


def obs_shape(cfg: IxConfig) -> tuple:
    return (cfg.obs_features, cfg.grid_shape[0], cfg.grid_shape[1]) if cfg.obs_type == 1 else (cfg.obs_vehicles, cfg.obs_features)


def agents(cfg: IxConfig) -> int:
    return max(1, int(cfg.num_agents))


def _agent_dims(cfg: IxConfig) -> tuple:
    return (cfg.num_envs, agents(cfg)) if agents(cfg) > 1 else (cfg.num_envs,)


def observe(cfg: IxConfig, st: dict) -> np.ndarray:
    """[E, *obs_shape], or [E, A, *obs_shape] with A > 1 controlled vehicles (MultiAgentObservation's tuple, stacked)."""
    obs = np.zeros((*_agent_dims(cfg), *obs_shape(cfg)), np.float32)
    s = _struct(st)
    rc = _lib().orc_ix_observe(C.byref(cfg), C.byref(s), obs.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0, rc
    return obs


def step(cfg: IxConfig, st: dict, actions) -> tuple:
    """AbstractEnv.step up to (not including) IntersectionEnv.step's clear / spawn.  With A > 1 controlled vehicles the obs
    is [E, A, ...], `reward` the mean of the agents' rewards as the reference sums them (intersection_env.py:62-66) and the
    info holds agents_rewards / agents_terminated [E, A] (:114-122); speed / crashed are the FIRST agent's (abstract.py:219)."""
    E, A = cfg.num_envs, agents(cfg)
    acts = np.ascontiguousarray(np.asarray(actions, np.int32).reshape(E * A))
    obs = np.zeros((*_agent_dims(cfg), *obs_shape(cfg)), np.float32)
    rewards, speed = np.zeros((E, A)), np.zeros((E, A))
    term, trunc, bits = np.zeros(E, np.uint8), np.zeros(E, np.uint8), np.zeros((E, A), np.uint8)
    s = _struct(st)
    u8 = C.POINTER(C.c_uint8)
    rc = _lib().orc_ix_step(C.byref(cfg), C.byref(s), acts.ctypes.data_as(_IP), obs.ctypes.data_as(C.POINTER(C.c_float)),
                            rewards.ctypes.data_as(_DP), term.ctypes.data_as(u8), trunc.ctypes.data_as(u8),
                            speed.ctypes.data_as(_DP), bits.ctypes.data_as(u8))
    assert rc == 0, rc
    reward = np.zeros(E)
    for a in range(A):  # Python's sum(): left to right from 0
        reward = reward + rewards[:, a]
    reward = reward / A
    crashed, arrived = (bits & 1).astype(bool), (bits & 2).astype(bool)
    info = {"speed": speed[:, 0].copy(), "crashed": crashed[:, 0].copy(), "agents_rewards": rewards,
            "agents_terminated": crashed | arrived}
    return obs, reward, term.astype(bool), trunc.astype(bool), info


def clear_spawn(cfg: IxConfig, st: dict, draws, n_draws) -> np.ndarray:
    """IntersectionEnv._clear_vehicles + _spawn_vehicle on recorded draws; returns the draws consumed per env."""
    d = np.ascontiguousarray(draws, np.float64)
    nd = np.ascontiguousarray(n_draws, np.int32)
    used = np.zeros(cfg.num_envs, np.int32)
    s = _struct(st)
    rc = _lib().orc_ix_clear_spawn(C.byref(cfg), C.byref(s), d.ctypes.data_as(_DP), nd.ctypes.data_as(_IP),
                                   C.c_int32(d.shape[-1]), used.ctypes.data_as(_IP))
    assert rc == 0, rc
    return used


# ---- adapters from the product's hwy_state / hwy_config (used by the tests, smoke() and bench.py's cpu_baseline) ----

def state_from_engine(h: dict, cfg) -> dict:
    """The product's hwy_state dict of the intersection scenario -> the oracle's (oracle/oracle_ix.py) layout."""
    from highwayenv_amd import _abi
    from highwayenv_amd import intersection as hix
    tab = hix.table_from_config(cfg)
    E, N = h["x"].shape
    st = alloc_state(E, N, 8)
    for k in STATE_F64:
        st[k][...] = h[k]
    f = h["flags"]
    st["present"][...] = (f & _abi.F_ABSENT) == 0
    for k, bit in (("crashed", _abi.F_CRASHED), ("has_impact", _abi.F_HAS_IMPACT), ("controlled", _abi.F_CONTROLLED),
                   ("is_yielding", _abi.F_YIELDING)):
        st[k][...] = (f & bit) != 0
    for k in ("lane", "target_lane", "speed_index"):
        st[k][...] = h[k]
    for e in range(E):
        for i in range(N):
            r = hix.route_unpack(int(h["route"][e, i])) if st["present"][e, i] else []
            st["route_len"][e, i] = len(r)
            for q, l in enumerate(r):
                st["route_from"][e, i, q], st["route_to"][e, i, q] = tab["from_node"][l], tab["to_node"][l]
                st["route_id"][e, i, q] = -1 if q else 0
    st["road_steps"][...] = h["road_steps"]
    st["time"][...] = h["time"]
    return st


def config_from_engine(cfg_dict: dict, cfg, num_envs: int):
    from highwayenv_amd import intersection as hix
    tab = hix.table_from_config(cfg)
    lane_tab = dict(tab)
    lane_tab["ex"] = lane_tab["ey"] = lane_tab["end_phase"] = np.zeros_like(tab["sx"])
    lane_tab["id"] = np.zeros_like(tab["kind"])
    return make_config(cfg_dict, lane_tab, hix.NODE_NAMES, num_envs, cfg.num_vehicles, 8)
