"""ctypes mirror of ``include/hwy_engine.h`` (POD types + constants) and the
derivation of the flat ``hwy_config`` from a reference-style config dict.

Reference for the dict: ``HighwayEnv.default_config`` (highway_env/envs/highway_env.py:25-53),
``HighwayEnvFast.default_config`` (:162-175), ``AbstractEnv.default_config``
(envs/common/abstract.py:101-125), ``KinematicObservation.__init__``
(envs/common/observation.py:160-197).
"""
from __future__ import annotations

import copy
import ctypes as C

import numpy as np

HWY_ABI_VERSION = 5
HWY_MAX_AGENTS = 16
HWY_MAX_FEATURES = 16
HWY_MAX_TARGET_SPEEDS = 8
HWY_MAX_LANES = 16
HWY_MAX_VEHICLES = 256
HWY_MAX_GLANES = 24
HWY_MAX_ROUTE = 11

# hwy_status
HWY_OK, HWY_ERR_INVALID_ARG, HWY_ERR_HIP, HWY_ERR_UNSUPPORTED, HWY_ERR_NO_DEVICE, HWY_ERR_ACTION = 0, -1, -2, -3, -4, -5

# per-vehicle flags
F_CRASHED, F_HAS_IMPACT, F_CHECK_COLLISIONS, F_CONTROLLED = 1, 2, 4, 8
F_OBSTACLE, F_ABSENT = 16, 32  # road-network scenarios only
F_YIELDING = 64                # intersection scenario only
SCENARIO_HIGHWAY, SCENARIO_MERGE, SCENARIO_MERGE_GENERIC, SCENARIO_INTERSECTION = 0, 1, 2, 3
# config flags
C_NORMALIZE_REWARD, C_OFFROAD_TERMINAL, C_OBS_ABSOLUTE, C_OBS_NORMALIZE, C_OBS_CLIP, C_OBS_SEE_BEHIND = 1, 2, 4, 8, 16, 32
C_EGO_ONLY_COLLISIONS = 64
C_GRID_ALIGN = 128
C_HOST_TRAFFIC = 256
C_CONNECTED_LANES = 512
C_OBS_UNSORTED = 1024
C_OBS_VEHICLES_ONLY = 2048
C_OBS_INTENTIONS = 4096
C_GRID_IMAGE = 8192
OBS_KINEMATICS, OBS_OCCUPANCY_GRID = 0, 1
HWY_MAX_GRID_CELLS = 65536

FEATURE_IDS = {name: i for i, name in enumerate(
    ["presence", "x", "y", "vx", "vy", "heading", "cos_h", "sin_h", "cos_d", "sin_d",
     "long_off", "lat_off", "ang_off", "on_road"])}

# DiscreteMetaAction.ACTIONS_ALL / ACTIONS_LONGI / ACTIONS_LAT (envs/common/action.py:204-210)
ACTIONS_ALL = {0: "LANE_LEFT", 1: "IDLE", 2: "LANE_RIGHT", 3: "FASTER", 4: "SLOWER"}
ACTIONS_LONGI = {0: "SLOWER", 1: "IDLE", 2: "FASTER"}
ACTIONS_LAT = {0: "LANE_LEFT", 1: "IDLE", 2: "LANE_RIGHT"}
ACTIONS_SET_ALL, ACTIONS_SET_LONGI, ACTIONS_SET_LAT = 0, 1, 2


def num_actions(cfg: "HwyConfig") -> int:
    return 5 if cfg.action_set == ACTIONS_SET_ALL else 3


class HwyLane(C.Structure):
    """hwy_lane: one x-aligned lane of a RoadNetwork (include/hwy_engine.h)."""
    _fields_ = [
        ("x0", C.c_double), ("y0", C.c_double), ("length", C.c_double), ("width", C.c_double),
        ("amplitude", C.c_double), ("pulsation", C.c_double), ("phase", C.c_double), ("speed_limit", C.c_double),
        ("road", C.c_int32), ("id", C.c_int32), ("road_first", C.c_int32), ("road_lanes", C.c_int32),
        ("next_first", C.c_int32), ("next_lanes", C.c_int32), ("forbidden", C.c_int32), ("connected", C.c_int32),
    ]


LANE_F64 = ["x0", "y0", "length", "width", "amplitude", "pulsation", "phase", "speed_limit"]
LANE_I32 = ["road", "id", "road_first", "road_lanes", "next_first", "next_lanes", "forbidden"]  # (+ "connected", derived)


class HwyGLane(C.Structure):
    """hwy_glane: one lane of a general RoadNetwork -- StraightLane of any direction or CircularLane."""
    _fields_ = ([(k, C.c_int32) for k in ["kind", "direction", "priority", "forbidden", "from_node", "to_node",
                                          "exit_lane", "reserved"]]
                + [(k, C.c_double) for k in ["sx", "sy", "heading", "dirx", "diry", "cx", "cy", "radius",
                                             "start_phase", "length", "width", "speed_limit"]])


GLANE_F64 = ["sx", "sy", "heading", "dirx", "diry", "cx", "cy", "radius", "start_phase", "length", "width", "speed_limit"]
GLANE_I32 = ["kind", "direction", "priority", "forbidden", "from_node", "to_node", "exit_lane"]


class HwyConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("num_envs", C.c_int32),
        ("num_vehicles", C.c_int32),
        ("num_agents", C.c_int32),
        ("agent_index", C.c_int32 * HWY_MAX_AGENTS),
        ("lanes_count", C.c_int32),
        ("frames_per_step", C.c_int32),
        ("flags", C.c_int32),
        ("obs_vehicles", C.c_int32),
        ("obs_features", C.c_int32),
        ("obs_feature_ids", C.c_int32 * HWY_MAX_FEATURES),
        ("num_target_speeds", C.c_int32),
        ("action_set", C.c_int32),
        ("target_speeds", C.c_double * HWY_MAX_TARGET_SPEEDS),
        ("dt", C.c_double),
        ("policy_dt", C.c_double),
        ("duration", C.c_double),
        ("lane_width", C.c_double),
        ("road_length", C.c_double),
        ("speed_limit", C.c_double),
        ("collision_reward", C.c_double),
        ("right_lane_reward", C.c_double),
        ("high_speed_reward", C.c_double),
        ("reward_speed_range", C.c_double * 2),
        ("perception_distance", C.c_double),
        ("obs_range_x", C.c_double * 2),
        ("obs_range_y", C.c_double * 2),
        ("obs_range_vx", C.c_double * 2),
        ("obs_range_vy", C.c_double * 2),
        ("obs_type", C.c_int32),
        ("grid_shape", C.c_int32 * 2),
        ("reserved1", C.c_int32),
        ("grid_min", C.c_double * 2),
        ("grid_step", C.c_double * 2),
        ("scenario", C.c_int32),
        ("net_lanes", C.c_int32),
        ("merge_lane", C.c_int32),
        ("reserved2", C.c_int32),
        ("merge_end_x", C.c_double),
        ("merging_speed_reward", C.c_double),
        ("lane_change_reward", C.c_double),
        ("net", HwyLane * HWY_MAX_LANES),
        # HWY_SCENARIO_INTERSECTION (ABI v4)
        ("gnet_lanes", C.c_int32),
        ("initial_vehicle_count", C.c_int32),
        ("destination", C.c_int32),
        ("reserved3", C.c_int32),
        ("access_lane", C.c_int32 * 4),
        ("exit_of", C.c_int32 * 4),
        ("spawn_probability", C.c_double),
        ("arrived_reward", C.c_double),
        ("idm_distance_wanted", C.c_double),
        ("idm_time_wanted", C.c_double),
        ("idm_comfort_acc_max", C.c_double),
        ("idm_comfort_acc_min", C.c_double),
        ("gnet", HwyGLane * HWY_MAX_GLANES),
        ("gnet_routes", (C.c_int64 * 4) * HWY_MAX_GLANES),
        # tuning (ABI v5): 0 = the engine's own choice
        ("tune_block_kernel", C.c_int32),
        ("tune_waves_per_eu", C.c_int32),
        ("tune_ix_no_helpers", C.c_int32),
        ("tune_ix_no_prewarm", C.c_int32),
        ("tune_extra_lds", C.c_int32),
        ("tune_prio_shift", C.c_int32),
        ("tune_ix_prewarm_frames", C.c_int32),
        ("tune_reserved", C.c_int32 * 1),
    ]


def obs_shape(cfg: "HwyConfig") -> tuple:
    """Per-agent observation shape: (V, F) Kinematics, (F, W, H) OccupancyGrid."""
    if cfg.obs_type == OBS_OCCUPANCY_GRID:
        return (cfg.obs_features, cfg.grid_shape[0], cfg.grid_shape[1])
    return (cfg.obs_vehicles, cfg.obs_features)


_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)

STATE_F64 = ["x", "y", "heading", "speed", "timer", "target_speed", "delta", "impact_x", "impact_y"]
STATE_I32 = ["lane", "target_lane", "speed_index", "flags"]


class HwyState(C.Structure):
    _fields_ = ([(n, _DP) for n in STATE_F64] + [(n, _IP) for n in STATE_I32] + [("time", _DP)]
                + [("route", C.POINTER(C.c_int64)), ("road_steps", _IP)])  # intersection scenario only (NULL otherwise)


def alloc_state(num_envs: int, num_vehicles: int) -> dict:
    """Host SoA (numpy) for hwy_set_state / hwy_get_state."""
    st = {k: np.zeros((num_envs, num_vehicles), np.float64) for k in STATE_F64}
    st.update({k: np.zeros((num_envs, num_vehicles), np.int32) for k in STATE_I32})
    st["time"] = np.zeros(num_envs, np.float64)
    return st


def alloc_state_ix(num_envs: int, num_vehicles: int) -> dict:
    """Host SoA of the intersection scenario: + packed planned routes and RegulatedRoad.steps; every slot absent."""
    st = alloc_state(num_envs, num_vehicles)
    st["flags"][...] = F_ABSENT
    st["route"] = np.zeros((num_envs, num_vehicles), np.int64)
    st["road_steps"] = np.zeros(num_envs, np.int32)
    return st


def state_struct(st: dict) -> HwyState:
    """ctypes view of a host SoA dict (arrays must stay alive while the struct is used)."""
    s = HwyState()
    for k in STATE_F64 + ["time"]:
        a = st[k]
        assert a.dtype == np.float64 and a.flags.c_contiguous, k
        setattr(s, k, a.ctypes.data_as(_DP))
    for k in STATE_I32:
        a = st[k]
        assert a.dtype == np.int32 and a.flags.c_contiguous, k
        setattr(s, k, a.ctypes.data_as(_IP))
    if "route" in st:  # intersection scenario
        a = st["route"]
        assert a.dtype == np.int64 and a.flags.c_contiguous, "route"
        s.route = a.ctypes.data_as(C.POINTER(C.c_int64))
    if "road_steps" in st:
        a = st["road_steps"]
        assert a.dtype == np.int32 and a.flags.c_contiguous, "road_steps"
        s.road_steps = a.ctypes.data_as(_IP)
    return s


def copy_state(st: dict) -> dict:
    return {k: v.copy() for k, v in st.items()}


# --------------------------------------------------------------------------- config dicts

def abstract_default_config() -> dict:
    """AbstractEnv.default_config (envs/common/abstract.py:101-125), minus rendering keys' effect."""
    return {
        "observation": {"type": "Kinematics"},
        "action": {"type": "DiscreteMetaAction"},
        "simulation_frequency": 15,
        "policy_frequency": 1,
        "other_vehicles_type": "highway_env.vehicle.behavior.IDMVehicle",
        "screen_width": 600,
        "screen_height": 150,
        "centering_position": [0.3, 0.5],
        "scaling": 5.5,
        "show_trajectories": False,
        "render_agent": True,
        "offscreen_rendering": None,
        "manual_control": False,
        "real_time_rendering": False,
        "neighbour_vehicles_connected_lanes": False,
    }


def highway_default_config() -> dict:
    """HighwayEnv.default_config (envs/highway_env.py:25-53)."""
    cfg = abstract_default_config()
    cfg.update({
        "observation": {"type": "Kinematics"},
        "action": {"type": "DiscreteMetaAction"},
        "lanes_count": 4,
        "vehicles_count": 50,
        "controlled_vehicles": 1,
        "initial_lane_id": None,
        "duration": 40,
        "ego_spacing": 2,
        "vehicles_density": 1,
        "collision_reward": -1,
        "right_lane_reward": 0.1,
        "high_speed_reward": 0.4,
        "lane_change_reward": 0,
        "reward_speed_range": [20, 30],
        "normalize_reward": True,
        "offroad_terminal": False,
    })
    return cfg


def highway_fast_default_config() -> dict:
    """HighwayEnvFast.default_config (envs/highway_env.py:162-175)."""
    cfg = highway_default_config()
    cfg.update({"simulation_frequency": 5, "lanes_count": 3, "vehicles_count": 20,
                "duration": 30, "ego_spacing": 1.5})
    return cfg


def near_split(x: int, num_bins: int) -> list:
    """utils.near_split (highway_env/utils.py:355-370), num_bins form."""
    quotient, remainder = divmod(x, num_bins)
    return [quotient + 1] * remainder + [quotient] * (num_bins - remainder)


def agent_indices(vehicles_count: int, controlled: int) -> list:
    """Positions of the controlled vehicles in Road.vehicles as built by
    HighwayEnv._create_vehicles (highway_env.py:72-98): ego_k followed by its share of others."""
    idx, pos = [], 0
    for others in near_split(vehicles_count, controlled):
        idx.append(pos)
        pos += 1 + others
    return idx


TUNING_KEYS = ("block_kernel", "waves_per_eu", "ix_no_helpers", "ix_no_prewarm", "extra_lds", "prio_shift", "ix_prewarm_frames")


def make_config(config: dict, num_envs: int, fast: bool = False, scenario: str = "highway",
                tuning: dict | None = None) -> HwyConfig:
    """Flatten a reference-style config dict into the POD the engine takes.

    ``tuning``: optional ``hwy_config.tune_*`` knobs (keys of ``TUNING_KEYS``; also read from ``config["tuning"]``);
    they select kernel variants and never change a result.

    ``scenario``: "highway" (HighwayEnv / HighwayEnvFast), "merge" (MergeEnv) or "merge-generic"
    (MergeGenericEnv) -- the latter two are filled in by ``highwayenv_amd.merge``.

    Raises the reference's errors for what it rejects (``ValueError("Unknown
    action type")`` action.py:346, ``ValueError("Unknown observation type")``
    observation.py:794) and ``NotImplementedError`` for reference features outside
    the hot-path scope.
    """
    cfg = copy.deepcopy(config)
    act = cfg["action"]
    if act["type"] == "MultiAgentAction":
        act = act["action_config"]
    if act["type"] not in ("DiscreteMetaAction", "ContinuousAction", "DiscreteAction", "MultiAgentAction"):
        raise ValueError("Unknown action type")
    if act["type"] != "DiscreteMetaAction":
        raise NotImplementedError(f"action type {act['type']} is outside the MI355X hot-path scope")
    ix = scenario == "intersection"
    longi, lat = bool(act.get("longitudinal", True)), bool(act.get("lateral", True))
    if not (longi or lat):
        raise ValueError("At least longitudinal or lateral actions must be included")  # action.py:246-249
    if ix and (not longi or lat):
        raise NotImplementedError("the intersection scenario takes longitudinal-only meta-actions")
    action_set = ACTIONS_SET_ALL if (longi and lat) else (ACTIONS_SET_LONGI if longi else ACTIONS_SET_LAT)
    obs = cfg["observation"]
    if obs["type"] == "MultiAgentObservation":
        obs = obs["observation_config"]
    known = ("Kinematics", "OccupancyGrid", "GrayscaleObservation", "TimeToCollision", "KinematicsGoal",
             "AttributesObservation", "MultiAgentObservation", "TupleObservation", "LidarObservation",
             "ExitObservation")
    if obs["type"] not in known:
        raise ValueError("Unknown observation type")
    if obs["type"] not in ("Kinematics", "OccupancyGrid"):
        raise NotImplementedError(f"observation type {obs['type']} is outside the MI355X hot-path scope")
    grid = obs["type"] == "OccupancyGrid"
    if cfg.get("other_vehicles_type", "highway_env.vehicle.behavior.IDMVehicle") != "highway_env.vehicle.behavior.IDMVehicle":
        raise NotImplementedError("only IDMVehicle traffic is in the hot-path scope")
    merge = scenario in ("merge", "merge-generic")
    if scenario != "highway" and not merge and not ix:
        raise ValueError(f"unknown scenario {scenario!r}")
    # neighbour_vehicles_connected_lanes on the single road 0->1 of highway-v0 adds no lane to the search list
    # (road.py:513-529: nothing leaves "1", nothing arrives at "0"), so the flag is accepted there and changes nothing
    if not grid and obs.get("order", "sorted") not in ("sorted", "shuffled"):
        raise ValueError("KinematicObservation order must be 'sorted' or 'shuffled'")

    c = HwyConfig()
    c.abi_version = HWY_ABI_VERSION
    c.num_envs = int(num_envs)
    for key, val in {**(cfg.get("tuning") or {}), **(tuning or {})}.items():
        if key not in TUNING_KEYS:
            raise KeyError(f"unknown tuning knob {key!r} (known: {TUNING_KEYS})")
        setattr(c, "tune_" + key, int(val))
    A = int(cfg.get("controlled_vehicles", 1))
    c.num_agents = A
    if not (1 <= A <= HWY_MAX_AGENTS):
        raise ValueError(f"controlled_vehicles must be in [1, {HWY_MAX_AGENTS}]")
    c.frames_per_step = int(cfg["simulation_frequency"] // cfg["policy_frequency"])
    c.dt = 1 / cfg["simulation_frequency"]
    c.policy_dt = 1 / cfg["policy_frequency"]
    c.lane_width = 4.0        # AbstractLane.DEFAULT_WIDTH (road/lane.py:16)
    if ix:
        from . import intersection as _ix
        _ix.fill_config(c, cfg)
        y_lanes = 1
    elif merge:
        from . import merge as _merge
        _merge.fill_config(c, cfg, generic=(scenario == "merge-generic"))
        y_lanes = _merge.ego_road_lanes(cfg, generic=(scenario == "merge-generic"))
    else:
        c.num_vehicles = int(cfg["vehicles_count"]) + A
        if not (A <= c.num_vehicles <= HWY_MAX_VEHICLES):
            raise ValueError(f"vehicles_count + controlled_vehicles must be <= {HWY_MAX_VEHICLES}")
        for k, i in enumerate(agent_indices(int(cfg["vehicles_count"]), A)):
            c.agent_index[k] = i
        c.lanes_count = int(cfg["lanes_count"])
        if not (1 <= c.lanes_count <= HWY_MAX_LANES):
            raise ValueError(f"lanes_count must be in [1, {HWY_MAX_LANES}]")
        c.duration = float(cfg["duration"])
        c.road_length = 10000.0   # RoadNetwork.straight_road_network length (road/road.py:296)
        c.speed_limit = 30.0      # HighwayEnv._create_road (highway_env.py:63)
        y_lanes = c.lanes_count
    ts = act.get("target_speeds")
    ts = np.linspace(20, 30, 3) if ts is None else np.asarray(ts, np.float64)  # controller.py:259
    if not (2 <= ts.size <= HWY_MAX_TARGET_SPEEDS):
        raise ValueError("target_speeds must hold 2..8 values")
    c.action_set = action_set
    c.num_target_speeds = int(ts.size)
    for k, v in enumerate(ts):
        c.target_speeds[k] = float(v)
    c.collision_reward = float(cfg["collision_reward"])
    c.right_lane_reward = float(cfg.get("right_lane_reward", 0.0))
    c.high_speed_reward = float(cfg["high_speed_reward"])
    c.reward_speed_range[0], c.reward_speed_range[1] = map(float, cfg["reward_speed_range"])
    c.perception_distance = 5.0 * 40.0  # AbstractEnv.PERCEPTION_DISTANCE (abstract.py:58)

    if grid:
        # OccupancyGridObservation.__init__ (observation.py:286-327)
        feats = obs["features"] if obs.get("features") is not None else ["presence", "vx", "vy", "on_road"]
        if obs.get("absolute", False):
            raise NotImplementedError()  # the reference raises it too (observation.py:358-359)
        gs = np.array(obs["grid_size"] if obs.get("grid_size") is not None else [[-27.5, 27.5], [-27.5, 27.5]], np.float64)
        step = np.array(obs["grid_step"] if obs.get("grid_step") is not None else [5, 5], np.float64)
        shape = np.asarray(np.floor((gs[:, 1] - gs[:, 0]) / step), dtype=np.intp)
        if shape.min() < 1 or int(shape[0]) * int(shape[1]) > HWY_MAX_GRID_CELLS:
            raise ValueError(f"occupancy grid must hold 1..{HWY_MAX_GRID_CELLS} cells")
        c.obs_type = OBS_OCCUPANCY_GRID
        c.grid_shape[0], c.grid_shape[1] = int(shape[0]), int(shape[1])
        c.grid_min[0], c.grid_min[1] = float(gs[0, 0]), float(gs[1, 0])
        c.grid_step[0], c.grid_step[1] = float(step[0]), float(step[1])
        c.obs_vehicles = 1
        default_range = {"vx": [-80.0, 80.0], "vy": [-80.0, 80.0]}  # OccupancyGridObservation.normalize (:347-351)
    else:
        feats = obs.get("features") or ["presence", "x", "y", "vx", "vy"]
        c.obs_type = OBS_KINEMATICS
        c.obs_vehicles = int(obs.get("vehicles_count", 5))
        # KinematicObservation.normalize_obs (observation.py:214-226): len(all_side_lanes) == lanes_count
        # (the range is fixed by the observer's road at the FIRST observation of the episode, :211-226)
        default_range = {"x": [-200.0, 200.0], "y": [-4.0 * y_lanes, 4.0 * y_lanes],
                         "vx": [-80.0, 80.0], "vy": [-80.0, 80.0]}
    if len(feats) > HWY_MAX_FEATURES:
        raise ValueError("too many observation features")
    c.obs_features = len(feats)
    for k, name in enumerate(feats):
        if name not in FEATURE_IDS or (name == "on_road" and not grid):
            raise KeyError(name)  # df[self.features] raises KeyError in the reference
        c.obs_feature_ids[k] = FEATURE_IDS[name]
    fr = obs.get("features_range") or default_range
    inf = float("inf")
    for name, field in (("x", c.obs_range_x), ("y", c.obs_range_y), ("vx", c.obs_range_vx), ("vy", c.obs_range_vy)):
        if name in fr:
            field[0], field[1] = float(fr[name][0]), float(fr[name][1])
        else:  # feature not normalised: +-inf sentinel
            field[0], field[1] = -inf, inf
    for name in fr:
        if name not in ("x", "y", "vx", "vy"):
            raise NotImplementedError(f"features_range for {name!r} is out of scope")
    flags = 0
    if cfg.get("normalize_reward", False) and not merge:  # (IntersectionEnv: lmap to [collision, arrived] rewards)
        flags |= C_NORMALIZE_REWARD
    if cfg.get("offroad_terminal", False) and not merge:
        flags |= C_OFFROAD_TERMINAL
    if obs.get("absolute", False):
        flags |= C_OBS_ABSOLUTE
    if obs.get("normalize", True):
        flags |= C_OBS_NORMALIZE
    if obs.get("clip", True):
        flags |= C_OBS_CLIP
    if obs.get("see_behind", False):
        flags |= C_OBS_SEE_BEHIND
    if (merge or ix) and cfg.get("neighbour_vehicles_connected_lanes", False):
        flags |= C_CONNECTED_LANES
    if not grid and obs.get("order", "sorted") == "shuffled":
        flags |= C_OBS_UNSORTED       # close_objects_to(sort=False); envs.py shuffles the rows on env.np_random
    if not grid and not obs.get("include_obstacles", True):
        flags |= C_OBS_VEHICLES_ONLY  # (only the merge scenarios have objects)
    if not grid and obs.get("observe_intentions", False):
        flags |= C_OBS_INTENTIONS     # (cos_d / sin_d of the other vehicles; zeros anyway where nothing has a route)
    if ix:
        for name in feats:
            if name not in ("presence", "x", "y", "vx", "vy", "heading", "cos_h", "sin_h") + (("on_road",) if grid else ("cos_d", "sin_d", "long_off", "lat_off", "ang_off")):
                raise NotImplementedError(f"feature {name!r} is out of scope for the intersection scenario's "
                                          f"{'OccupancyGrid' if grid else 'Kinematics'} observation")
        if cfg.get("host_traffic", False):
            flags |= C_HOST_TRAFFIC
    if merge:
        # the Obstacle's to_dict (vehicle/objects.py:141-160) has no heading / lane-offset keys: the reference
        # would put NaN in those columns whenever the obstacle is observed
        for name in feats:
            if name not in ("presence", "x", "y", "vx", "vy", "cos_h", "sin_h", "cos_d", "sin_d") + (("on_road",) if grid else ()):
                raise NotImplementedError(f"feature {name!r} is undefined for the Obstacle of the merge scenarios")
    if grid and obs.get("align_to_vehicle_axes", False):
        flags |= C_GRID_ALIGN
    if grid and obs.get("as_image", False):
        flags |= C_GRID_IMAGE         # uint8 cells (the engine writes integer-valued f32; envs.py casts)
    if fast:  # HighwayEnvFast._create_vehicles (highway_env.py:177-182)
        flags |= C_EGO_ONLY_COLLISIONS
    c.flags = flags
    return c
