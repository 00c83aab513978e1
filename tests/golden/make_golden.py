#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports the reference package through ``oracle/ref_stub.py`` (gymnasium and
pygame are not installed here), drives ``HighwayEnvFast`` / ``HighwayEnv``
(``highway_env/envs/highway_env.py``) exactly like ``AbstractEnv.step`` does
(``envs/common/abstract.py:259-317``) and records

* the full vehicle state after ``reset(seed=s)``                       -> ``init_*``
* the full vehicle state after EVERY simulation frame (``Road.step``)   -> ``frame_*``
* ``obs, reward, terminated, truncated, info["speed"|"crashed"]`` per step.

The fixtures travel to the GPU box (the reference does not); tests compare the
C oracle and the HIP engine against them.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_stub  # noqa: E402

ref_stub.install()

from highway_env.envs.highway_env import HighwayEnv, HighwayEnvFast  # noqa: E402
from highway_env.vehicle.behavior import IDMVehicle  # noqa: E402
from highway_env.vehicle.controller import MDPVehicle  # noqa: E402

F64_FIELDS = ["x", "y", "heading", "speed", "timer", "target_speed", "delta",
              "impact_x", "impact_y", "act_steering", "act_accel"]
I8_FIELDS = ["lane", "target_lane", "speed_index", "crashed", "has_impact",
             "check_collisions", "controlled"]


def dump_state(env) -> dict:
    vs = env.road.vehicles
    n = len(vs)
    out = {k: np.zeros(n, np.float64) for k in F64_FIELDS}
    out.update({k: np.zeros(n, np.int8) for k in I8_FIELDS})
    for i, v in enumerate(vs):
        out["x"][i], out["y"][i] = v.position
        out["heading"][i] = v.heading
        out["speed"][i] = v.speed
        out["timer"][i] = getattr(v, "timer", np.nan)
        out["target_speed"][i] = v.target_speed
        out["delta"][i] = v.DELTA if isinstance(v, IDMVehicle) else np.nan
        if v.impact is not None:
            out["impact_x"][i], out["impact_y"][i] = v.impact
            out["has_impact"][i] = 1
        out["act_steering"][i] = v.action["steering"]
        out["act_accel"][i] = v.action["acceleration"]
        out["lane"][i] = v.lane_index[2]
        out["target_lane"][i] = v.target_lane_index[2]
        out["speed_index"][i] = v.speed_index if isinstance(v, MDPVehicle) else -1
        out["crashed"][i] = v.crashed
        out["check_collisions"][i] = v.check_collisions
        out["controlled"][i] = v in env.controlled_vehicles
    return out


SCENARIOS = [
    # BASELINE config 1: highway-fast-v0 defaults (N=21, 3 lanes, 5 Hz, ego-only collisions)
    dict(name="cfg1_fast_default", cls=HighwayEnvFast, config={}, seeds=list(range(6)),
         steps=20, action_seed=1234, frames_for=3),
    # per-env workload of BASELINE config 2: N=51, 4 lanes
    dict(name="cfg2_fast_n50_l4", cls=HighwayEnvFast,
         config={"vehicles_count": 50, "lanes_count": 4}, seeds=list(range(4)),
         steps=12, action_seed=4321, frames_for=2),
    # highway-v0 defaults: N=51, 15 Hz, full pairwise collisions
    dict(name="v0_default", cls=HighwayEnv, config={}, seeds=[0, 1], steps=6,
         action_seed=99, frames_for=1),
    # per-env workload of BASELINE config 3: N=101
    dict(name="cfg3_v0_n100", cls=HighwayEnv, config={"vehicles_count": 100}, seeds=[0, 1, 2],
         steps=3, action_seed=7, frames_for=3),
    # crash-rich: dense traffic, full collisions, ego keeps accelerating / weaving
    dict(name="dense_crash", cls=HighwayEnv,
         config={"vehicles_count": 30, "vehicles_density": 2.5, "lanes_count": 3,
                 "ego_spacing": 1.0, "duration": 20},
         seeds=[3, 5, 11], steps=14, action_seed=5, frames_for=3,
         action_p=[0.25, 0.05, 0.25, 0.4, 0.05]),
    # many first crashes (what a user sees WITH terminated=True): ego-only collisions, 5 Hz ...
    dict(name="crash_many_fast", cls=HighwayEnvFast,
         config={"vehicles_count": 30, "vehicles_density": 2.0, "lanes_count": 3, "ego_spacing": 1.0, "duration": 20},
         seeds=list(range(100, 140)), steps=10, action_seed=21, frames_for=0,
         action_p=[0.25, 0.05, 0.25, 0.4, 0.05]),
    # ... and full pairwise collisions at 15 Hz, with per-frame states for the first envs
    dict(name="crash_many_v0", cls=HighwayEnv,
         config={"vehicles_count": 30, "vehicles_density": 2.5, "lanes_count": 3, "ego_spacing": 1.0, "duration": 20},
         seeds=list(range(200, 224)), steps=8, action_seed=22, frames_for=6,
         action_p=[0.25, 0.05, 0.25, 0.4, 0.05]),
    # DiscreteMetaAction with one of the two action tables only (action.py:204-253): ids index ACTIONS_LAT / ACTIONS_LONGI
    dict(name="fast_lateral_only", cls=HighwayEnvFast,
         config={"vehicles_count": 25, "lanes_count": 4, "duration": 20,
                 "action": {"type": "DiscreteMetaAction", "longitudinal": False}},
         seeds=[31, 32, 33], steps=12, action_seed=31, frames_for=2, n_actions=3),
    dict(name="v0_longitudinal_only", cls=HighwayEnv,
         config={"vehicles_count": 20, "lanes_count": 3, "duration": 20, "simulation_frequency": 5,
                 "action": {"type": "DiscreteMetaAction", "lateral": False, "target_speeds": [15, 20, 25, 30, 35]}},
         seeds=[41, 42, 43], steps=12, action_seed=41, frames_for=2, n_actions=3),
    # KinematicObservation(order="shuffled"): close_objects_to(sort=False) + np_random.shuffle(obs[1:]) (observation.py:245,273)
    dict(name="fast_shuffled", cls=HighwayEnvFast,
         config={"vehicles_count": 25, "lanes_count": 4, "duration": 20,
                 "observation": {"type": "Kinematics", "order": "shuffled", "vehicles_count": 7}},
         seeds=[51, 52, 53], steps=10, action_seed=51, frames_for=0),
    # all-IDLE free run (no agent interference): long horizon, many MOBIL decisions
    dict(name="fast_idle_long", cls=HighwayEnvFast,
         config={"vehicles_count": 30, "lanes_count": 4, "duration": 40}, seeds=[21, 22],
         steps=40, action_seed=None, frames_for=0),
    # reward/termination variants
    dict(name="fast_offroad_terminal", cls=HighwayEnvFast,
         config={"offroad_terminal": True, "normalize_reward": False, "lanes_count": 2,
                 "vehicles_count": 10}, seeds=[0, 1, 2, 4], steps=10, action_seed=17,
         frames_for=0, action_p=[0.4, 0.1, 0.4, 0.05, 0.05]),
    # OccupancyGridObservation on the straight highway: defaults ...
    dict(name="grid_default", cls=HighwayEnvFast,
         config={"vehicles_count": 30, "lanes_count": 4, "observation": {"type": "OccupancyGrid"}},
         seeds=[0, 1, 2], steps=8, action_seed=11, frames_for=0),
    # ... vehicle-aligned axes, finer / asymmetric grid, more features, no clipping
    dict(name="grid_aligned_fine", cls=HighwayEnvFast,
         config={"vehicles_count": 40, "lanes_count": 3,
                 "observation": {"type": "OccupancyGrid", "align_to_vehicle_axes": True,
                                 "grid_size": [[-30, 60], [-9, 9]], "grid_step": [3, 2],
                                 "features": ["presence", "x", "y", "vx", "vy", "cos_h", "sin_h", "on_road"],
                                 "clip": False}},
         seeds=[3, 4], steps=8, action_seed=12, frames_for=0, action_p=[0.3, 0.1, 0.3, 0.2, 0.1]),
    # ... as_image: uint8 cells, ((clip(v, -1, 1) + 1) / 2 * 255).astype(uint8), an empty cell 0 (observation.py:408-409)
    dict(name="grid_image", cls=HighwayEnvFast,
         config={"vehicles_count": 30, "lanes_count": 4,
                 "observation": {"type": "OccupancyGrid", "as_image": True,
                                 "features": ["presence", "vx", "vy", "cos_h", "sin_h", "on_road"]}},
         seeds=[6, 7, 8], steps=8, action_seed=14, frames_for=0),
    # ... x/y in features_range (the reference normalises then de-normalises the coordinates)
    dict(name="grid_xy_range", cls=HighwayEnv,
         config={"vehicles_count": 25, "lanes_count": 4, "simulation_frequency": 5, "duration": 12,
                 "observation": {"type": "OccupancyGrid",
                                 "features_range": {"x": [-100, 100], "y": [-20, 20], "vx": [-40, 40], "vy": [-10, 10]},
                                 "features": ["presence", "vx", "vy", "x", "on_road"]}},
         seeds=[5], steps=6, action_seed=13, frames_for=0),
]


def run_scenario(sc: dict, only_envs=None) -> dict:
    """`only_envs`: simulate only these env indices (tests/test_fixture_freshness.py regenerates env 0 of every fixture); the action
    table is drawn for all of them either way, so an env's trajectory does not depend on which others are simulated."""
    ref_stub.restore_class_defaults()   # (an IntersectionEnv created earlier in this process has re-tuned the IDM CLASS)
    seeds, steps = sc["seeds"], sc["steps"]
    E = len(seeds)
    if sc["action_seed"] is None:
        actions = np.ones((steps, E), np.int32)
    else:
        rng = np.random.default_rng(sc["action_seed"])
        p = sc.get("action_p")
        n_act = sc.get("n_actions", 5)
        actions = (rng.choice(n_act, size=(steps, E), p=p) if p is not None
                   else rng.integers(0, n_act, size=(steps, E))).astype(np.int32)
    out: dict = {"seeds": np.asarray(seeds, np.int64), "actions": actions}
    per_env = []
    frames_for = sc["frames_for"]
    for e, seed in enumerate(seeds):
        if only_envs is not None and e not in only_envs:
            continue
        env = sc["cls"](dict(sc["config"]))
        obs0, _ = env.reset(seed=int(seed))
        T = int(env.config["simulation_frequency"] // env.config["policy_frequency"])
        rec = {"obs0": obs0, "init": dump_state(env), "obs": [], "reward": [], "terminated": [],
               "truncated": [], "speed": [], "crashed": [], "step_state": [], "frames": []}
        record_frames = e < frames_for
        if record_frames:
            road = env.road
            orig_step = road.step

            def step_and_dump(dt, _orig=orig_step, _env=env, _rec=rec):
                _orig(dt)
                _rec["frames"].append(dump_state(_env))

            road.step = step_and_dump
        for t in range(steps):
            o, r, te, tr, info = env.step(int(actions[t, e]))
            rec["obs"].append(o)
            rec["reward"].append(r)
            rec["terminated"].append(te)
            rec["truncated"].append(tr)
            rec["speed"].append(info["speed"])
            rec["crashed"].append(info["crashed"])
            rec["step_state"].append(dump_state(env))
        rec["T"] = T
        rec["cfg"] = dict(env.config)
        per_env.append(rec)
    E = len(per_env)
    cfg = per_env[0]["cfg"]
    N = len(per_env[0]["init"]["x"])
    out["meta"] = np.asarray([E, N, per_env[0]["T"], steps, frames_for], np.int64)
    out["cfg_lanes_count"] = np.int64(cfg["lanes_count"])
    out["cfg_vehicles_count"] = np.int64(cfg["vehicles_count"])
    out["cfg_simulation_frequency"] = np.int64(cfg["simulation_frequency"])
    out["cfg_policy_frequency"] = np.int64(cfg["policy_frequency"])
    out["cfg_duration"] = np.float64(cfg["duration"])
    out["cfg_ego_spacing"] = np.float64(cfg["ego_spacing"])
    out["cfg_vehicles_density"] = np.float64(cfg["vehicles_density"])
    out["cfg_fast"] = np.int64(sc["cls"] is HighwayEnvFast)
    out["cfg_normalize_reward"] = np.int64(cfg["normalize_reward"])
    out["cfg_offroad_terminal"] = np.int64(cfg["offroad_terminal"])
    out["cfg_collision_reward"] = np.float64(cfg["collision_reward"])
    out["cfg_right_lane_reward"] = np.float64(cfg["right_lane_reward"])
    out["cfg_high_speed_reward"] = np.float64(cfg["high_speed_reward"])
    out["cfg_reward_speed_range"] = np.asarray(cfg["reward_speed_range"], np.float64)
    import json
    out["cfg_observation_json"] = np.asarray(json.dumps(cfg["observation"]))
    out["cfg_action_json"] = np.asarray(json.dumps(cfg["action"]))
    out["obs0"] = np.stack([r["obs0"] for r in per_env])
    out["obs"] = np.stack([np.stack(r["obs"]) for r in per_env], axis=1)          # [steps,E,V,F]
    out["reward"] = np.asarray([r["reward"] for r in per_env], np.float64).T      # [steps,E]
    out["terminated"] = np.asarray([r["terminated"] for r in per_env], np.int8).T
    out["truncated"] = np.asarray([r["truncated"] for r in per_env], np.int8).T
    out["info_speed"] = np.asarray([r["speed"] for r in per_env], np.float64).T
    out["info_crashed"] = np.asarray([r["crashed"] for r in per_env], np.int8).T
    for k in F64_FIELDS + I8_FIELDS:
        out["init_" + k] = np.stack([r["init"][k] for r in per_env])              # [E,N]
        out["step_" + k] = np.stack([np.stack([s[k] for s in r["step_state"]])
                                     for r in per_env], axis=1)                   # [steps,E,N]
        if frames_for:
            out["frame_" + k] = np.stack([np.stack([s[k] for s in r["frames"]])
                                          for r in per_env[:frames_for]], axis=1)  # [steps*T,Ef,N]
    return out


def main() -> None:
    only = set(sys.argv[1:])
    for sc in SCENARIOS:
        if only and sc["name"] not in only:
            continue
        data = run_scenario(sc)
        path = os.path.join(HERE, sc["name"] + ".npz")
        np.savez_compressed(path, **data)
        print(f"{sc['name']}: E,N,T,steps,frames_for={data['meta'].tolist()} "
              f"terminated_any={bool(data['terminated'].any())} "
              f"crashed_total={int(data['step_crashed'][-1].sum())} "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
