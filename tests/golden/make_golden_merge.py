#!/usr/bin/env python3
"""Generate the merge-scenario golden fixtures from the UNMODIFIED reference.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_merge.py

Drives ``MergeEnv`` / ``MergeGenericEnv`` (``highway_env/envs/merge_env.py``) through the
``oracle/ref_stub.py`` import shim exactly like ``AbstractEnv.step`` does and records, per environment,

* the lane table of the reference's ``RoadNetwork`` in ``get_closest_lane_index`` iteration order
  (``road/road.py:55-71``) -- the tests check the product's own table builder against it;
* the full state of every vehicle AND road object after ``reset(seed=s)``, after every simulation frame
  (for the first ``frames_for`` environments) and after every policy step; lane indices are positions in
  that lane table;
* ``obs, reward, terminated, truncated, info["speed"|"crashed"]`` per step.

The multi-agent scenario (BASELINE config 5: "merge multi-agent, 4 controlled agents/env") is not
expressible with the reference's own environment classes (``MergeEnv._make_vehicles`` creates ONE
controlled vehicle), so this script defines ``MergeGenericMultiAgent``: a subclass that, after the
reference's own ``_make_vehicles``, re-creates the first A-1 traffic vehicles as controlled
``MDPVehicle``s in place.  Everything that steps, observes and rewards is still the reference's code.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_stub  # noqa: E402

ref_stub.install()

from highway_env.envs.merge_env import (ConnectedLaneMergeEnv, ConnectedLaneMergeGenericEnv,  # noqa: E402
                                         MergeEnv, MergeGenericEnv)
from highway_env.road.lane import SineLane  # noqa: E402
from highway_env.vehicle.behavior import IDMVehicle  # noqa: E402
from highway_env.vehicle.controller import MDPVehicle  # noqa: E402
from highway_env.vehicle.objects import Obstacle  # noqa: E402


class MergeGenericMultiAgent(MergeGenericEnv):
    """MergeGenericEnv with config["controlled_vehicles"] agents (see the module docstring)."""

    @classmethod
    def default_config(cls) -> dict:
        cfg = super().default_config()
        cfg.update({"controlled_vehicles": 4})
        return cfg

    def _make_vehicles(self) -> None:
        super()._make_vehicles()
        vs = self.road.vehicles
        A = self.config["controlled_vehicles"]
        agents = [vs[0]]
        for k in range(1, A):
            old = vs[k]
            new = self.action_type.vehicle_class(self.road, old.position, speed=old.speed)
            vs[k] = new
            agents.append(new)
        self.controlled_vehicles = agents


class MergeRampEgo(MergeEnv):
    """MergeEnv whose ego starts ON the acceleration lane ("b", "c", 2), which ends at the ``Obstacle`` (merge_env.py:160), and
    whose merging IDM vehicle is placed at speed a few car lengths behind it: the vehicle-vs-``Obstacle`` branch of
    ``handle_collisions`` (objects.py:215-222 via :101-113: the vehicle takes the WHOLE translation) with a terminal observation.
    Only the initial placement differs; everything that steps, observes and rewards is the reference's code."""

    def _make_vehicles(self) -> None:
        super()._make_vehicles()
        road, rng = self.road, self.np_random
        lane = road.network.get_lane(("b", "c", 2))
        ego, merging = self.vehicle, road.vehicles[-1]
        s_ego = rng.uniform(5.0, 45.0)
        fresh = self.action_type.vehicle_class(road, lane.position(s_ego, rng.uniform(-0.4, 0.4)), heading=lane.heading_at(s_ego),
                                               speed=rng.uniform(24.0, 30.0))
        road.vehicles[0] = fresh
        self.vehicle = fresh
        self.controlled_vehicles = [fresh]
        s_m = s_ego - rng.uniform(12.0, 30.0)
        src = road.network.get_lane(("k", "b", 0))
        pos = lane.position(s_m, 0.0) if s_m >= 0 else src.position(src.length + s_m, 0.0)
        m = type(merging)(road, pos, heading=(lane.heading_at(s_m) if s_m >= 0 else src.heading_at(src.length + s_m)),
                          speed=rng.uniform(22.0, 30.0))
        m.target_speed = 30.0
        road.vehicles[-1] = m
        del ego


F64_FIELDS = ["x", "y", "heading", "speed", "timer", "target_speed", "delta", "impact_x", "impact_y"]
I8_FIELDS = ["lane", "target_lane", "speed_index", "crashed", "has_impact", "check_collisions", "controlled",
             "obstacle", "present"]
LANE_F64 = ["x0", "y0", "length", "width", "amplitude", "pulsation", "phase", "speed_limit"]
LANE_I32 = ["road", "id", "road_first", "road_lanes", "next_first", "next_lanes", "forbidden"]


def lane_table(net) -> tuple:
    """Lanes in RoadNetwork.get_closest_lane_index order + index of every (from, to, id)."""
    index, rows = {}, []
    roads = []
    for _from, to_dict in net.graph.items():
        for _to, lanes in to_dict.items():
            roads.append((_from, _to))
            for _id, lane in enumerate(lanes):
                index[(_from, _to, _id)] = len(rows)
                rows.append((_from, _to, _id, lane))
    tab = {k: np.zeros(len(rows), np.float64) for k in LANE_F64}
    tab.update({k: np.zeros(len(rows), np.int32) for k in LANE_I32})
    for k, (_from, _to, _id, lane) in enumerate(rows):
        assert abs(lane.direction[0] - 1.0) == 0 and lane.direction[1] == 0, "x-aligned lanes only"
        tab["x0"][k], tab["y0"][k] = lane.start
        tab["length"][k] = lane.length
        tab["width"][k] = lane.width
        if isinstance(lane, SineLane):
            tab["amplitude"][k], tab["pulsation"][k], tab["phase"][k] = lane.amplitude, lane.pulsation, lane.phase
        tab["speed_limit"][k] = lane.speed_limit
        tab["forbidden"][k] = lane.forbidden
        tab["road"][k] = roads.index((_from, _to))
        tab["id"][k] = _id
        tab["road_first"][k] = index[(_from, _to, 0)]
        tab["road_lanes"][k] = len(net.graph[_from][_to])
        nxt = list(net.graph.get(_to, {}).keys())
        assert len(nxt) <= 1, "one successor road per node"
        if nxt:
            tab["next_first"][k] = -1  # patched below (the successor may not be indexed yet)
            tab["next_lanes"][k] = len(net.graph[_to][nxt[0]])
        else:
            tab["next_first"][k] = -1
            tab["next_lanes"][k] = 0
    for k, (_from, _to, _id, lane) in enumerate(rows):
        nxt = list(net.graph.get(_to, {}).keys())
        if nxt:
            tab["next_first"][k] = index[(_to, nxt[0], 0)]
    return tab, index


def dump_state(env, index, n_slots) -> dict:
    """Vehicles in list order, absent slots, then the road objects (obstacles) at the END."""
    vs, objs = env.road.vehicles, env.road.objects
    assert len(vs) + len(objs) <= n_slots
    out = {k: np.zeros(n_slots, np.float64) for k in F64_FIELDS}
    out.update({k: np.zeros(n_slots, np.int8) for k in I8_FIELDS})
    for i, v in enumerate(vs):
        out["present"][i] = 1
        out["x"][i], out["y"][i] = v.position
        out["heading"][i] = v.heading
        out["speed"][i] = v.speed
        out["timer"][i] = getattr(v, "timer", 0.0)
        out["target_speed"][i] = v.target_speed
        out["delta"][i] = v.DELTA if isinstance(v, IDMVehicle) else 0.0
        if v.impact is not None:
            out["impact_x"][i], out["impact_y"][i] = v.impact
            out["has_impact"][i] = 1
        out["lane"][i] = index[tuple(v.lane_index)]
        out["target_lane"][i] = index[tuple(v.target_lane_index)]
        out["speed_index"][i] = v.speed_index if isinstance(v, MDPVehicle) else 0
        out["crashed"][i] = v.crashed
        out["check_collisions"][i] = v.check_collisions
        out["controlled"][i] = any(v is c for c in env.controlled_vehicles)
    for k, o in enumerate(objs):
        assert isinstance(o, Obstacle)
        i = n_slots - len(objs) + k
        out["present"][i] = 1
        out["obstacle"][i] = 1
        out["x"][i], out["y"][i] = o.position
        out["heading"][i] = o.heading
        out["speed"][i] = o.speed
        out["lane"][i] = out["target_lane"][i] = index[tuple(o.lane_index)]
        out["crashed"][i] = o.crashed
        out["check_collisions"][i] = o.check_collisions
    return out


# features_range is given explicitly: left to its default, every agent's KinematicObservation would freeze its
# own y range from the road it happens to start on (observation.py:211-226), and the extra agents start anywhere.
MA_CFG = {"action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
          "observation": {"type": "MultiAgentObservation",
                          "observation_config": {"type": "Kinematics",
                                                 "features_range": {"x": [-200, 200], "y": [-16, 16],
                                                                    "vx": [-80, 80], "vy": [-80, 80]}}}}

SCENARIOS = [
    # merge-v0 defaults: ego + 3 IDM + merging vehicle + obstacle, 15 frames per step
    dict(name="merge_default", cls=MergeEnv, config={}, seeds=list(range(8)), steps=14, action_seed=31,
         frames_for=3, n_slots=6),
    # merge-generic-v0 with more lanes / traffic (rejection-sampled spawn => per-env vehicle counts differ)
    dict(name="merge_generic_l3", cls=MergeGenericEnv, config={"lanes_count": 3, "vehicles_count": 20},
         seeds=list(range(5)), steps=13, action_seed=32, frames_for=2, n_slots=23,
         action_p=[0.2, 0.2, 0.3, 0.2, 0.1]),
    # other section lengths, fewer frames per step, see_behind + absolute observation variant
    dict(name="merge_generic_sections", cls=MergeGenericEnv,
         config={"lanes_count": 2, "vehicles_count": 8, "before_merge_length": 100, "converge_merge_length": 60,
                 "parallel_merge_length": 120, "after_merge_length": 100, "simulation_frequency": 5,
                 "observation": {"type": "Kinematics", "see_behind": True, "vehicles_count": 6,
                                 "features": ["presence", "x", "y", "vx", "vy", "cos_h", "sin_h"]}},
         seeds=[7, 8, 9], steps=14, action_seed=33, frames_for=1, n_slots=11),
    # BASELINE config 5 shape: 4 lanes, 40 traffic vehicles, 4 controlled agents (see MergeGenericMultiAgent)
    dict(name="merge_ma4", cls=MergeGenericMultiAgent,
         config=dict(MA_CFG, lanes_count=4, vehicles_count=40, controlled_vehicles=4),
         seeds=[0, 1, 2], steps=11, action_seed=34, frames_for=1, n_slots=43),
    # merge-v1 / merge-generic-v1: Road.neighbour_vehicles also searches the connected lane segments (road.py:508-529)
    dict(name="merge_v1", cls=ConnectedLaneMergeEnv, config={}, seeds=list(range(8)), steps=14, action_seed=35,
         frames_for=3, n_slots=6),
    # KinematicObservation(include_obstacles=False): the Obstacle at the end of the acceleration lane is not observed
    # (observation.py:172,246 -> close_objects_to(vehicles_only=True)); a short ramp so that the ego gets close to it
    dict(name="merge_no_obstacles", cls=MergeEnv,
         config={"observation": {"type": "Kinematics", "include_obstacles": False, "vehicles_count": 6}},
         seeds=list(range(20, 26)), steps=12, action_seed=37, frames_for=1, n_slots=6,
         action_p=[0.05, 0.2, 0.45, 0.25, 0.05]),
    # OccupancyGridObservation on the merge network: vehicles only (no Obstacle), on-road layer over StraightLane and
    # SineLane waypoints (observation.py:454-484); world-aligned default grid and a vehicle-aligned finer one
    dict(name="merge_grid", cls=MergeEnv, config={"observation": {"type": "OccupancyGrid"}},
         seeds=list(range(30, 34)), steps=10, action_seed=38, frames_for=0, n_slots=6),
    dict(name="merge_grid_image", cls=MergeEnv, config={"observation": {"type": "OccupancyGrid", "as_image": True}},
         seeds=list(range(34, 37)), steps=8, action_seed=39, frames_for=0, n_slots=6),
    dict(name="merge_generic_grid_aligned", cls=MergeGenericEnv,
         config={"lanes_count": 3, "vehicles_count": 15,
                 "observation": {"type": "OccupancyGrid", "align_to_vehicle_axes": True, "grid_size": [[-24, 48], [-12, 12]],
                                 "grid_step": [3, 3], "features": ["presence", "x", "y", "vx", "vy", "cos_h", "on_road"],
                                 "features_range": {"x": [-60, 60], "y": [-20, 20], "vx": [-20, 20], "vy": [-10, 10]}}},
         seeds=[40, 41, 42], steps=9, action_seed=39, frames_for=0, n_slots=18, action_p=[0.2, 0.2, 0.3, 0.2, 0.1]),
    # crash-rich (round 3): FIRST crashes on the merge networks -- what a user sees WITH terminated=True.  Dense generic traffic
    # with a weaving / accelerating ego; the 4-agent BASELINE config-5 shape; and vehicle-vs-Obstacle hits (MergeRampEgo)
    dict(name="merge_crash_generic", cls=MergeGenericEnv, config={"lanes_count": 3, "vehicles_count": 30},
         seeds=list(range(400, 432)), steps=9, action_seed=71, frames_for=6, n_slots=33,
         action_p=[0.25, 0.05, 0.25, 0.4, 0.05]),
    dict(name="merge_crash_ma4", cls=MergeGenericMultiAgent,
         config=dict(MA_CFG, lanes_count=4, vehicles_count=40, controlled_vehicles=4),
         seeds=list(range(440, 452)), steps=8, action_seed=72, frames_for=2, n_slots=43,
         action_p=[0.25, 0.05, 0.25, 0.4, 0.05]),
    # (features_range explicit: by default KinematicObservation freezes its y range from the side lanes of the road the ego
    # STARTS on, observation.py:211-226 -- three lanes on b -> c, two on a -> b)
    dict(name="merge_crash_obstacle", cls=MergeRampEgo,
         config={"observation": {"type": "Kinematics",
                                 "features_range": {"x": [-200, 200], "y": [-12, 12], "vx": [-80, 80], "vy": [-80, 80]}}},
         seeds=list(range(460, 484)), steps=5, action_seed=73,
         frames_for=8, n_slots=6, action_p=[0.1, 0.35, 0.15, 0.35, 0.05]),
    dict(name="merge_generic_v1", cls=ConnectedLaneMergeGenericEnv, config={"lanes_count": 3, "vehicles_count": 20},
         seeds=list(range(5)), steps=13, action_seed=36, frames_for=2, n_slots=23,
         action_p=[0.2, 0.2, 0.3, 0.2, 0.1]),
]


def run_scenario(sc: dict, only_envs=None) -> dict:
    """`only_envs`: simulate only these env indices (tests/test_fixture_freshness.py regenerates env 0 of every fixture); the action
    table is drawn for all of them either way, so an env's trajectory does not depend on which others are simulated."""
    ref_stub.restore_class_defaults()   # (an IntersectionEnv created earlier in this process has re-tuned the IDM CLASS)
    seeds, steps, n_slots = sc["seeds"], sc["steps"], sc["n_slots"]
    E = len(seeds)
    A = int(sc["config"].get("controlled_vehicles", 1))
    rng = np.random.default_rng(sc["action_seed"])
    p = sc.get("action_p")
    actions = (rng.choice(5, size=(steps, E, A), p=p) if p is not None
               else rng.integers(0, 5, size=(steps, E, A))).astype(np.int32)
    out: dict = {"seeds": np.asarray(seeds, np.int64), "actions": actions}
    per_env = []
    frames_for = sc["frames_for"]
    tab0 = None
    for e, seed in enumerate(seeds):
        if only_envs is not None and e not in only_envs:
            continue
        env = sc["cls"](dict(sc["config"]))
        obs0, _ = env.reset(seed=int(seed))
        tab, index = lane_table(env.road.network)
        if tab0 is None:
            tab0 = tab
        T = int(env.config["simulation_frequency"] // env.config["policy_frequency"])
        rec = {"obs0": np.asarray(obs0), "init": dump_state(env, index, n_slots), "obs": [], "reward": [],
               "terminated": [], "truncated": [], "speed": [], "crashed": [], "step_state": [], "frames": []}
        if e < frames_for:
            road = env.road
            orig_step = road.step

            def step_and_dump(dt, _orig=orig_step, _env=env, _rec=rec, _index=index):
                _orig(dt)
                _rec["frames"].append(dump_state(_env, _index, n_slots))

            road.step = step_and_dump
        for t in range(steps):
            a = actions[t, e]
            o, r, te, tr, info = env.step(tuple(int(x) for x in a) if A > 1 else int(a[0]))
            rec["obs"].append(np.asarray(o))
            rec["reward"].append(r)
            rec["terminated"].append(te)
            rec["truncated"].append(tr)
            rec["speed"].append(info["speed"])
            rec["crashed"].append(info["crashed"])
            rec["step_state"].append(dump_state(env, index, n_slots))
        rec["T"] = T
        rec["cfg"] = dict(env.config)
        rec["end_position"] = float(getattr(env, "end_position", 0) or 370.0)
        per_env.append(rec)
    E = len(per_env)
    cfg = per_env[0]["cfg"]
    out["meta"] = np.asarray([E, n_slots, per_env[0]["T"], steps, frames_for, A], np.int64)
    out["cfg_json"] = np.asarray(json.dumps({k: v for k, v in cfg.items()
                                             if isinstance(v, (int, float, str, bool, list, dict, type(None)))}))
    out["cfg_generic"] = np.int64(issubclass(sc["cls"], MergeGenericEnv))
    out["end_position"] = np.float64(per_env[0]["end_position"])
    for k in LANE_F64 + LANE_I32:
        out["lane_" + k] = tab0[k]
    # per-agent observation shape: (V, F) Kinematics or (F, W, H) OccupancyGrid; a single agent has no agent axis
    one = np.asarray(per_env[0]["obs0"])
    oshape = one.shape[1:] if A > 1 else one.shape
    out["obs0"] = np.stack([r["obs0"] for r in per_env]).reshape(E, A, *oshape)
    obs = np.stack([np.stack(r["obs"]) for r in per_env], axis=1)                  # [steps,E,(A,)...]
    out["obs"] = obs.reshape(steps, E, A, *oshape)
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
        n_veh = data["init_present"].sum(axis=1) - data["init_obstacle"].sum(axis=1)
        print(f"{sc['name']}: E,N,T,steps,frames_for,A={data['meta'].tolist()} vehicles/env={n_veh.tolist()} "
              f"terminated_any={bool(data['terminated'].any())} "
              f"crashed_total={int(data['step_crashed'][-1].sum())} "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
