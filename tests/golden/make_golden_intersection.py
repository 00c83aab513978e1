#!/usr/bin/env python3
"""Generate the intersection golden fixtures from the UNMODIFIED reference.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_intersection.py

Drives ``IntersectionEnv`` (``highway_env/envs/intersection_env.py``) through the ``oracle/ref_stub.py`` import
shim and records, per environment,

* the lane table of the reference's ``RoadNetwork`` in ``get_closest_lane_index`` iteration order
  (``road/road.py:55-71``): straight lanes of any direction and ``CircularLane``s, with priorities;
* the full state of every vehicle (``Road.vehicles`` order, compact; the list grows and shrinks between policy
  steps: ``IntersectionEnv.step`` clears leaving vehicles and spawns new ones, intersection_env.py:136-140) after
  ``reset(seed=s)``, after every simulation frame (for the first ``frames_for`` environments), after every policy
  step BEFORE clear/spawn (what ``observe`` / ``_reward`` saw) and AFTER clear/spawn (where the next step starts),
  including the planned route, the yielding flag / timer of ``RegulatedRoad`` and a unique vehicle id;
* ``RegulatedRoad.steps`` (the regulation runs every ``int(1 / dt / 2)`` frames, road/regulation.py:36-40);
* every draw ``_spawn_vehicle`` takes from ``np_random`` per policy step (a logging proxy around the Generator),
  so that the spawn RULE can be replayed on the same numbers;
* ``obs, reward, terminated, truncated, info["speed"|"crashed"]`` per step.
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

from highway_env.envs.intersection_env import (ConnectedLaneIntersectionEnv, IntersectionEnv,  # noqa: E402
                                               MultiAgentIntersectionEnv)
from highway_env.road.lane import CircularLane, StraightLane  # noqa: E402
from highway_env.vehicle.behavior import IDMVehicle  # noqa: E402
from highway_env.vehicle.controller import MDPVehicle  # noqa: E402

R_MAX = 4       # route entries kept per vehicle (single-agent routes hold at most 3; multi-agent fixtures use R_MAX_MA)
R_MAX_MA = 8    # MultiAgentIntersectionEnv: the second agent's default route o1 -> o1 loops through another arm (6 roads)
DRAWS_MAX = 8   # np_random results logged per policy step (a spawn takes at most 6)

F64_FIELDS = ["x", "y", "heading", "speed", "timer", "target_speed", "delta", "impact_x", "impact_y"]
INT_FIELDS = ["present", "lane", "target_lane", "speed_index", "crashed", "has_impact", "controlled", "is_yielding",
              "yield_timer", "route_len", "vid"]
ROUTE_FIELDS = ["route_from", "route_to", "route_id"]
LANE_F64 = ["sx", "sy", "ex", "ey", "heading", "dirx", "diry", "cx", "cy", "radius", "start_phase", "end_phase",
            "length", "width", "speed_limit"]
LANE_I32 = ["kind", "direction", "priority", "forbidden", "from_node", "to_node", "id"]


class LoggingRandom:
    """Delegates to the Generator and logs every result (flattened) into ``self.log``."""

    def __init__(self, gen):
        self._gen = gen
        self.log = []

    def _rec(self, out):
        self.log.extend(np.asarray(out, np.float64).ravel().tolist())
        return out

    def uniform(self, *a, **k):
        return self._rec(self._gen.uniform(*a, **k))

    def normal(self, *a, **k):
        return self._rec(self._gen.normal(*a, **k))

    def choice(self, *a, **k):
        return self._rec(self._gen.choice(*a, **k))

    def integers(self, *a, **k):
        return self._rec(self._gen.integers(*a, **k))

    def __getattr__(self, name):
        return getattr(self._gen, name)


def lane_table(net) -> tuple:
    nodes = {name: k for k, name in enumerate(net.graph.keys())}
    index, rows = {}, []
    for _from, to_dict in net.graph.items():
        for _to, lanes in to_dict.items():
            assert _to in nodes, "every node has outgoing roads in this network"
            for _id, lane in enumerate(lanes):
                index[(_from, _to, _id)] = len(rows)
                rows.append((_from, _to, _id, lane))
    tab = {k: np.zeros(len(rows), np.float64) for k in LANE_F64}
    tab.update({k: np.zeros(len(rows), np.int32) for k in LANE_I32})
    for k, (_from, _to, _id, lane) in enumerate(rows):
        tab["from_node"][k], tab["to_node"][k], tab["id"][k] = nodes[_from], nodes[_to], _id
        tab["length"][k], tab["width"][k] = lane.length, lane.width
        tab["speed_limit"][k], tab["priority"][k], tab["forbidden"][k] = lane.speed_limit, lane.priority, lane.forbidden
        if isinstance(lane, CircularLane):
            tab["kind"][k] = 1
            tab["cx"][k], tab["cy"][k] = lane.center
            tab["radius"][k] = lane.radius
            tab["start_phase"][k], tab["end_phase"][k] = lane.start_phase, lane.end_phase
            tab["direction"][k] = lane.direction
        else:
            assert type(lane) is StraightLane
            tab["sx"][k], tab["sy"][k] = lane.start
            tab["ex"][k], tab["ey"][k] = lane.end
            tab["heading"][k] = lane.heading
            tab["dirx"][k], tab["diry"][k] = lane.direction
    return tab, index, nodes


def dump_state(env, index, nodes, n_slots, vids, r_max=R_MAX) -> dict:
    vs = env.road.vehicles
    assert len(vs) <= n_slots and not env.road.objects
    out = {k: np.zeros(n_slots, np.float64) for k in F64_FIELDS}
    out.update({k: np.zeros(n_slots, np.int32) for k in INT_FIELDS})
    out.update({k: np.full((n_slots, r_max), -1, np.int32) for k in ROUTE_FIELDS})
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
        out["controlled"][i] = any(v is c for c in env.controlled_vehicles)
        out["is_yielding"][i] = bool(getattr(v, "is_yielding", False))
        out["yield_timer"][i] = int(getattr(v, "yield_timer", 0))
        assert v.route is not None and len(v.route) <= r_max
        out["route_len"][i] = len(v.route)
        for k, (_f, _t, _i) in enumerate(v.route):
            out["route_from"][i, k], out["route_to"][i, k] = nodes[_f], nodes[_t]
            out["route_id"][i, k] = -1 if _i is None else _i
        # (the table keeps the vehicle alive: CPython reuses the id() of a collected object, which made a spawned vehicle
        #  inherit the number of a cleared one now and then -- and the fixture depend on the allocator)
        out["vid"][i] = vids.setdefault(id(v), (len(vids), v))[0]
        assert v.check_collisions and isinstance(v, (IDMVehicle, MDPVehicle))
    return out


SCENARIOS = [
    # intersection-v0 defaults: 3 actions (SLOWER, IDLE, FASTER), Kinematics 15 x 7 absolute, duration 13
    dict(name="intersection_default", config={}, seeds=list(range(6)), steps=13, action_seed=41, frames_for=3,
         n_slots=24),
    # denser traffic, longer episodes (more yielding and more crashes to follow)
    dict(name="intersection_dense", config={"initial_vehicle_count": 14, "spawn_probability": 0.9, "duration": 20},
         seeds=[11, 12, 13, 14], steps=18, action_seed=42, frames_for=2, n_slots=32),
    # BASELINE config 4's observation: OccupancyGrid with its defaults (presence, vx, vy, on_road; 11 x 11 cells of 5 m)
    dict(name="intersection_grid", config={"observation": {"type": "OccupancyGrid"}}, seeds=[21, 22, 23], steps=12,
         action_seed=43, frames_for=0, n_slots=24),
    # vehicle-aligned grid with position / heading layers and an x / y range (the normalise -> denormalise round trip)
    dict(name="intersection_grid_aligned",
         config={"observation": {"type": "OccupancyGrid", "align_to_vehicle_axes": True, "grid_size": [[-32, 32], [-16, 16]],
                                 "grid_step": [4, 4], "features": ["presence", "x", "y", "vx", "vy", "cos_h", "sin_h", "on_road"],
                                 "features_range": {"x": [-40, 40], "y": [-40, 40], "vx": [-20, 20], "vy": [-20, 20]}}},
         seeds=[31, 32], steps=12, action_seed=44, frames_for=0, n_slots=24),
    # OccupancyGrid as_image (uint8 cells)
    dict(name="intersection_grid_image", config={"observation": {"type": "OccupancyGrid", "as_image": True}}, seeds=[24, 25, 26],
         steps=10, action_seed=52, frames_for=0, n_slots=24),
    # config["destination"] = None: "o" + str(np_random.integers(1, 4)) per episode (intersection_env.py:295-297)
    dict(name="intersection_random_destination", config={"destination": None}, seeds=list(range(51, 59)), steps=8,
         action_seed=46, frames_for=0, n_slots=24),
    # MultiAgentIntersectionEnv (intersection-multi-agent-v0, intersection_env.py:376-420): 2 agents, MultiAgentAction /
    # MultiAgentObservation, the second agent's route o1 -> o1 (6 roads); and 3 agents with random destinations
    dict(name="intersection_multi_agent", cls="MultiAgentIntersectionEnv", config={}, seeds=list(range(61, 67)), steps=10,
         action_seed=47, frames_for=2, n_slots=24),
    dict(name="intersection_multi_agent3", cls="MultiAgentIntersectionEnv",
         config={"controlled_vehicles": 3, "destination": None, "initial_vehicle_count": 8}, seeds=list(range(71, 75)),
         steps=9, action_seed=48, frames_for=0, n_slots=24),
    # the destination features cos_d / sin_d (Vehicle.destination_direction, kinematics.py:211-235): with observe_intentions
    # every observed vehicle's, without it (the default) only the observer's own (objects.py / kinematics.py to_dict)
    dict(name="intersection_intentions",
         config={"observation": {"type": "Kinematics", "vehicles_count": 10,
                                 "features": ["presence", "x", "y", "vx", "vy", "cos_h", "sin_h", "cos_d", "sin_d"],
                                 "features_range": {"x": [-100, 100], "y": [-100, 100], "vx": [-20, 20], "vy": [-20, 20]},
                                 "absolute": False, "observe_intentions": True}, "destination": None},
         seeds=list(range(81, 85)), steps=9, action_seed=49, frames_for=0, n_slots=24),
    dict(name="intersection_no_intentions",
         config={"observation": {"type": "Kinematics", "vehicles_count": 15,
                                 "features": ["presence", "x", "y", "cos_d", "sin_d"],
                                 "features_range": {"x": [-100, 100], "y": [-100, 100]},
                                 "absolute": True, "observe_intentions": False}},
         seeds=list(range(91, 94)), steps=8, action_seed=50, frames_for=0, n_slots=24),
    # the lane-offset features (Vehicle.lane_offset, kinematics.py:228-235) on straight lanes of every direction and on arcs
    dict(name="intersection_lane_offsets",
         config={"observation": {"type": "Kinematics", "vehicles_count": 12,
                                 "features": ["presence", "x", "y", "long_off", "lat_off", "ang_off"],
                                 "features_range": {"x": [-100, 100], "y": [-100, 100]}, "absolute": True},
                 "initial_vehicle_count": 12, "spawn_probability": 0.8},
         seeds=list(range(101, 105)), steps=10, action_seed=51, frames_for=0, n_slots=24),
    # intersection-v2: Road.neighbour_vehicles also searches the connected lane segments (road.py:508-529)
    dict(name="intersection_v2", cls="ConnectedLaneIntersectionEnv",
         config={"initial_vehicle_count": 12, "spawn_probability": 0.8, "duration": 16},
         seeds=[41, 42, 43, 44], steps=15, action_seed=45, frames_for=2, n_slots=32),
    # crash-rich (round 3): dense traffic and an ego that mostly accelerates through the junction -- many FIRST crashes, i.e. the
    # observation / reward a user sees WITH terminated=True, plus per-frame states (signed impacts) for the first environments
    dict(name="intersection_crash", config={"initial_vehicle_count": 16, "spawn_probability": 0.9, "duration": 20},
         seeds=list(range(300, 332)), steps=9, action_seed=61, frames_for=6, n_slots=32, action_p=[0.1, 0.2, 0.7]),
    # the same with three agents (any crashed agent terminates the episode, intersection_env.py:100-106)
    dict(name="intersection_crash_ma3", cls="MultiAgentIntersectionEnv",
         config={"controlled_vehicles": 3, "initial_vehicle_count": 12, "spawn_probability": 0.9, "duration": 20},
         seeds=list(range(340, 356)), steps=8, action_seed=62, frames_for=2, n_slots=32, action_p=[0.1, 0.2, 0.7]),
]


def run_scenario(sc: dict, only_envs=None) -> dict:
    """`only_envs`: simulate only these env indices (tests/test_fixture_freshness.py regenerates env 0 of every fixture); the action
    table is drawn for all of them either way, so an env's trajectory does not depend on which others are simulated."""
    ref_stub.restore_class_defaults()   # (an IntersectionEnv created earlier in this process has re-tuned the IDM CLASS)
    seeds, steps, n_slots, frames_for = sc["seeds"], sc["steps"], sc["n_slots"], sc["frames_for"]
    E = len(seeds)
    cls = {"ConnectedLaneIntersectionEnv": ConnectedLaneIntersectionEnv, "MultiAgentIntersectionEnv": MultiAgentIntersectionEnv}.get(
        sc.get("cls"), IntersectionEnv)
    multi = cls is MultiAgentIntersectionEnv
    A = int(dict(cls.default_config(), **sc["config"])["controlled_vehicles"])
    r_max = int(sc.get("r_max", R_MAX_MA if multi else R_MAX))   # (route slots recorded per vehicle)
    rng = np.random.default_rng(sc["action_seed"])
    p = sc.get("action_p")
    actions = (rng.choice(3, size=(steps, E, A), p=p) if p is not None
               else rng.integers(0, 3, size=(steps, E, A))).astype(np.int32)
    out: dict = {"seeds": np.asarray(seeds, np.int64), "actions": actions}
    per_env, tab0 = [], None
    for e, seed in enumerate(seeds):
        if only_envs is not None and e not in only_envs:
            continue
        env = cls(dict(sc["config"]))
        obs0, _ = env.reset(seed=int(seed))
        tab, index, nodes = lane_table(env.road.network)
        tab0 = tab if tab0 is None else tab0
        T = int(env.config["simulation_frequency"] // env.config["policy_frequency"])
        vids: dict = {}
        proxy = LoggingRandom(env.np_random)
        env.np_random = proxy
        env.road.np_random = proxy
        rec = {"obs0": np.asarray(obs0), "init": dump_state(env, index, nodes, n_slots, vids, r_max),
               "road_steps0": env.road.steps, "obs": [], "reward": [], "terminated": [], "truncated": [], "speed": [],
               "crashed": [], "step_state": [], "next_state": [], "frames": [], "draws": [], "n_draws": [],
               "agents_rewards": [], "agents_terminated": []}
        if e < frames_for:
            road = env.road
            orig_step = road.step

            def step_and_dump(dt, _orig=orig_step, _env=env, _rec=rec, _index=index, _nodes=nodes, _vids=vids):
                _orig(dt)
                _rec["frames"].append(dump_state(_env, _index, _nodes, n_slots, _vids, r_max))

            road.step = step_and_dump
        orig_clear = env._clear_vehicles

        def clear_and_dump(_orig=orig_clear, _env=env, _rec=rec, _index=index, _nodes=nodes, _vids=vids):
            _rec["step_state"].append(dump_state(_env, _index, _nodes, n_slots, _vids, r_max))
            _orig()

        env._clear_vehicles = clear_and_dump
        for t in range(steps):
            proxy.log = []
            o, r, te, tr, info = env.step(tuple(int(a) for a in actions[t, e]) if multi else int(actions[t, e, 0]))
            rec["agents_rewards"].append(np.asarray(info["agents_rewards"], np.float64))
            rec["agents_terminated"].append(np.asarray(info["agents_terminated"], np.int8))
            rec["obs"].append(np.asarray(o))
            rec["reward"].append(r)
            rec["terminated"].append(te)
            rec["truncated"].append(tr)
            rec["speed"].append(info["speed"])
            rec["crashed"].append(info["crashed"])
            rec["next_state"].append(dump_state(env, index, nodes, n_slots, vids, r_max))
            assert len(proxy.log) <= DRAWS_MAX
            rec["n_draws"].append(len(proxy.log))
            rec["draws"].append(proxy.log + [0.0] * (DRAWS_MAX - len(proxy.log)))
        rec["T"], rec["cfg"], rec["nodes"] = T, dict(env.config), nodes
        per_env.append(rec)
    E = len(per_env)
    cfg = per_env[0]["cfg"]
    out["meta"] = np.asarray([E, n_slots, per_env[0]["T"], steps, frames_for, A, r_max], np.int64)
    out["agents_rewards"] = np.stack([np.stack(r["agents_rewards"]) for r in per_env], axis=1)        # [steps,E,A]
    out["agents_terminated"] = np.stack([np.stack(r["agents_terminated"]) for r in per_env], axis=1)  # [steps,E,A]
    out["cfg_json"] = np.asarray(json.dumps({k: v for k, v in cfg.items()
                                             if isinstance(v, (int, float, str, bool, list, dict, type(None)))}))
    for k in LANE_F64 + LANE_I32:
        out["lane_" + k] = tab0[k]
    out["node_names"] = np.asarray(list(per_env[0]["nodes"].keys()))
    out["road_steps0"] = np.asarray([r["road_steps0"] for r in per_env], np.int64)
    out["obs0"] = np.stack([r["obs0"] for r in per_env])
    out["obs"] = np.stack([np.stack(r["obs"]) for r in per_env], axis=1)          # [steps,E,V,F]
    out["reward"] = np.asarray([r["reward"] for r in per_env], np.float64).T      # [steps,E]
    out["terminated"] = np.asarray([r["terminated"] for r in per_env], np.int8).T
    out["truncated"] = np.asarray([r["truncated"] for r in per_env], np.int8).T
    out["info_speed"] = np.asarray([r["speed"] for r in per_env], np.float64).T
    out["info_crashed"] = np.asarray([r["crashed"] for r in per_env], np.int8).T
    out["draws"] = np.asarray([r["draws"] for r in per_env], np.float64).transpose(1, 0, 2)   # [steps,E,DRAWS_MAX]
    out["n_draws"] = np.asarray([r["n_draws"] for r in per_env], np.int32).T
    for k in F64_FIELDS + INT_FIELDS + ROUTE_FIELDS:
        out["init_" + k] = np.stack([r["init"][k] for r in per_env])
        out["step_" + k] = np.stack([np.stack([s[k] for s in r["step_state"]]) for r in per_env], axis=1)
        out["next_" + k] = np.stack([np.stack([s[k] for s in r["next_state"]]) for r in per_env], axis=1)
        if frames_for:
            out["frame_" + k] = np.stack([np.stack([s[k] for s in r["frames"]])
                                          for r in per_env[:frames_for]], axis=1)  # [steps*T,Ef,N(,R)]
    return out


def main() -> None:
    only = set(sys.argv[1:])
    for sc in SCENARIOS:
        if only and sc["name"] not in only:
            continue
        data = run_scenario(sc)
        path = os.path.join(HERE, sc["name"] + ".npz")
        np.savez_compressed(path, **data)
        print(f"{sc['name']}: E,N,T,steps,frames_for,A,R={data['meta'].tolist()} "
              f"vehicles/env at reset={data['init_present'].sum(axis=1).tolist()} "
              f"max vehicles={int(data['next_present'].sum(axis=2).max())} "
              f"yielding frames={int(data['frame_is_yielding'].sum()) if 'frame_is_yielding' in data else -1} "
              f"terminated={data['terminated'].any(axis=0).astype(int).tolist()} "
              f"crashed_total={int(data['step_crashed'][-1].sum())} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
