"""Intersection scenario: config dict, the lane table of the 4-way junction, routes, and the host-side traffic
management on numpy's stream.

Mirrors ``highway_env/envs/intersection_env.py`` (``IntersectionEnv``, intersection-v0): four access roads
``o_k -> ir_k``, for each of them a right turn, a left turn (``CircularLane``) and a crossing lane
``ir_k -> il_j``, and four exit roads ``il_k -> o_k``; the horizontal road has priority (``_make_road``, :142-230).
The network travels through the ABI as ``hwy_config.gnet`` (``include/hwy_engine.h``: ``hwy_glane``) in the
iteration order of ``RoadNetwork.get_closest_lane_index`` (road/road.py:55-71); every road has one lane, so a
planned route (``ControlledVehicle.plan_route_to``, vehicle/controller.py:71-87) is a list of table indices.

The reference draws traffic from ``np_random`` WHILE an episode runs (``IntersectionEnv.step`` clears leaving
vehicles and spawns new ones, :136-140) and simulates three seconds inside ``reset`` (:232-290).  Two modes:

* ``HWY_C_HOST_TRAFFIC`` (reference stream): this module replays ``_clear_vehicles`` / ``_spawn_vehicle`` /
  ``_make_vehicles`` on the env's numpy Generator in the reference's draw order and moves the state through
  ``hwy_get_state`` / ``hwy_set_state``; the three warm-up seconds of ``reset`` run on the engine;
* device traffic (default for throughput): the step kernel does all of it on Philox draws (csrc/hwy_ix.h).
"""
from __future__ import annotations

import numpy as np

from . import _abi

NODE_NAMES = ["o0", "ir0", "il3", "o1", "ir1", "il0", "o2", "ir2", "il1", "o3", "ir3", "il2"]  # graph key order
VEH_LENGTH = 5.0


def intersection_default_config() -> dict:
    """IntersectionEnv.default_config (intersection_env.py:17-58)."""
    cfg = _abi.abstract_default_config()
    cfg.update({
        "observation": {"type": "Kinematics", "vehicles_count": 15,
                        "features": ["presence", "x", "y", "vx", "vy", "cos_h", "sin_h"],
                        "features_range": {"x": [-100, 100], "y": [-100, 100], "vx": [-20, 20], "vy": [-20, 20]},
                        "absolute": True, "flatten": False, "observe_intentions": False},
        "action": {"type": "DiscreteMetaAction", "longitudinal": True, "lateral": False, "target_speeds": [0, 4.5, 9]},
        "duration": 13,
        "destination": "o1",
        "controlled_vehicles": 1,
        "initial_vehicle_count": 10,
        "spawn_probability": 0.6,
        "collision_reward": -5,
        "high_speed_reward": 1,
        "arrived_reward": 1,
        "reward_speed_range": [7.0, 9.0],
        "normalize_reward": False,
        "offroad_terminal": False,
    })
    return cfg


# --------------------------------------------------------------------------- lane table (IntersectionEnv._make_road)
def lane_table() -> dict:
    """The 20 lanes of the junction as arrays (keys: _abi.GLANE_F64 + _abi.GLANE_I32), built with the reference's
    own numpy expressions (intersection_env.py:155-222; StraightLane / CircularLane ctors, road/lane.py:162-194,
    314-339) so that every coordinate is bit-identical."""
    lane_width = 4.0
    right_turn_radius = lane_width + 5
    left_turn_radius = right_turn_radius + lane_width
    outer_distance = right_turn_radius + lane_width / 2
    access_length = 50 + 50
    node = {n: k for k, n in enumerate(NODE_NAMES)}
    rows = []

    def straight(_from, _to, start, end, priority):
        start, end = np.array(start), np.array(end)
        length = np.linalg.norm(end - start)
        direction = (end - start) / length
        rows.append(dict(kind=0, direction=0, priority=priority, forbidden=0, from_node=node[_from], to_node=node[_to],
                         exit_lane=int("il" in _from and "o" in _to), sx=start[0], sy=start[1],
                         heading=np.arctan2(end[1] - start[1], end[0] - start[0]), dirx=direction[0], diry=direction[1],
                         cx=0.0, cy=0.0, radius=0.0, start_phase=0.0, length=length, width=lane_width, speed_limit=10.0))

    def circular(_from, _to, center, radius, start_phase, end_phase, clockwise, priority):
        direction = 1 if clockwise else -1
        rows.append(dict(kind=1, direction=direction, priority=priority, forbidden=0, from_node=node[_from],
                         to_node=node[_to], exit_lane=0, sx=0.0, sy=0.0, heading=0.0, dirx=0.0, diry=0.0,
                         cx=center[0], cy=center[1], radius=radius, start_phase=start_phase,
                         length=radius * (end_phase - start_phase) * direction, width=lane_width, speed_limit=10.0))

    for corner in range(4):
        angle = np.radians(90 * corner)
        is_horizontal = corner % 2
        priority = 3 if is_horizontal else 1
        rotation = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])
        o, ir = f"o{corner}", f"ir{corner}"
        straight(o, ir, rotation @ np.array([lane_width / 2, access_length + outer_distance]),
                 rotation @ np.array([lane_width / 2, outer_distance]), priority)
        circular(ir, f"il{(corner - 1) % 4}", rotation @ np.array([outer_distance, outer_distance]), right_turn_radius,
                 angle + np.radians(180), angle + np.radians(270), True, priority)
        circular(ir, f"il{(corner + 1) % 4}",
                 rotation @ np.array([-left_turn_radius + lane_width / 2, left_turn_radius - lane_width / 2]),
                 left_turn_radius, angle + np.radians(0), angle + np.radians(-90), False, priority - 1)
        straight(ir, f"il{(corner + 2) % 4}", rotation @ np.array([lane_width / 2, outer_distance]),
                 rotation @ np.array([lane_width / 2, -outer_distance]), priority)
        start = rotation @ np.flip([lane_width / 2, access_length + outer_distance], axis=0)
        end = rotation @ np.flip([lane_width / 2, outer_distance], axis=0)
        straight(f"il{(corner - 1) % 4}", f"o{(corner - 1) % 4}", end, start, priority)
    tab = {k: np.array([r[k] for r in rows], np.float64) for k in _abi.GLANE_F64}
    tab.update({k: np.array([r[k] for r in rows], np.int32) for k in _abi.GLANE_I32})
    return tab


def lane_index_of(tab: dict, _from: str, _to: str) -> int:
    f, t = NODE_NAMES.index(_from), NODE_NAMES.index(_to)
    hit = np.nonzero((tab["from_node"] == f) & (tab["to_node"] == t))[0]
    return int(hit[0])


def table_from_config(c: _abi.HwyConfig) -> dict:
    n = c.gnet_lanes
    tab = {k: np.array([getattr(c.gnet[i], k) for i in range(n)], np.float64) for k in _abi.GLANE_F64}
    tab.update({k: np.array([getattr(c.gnet[i], k) for i in range(n)], np.int32) for k in _abi.GLANE_I32})
    return tab


def fill_config(c: _abi.HwyConfig, cfg: dict) -> None:
    """The intersection-specific part of _abi.make_config."""
    if not (1 <= int(cfg.get("controlled_vehicles", 1)) <= 4):
        raise ValueError("the intersection holds 1..4 controlled vehicles (one per access road; a fifth would be "
                         "created on top of the first, intersection_env.py:292-294)")
    dest = cfg.get("destination")
    # config["destination"] or "o" + str(np_random.integers(1, 4)) (intersection_env.py:295-297): None / "" = a random exit
    if dest and not (isinstance(dest, str) and dest in ("o0", "o1", "o2", "o3")):
        raise ValueError("destination must be one of 'o0'..'o3', or None for a random one")
    c.scenario = _abi.SCENARIO_INTERSECTION
    c.num_vehicles = int(cfg.get("max_vehicles", 32))  # slots per environment (the list grows while an episode runs)
    if not (4 <= c.num_vehicles <= 64):
        raise ValueError("max_vehicles must be in [4, 64] (one wavefront per environment)")
    for a in range(int(cfg.get("controlled_vehicles", 1))):
        c.agent_index[a] = a      # (unused: agent a is the a-th slot with F_CONTROLLED; slots move when the list is re-compacted)
    c.lanes_count = 1
    c.duration = float(cfg["duration"])
    c.road_length = 100.0
    c.speed_limit = 10.0
    tab = lane_table()
    c.gnet_lanes = len(tab["kind"])
    for k in range(c.gnet_lanes):
        for f in _abi.GLANE_F64:
            setattr(c.gnet[k], f, float(tab[f][k]))
        for f in _abi.GLANE_I32:
            setattr(c.gnet[k], f, int(tab[f][k]))
    for q in range(4):
        c.access_lane[q] = lane_index_of(tab, f"o{q}", f"ir{q}")
        c.exit_of[q] = lane_index_of(tab, f"il{q}", f"o{q}")
    c.destination = int(dest[1]) if dest else -1  # -1: drawn per episode (1..3)
    rt = route_table(tab)  # plan_route_to for every (lane, destination): BFS on the host, looked up by the kernel
    for k in range(c.gnet_lanes):
        for q in range(4):
            c.gnet_routes[k][q] = int(rt[k, q])
    c.initial_vehicle_count = int(cfg["initial_vehicle_count"])
    c.spawn_probability = float(cfg["spawn_probability"])
    c.arrived_reward = float(cfg["arrived_reward"])
    # IntersectionEnv._make_vehicles sets these on the traffic class (intersection_env.py:243-247)
    c.idm_distance_wanted, c.idm_time_wanted, c.idm_comfort_acc_max, c.idm_comfort_acc_min = 7.0, 1.5, 6.0, -3.0


# --------------------------------------------------------------------------- routes
def route_pack(lanes) -> int:
    """Route word of hwy_state.route: gnet indices of the remaining roads, 5 bits each from bit 0, their number in bits 56..59."""
    lanes = list(lanes)
    assert len(lanes) <= _abi.HWY_MAX_ROUTE
    w = len(lanes) << 56
    for k, l in enumerate(lanes):
        w |= (int(l) & 0x1f) << (5 * k)
    return w


def route_unpack(w: int) -> list:
    return [(int(w) >> (5 * k)) & 0x1f for k in range((int(w) >> 56) & 0xf)]


def shortest_path(tab: dict, start: int, goal: int) -> list:
    """RoadNetwork.shortest_path (road.py:159-188) on node ids: the first path of the breadth-first search whose frontier
    expands the neighbours of a node in SORTED NAME order and never revisits a node of its own path; [] if there is none."""
    out = {}
    for k in range(len(tab["kind"])):
        out.setdefault(int(tab["from_node"][k]), set()).add(int(tab["to_node"][k]))
    queue = [(start, [start])]
    while queue:
        node, path = queue.pop(0)
        if node not in out:  # `yield []` for a node without successors: shortest_path returns it as "no path"
            return []
        for nxt in sorted((n for n in out[node] if n not in path), key=lambda n: NODE_NAMES[n]):
            if nxt == goal:
                return path + [nxt]
            if nxt in out:
                queue.append((nxt, path + [nxt]))
    return []


def plan_route(tab: dict, lane: int, dest: int) -> list:
    """plan_route_to("o" + dest) (controller.py:71-87): [lane_index] + the roads of the shortest path from lane_index[1]
    (one lane per road on this network: a road == its gnet index)."""
    goal = NODE_NAMES.index(f"o{dest}")
    path = shortest_path(tab, int(tab["to_node"][lane]), goal)
    roads = [lane]
    for a, b in zip(path[:-1], path[1:]):
        hit = np.nonzero((tab["from_node"] == a) & (tab["to_node"] == b))[0]
        roads.append(int(hit[0]))
    return roads


def route_table(tab: dict) -> np.ndarray:
    """hwy_config.gnet_routes: for every lane L and destination k the route word of plan_route(L, k)[1:]."""
    n = len(tab["kind"])
    out = np.zeros((n, 4), np.int64)
    for lane in range(n):
        for k in range(4):
            out[lane, k] = route_pack(plan_route(tab, lane, k)[1:])
    return out


# --------------------------------------------------------------------------- lane geometry on the host
def lane_position(tab: dict, k: int, s: float) -> np.ndarray:
    if tab["kind"][k] == 0:   # StraightLane.position (lane.py:196-201)
        start = np.array([tab["sx"][k], tab["sy"][k]])
        direction = np.array([tab["dirx"][k], tab["diry"][k]])
        lateral = np.array([-direction[1], direction[0]])
        return start + s * direction + 0.0 * lateral
    phi = tab["direction"][k] * s / tab["radius"][k] + tab["start_phase"][k]   # CircularLane.position (:341-345)
    return np.array([tab["cx"][k], tab["cy"][k]]) + (tab["radius"][k] - 0.0 * tab["direction"][k]) * np.array(
        [np.cos(phi), np.sin(phi)])


def lane_heading_at(tab: dict, k: int, s: float) -> float:
    if tab["kind"][k] == 0:
        return float(tab["heading"][k])
    phi = tab["direction"][k] * s / tab["radius"][k] + tab["start_phase"][k]
    return float(phi + np.pi / 2 * tab["direction"][k])


def lane_local(tab: dict, k: int, pos) -> tuple:
    if tab["kind"][k] == 0:   # StraightLane.local_coordinates (lane.py:209-213)
        delta = np.asarray(pos) - np.array([tab["sx"][k], tab["sy"][k]])
        direction = np.array([tab["dirx"][k], tab["diry"][k]])
        return float(np.dot(delta, direction)), float(np.dot(delta, np.array([-direction[1], direction[0]])))
    delta = np.asarray(pos) - np.array([tab["cx"][k], tab["cy"][k]])   # CircularLane.local_coordinates (:355-362)
    phi = np.arctan2(delta[1], delta[0])
    phi = tab["start_phase"][k] + ((phi - tab["start_phase"][k] + np.pi) % (2 * np.pi) - np.pi)
    r = np.linalg.norm(delta)
    return float(tab["direction"][k] * (phi - tab["start_phase"][k]) * tab["radius"][k]), float(
        tab["direction"][k] * (tab["radius"][k] - r))


def closest_lane(tab: dict, pos, heading: float) -> int:
    """RoadNetwork.get_closest_lane_index (road.py:55-71) with distance_with_heading (lane.py:132-147)."""
    best, bd = 0, None
    for k in range(len(tab["kind"])):
        s, r = lane_local(tab, k, pos)
        angle = abs(((heading - lane_heading_at(tab, k, s)) + np.pi) % (2 * np.pi) - np.pi)
        d = abs(r) + max(s - tab["length"][k], 0) + max(0 - s, 0) + 1.0 * angle
        if bd is None or d < bd:
            best, bd = k, d
    return best


# --------------------------------------------------------------------------- traffic management on numpy's stream
def _n_present(st: dict, e: int) -> int:
    return int(((st["flags"][e] & _abi.F_ABSENT) == 0).sum())


def _put_idm(c, tab, st, e, i, pos, heading, speed, dest, delta):
    st["x"][e, i], st["y"][e, i], st["heading"][e, i], st["speed"][e, i] = pos[0], pos[1], heading, speed
    lane = closest_lane(tab, pos, heading)
    st["lane"][e, i] = st["target_lane"][e, i] = lane
    st["target_speed"][e, i] = speed
    st["timer"][e, i] = (np.sum(pos) * np.pi) % 1.0          # IDMVehicle ctor (behavior.py:64)
    st["delta"][e, i] = delta
    st["impact_x"][e, i] = st["impact_y"][e, i] = 0.0
    st["speed_index"][e, i] = 0
    st["flags"][e, i] = _abi.F_CHECK_COLLISIONS
    st["route"][e, i] = route_pack(plan_route(tab, lane, dest))


def spawn_vehicle(c, tab, st, e, rng, longitudinal=0.0, position_deviation=1.0, speed_deviation=1.0,
                  spawn_probability=0.6, go_straight=False) -> bool:
    """IntersectionEnv._spawn_vehicle (intersection_env.py:292-324) in the reference's draw order."""
    if rng.uniform() > spawn_probability:
        return False
    route = rng.choice(range(4), size=2, replace=False)
    route[1] = (route[0] + 2) % 4 if go_straight else route[1]
    access = lane_index_of(tab, f"o{route[0]}", f"ir{route[0]}")
    lon = longitudinal + 5.0 + rng.normal() * position_deviation
    speed = 8.0 + rng.normal() * speed_deviation
    pos = lane_position(tab, access, lon)
    heading = lane_heading_at(tab, access, lon)
    present = (st["flags"][e] & _abi.F_ABSENT) == 0
    for i in np.nonzero(present)[0]:
        if np.linalg.norm(np.array([st["x"][e, i], st["y"][e, i]]) - pos) < 15:
            return False
    n = int(present.sum())
    if n >= c.num_vehicles:
        raise RuntimeError("max_vehicles slots exhausted: raise config['max_vehicles']")
    delta = rng.uniform(low=3.5, high=4.5)                   # randomize_behavior (behavior.py:66-69)
    _put_idm(c, tab, st, e, n, pos, heading, speed, int(route[1]), delta)
    return True


def _compact(st: dict, e: int, keep: np.ndarray) -> None:
    order = np.concatenate([np.nonzero(keep)[0], np.nonzero(~keep)[0]])
    for k, a in st.items():
        if a.ndim == 2:
            a[e] = a[e][order]
    st["flags"][e, int(keep.sum()):] = _abi.F_ABSENT


def clear_vehicles(c, tab, st, e) -> None:
    """IntersectionEnv._clear_vehicles (intersection_env.py:326-338)."""
    present = (st["flags"][e] & _abi.F_ABSENT) == 0
    keep = present.copy()
    for i in np.nonzero(present)[0]:
        lane = int(st["lane"][e, i])
        s, _ = lane_local(tab, lane, (st["x"][e, i], st["y"][e, i]))
        leaving = bool(tab["exit_lane"][lane]) and s >= tab["length"][lane] - 4 * VEH_LENGTH
        if leaving and not (st["flags"][e, i] & _abi.F_CONTROLLED):
            keep[i] = False
    if not (keep == present).all():
        _compact(st, e, keep)


def make_vehicles_before_warmup(c, cfg, tab, st, e, rng) -> None:
    """First half of IntersectionEnv._make_vehicles (:249-251): the initial random traffic."""
    n = int(cfg["initial_vehicle_count"])
    for t in range(n - 1):
        spawn_vehicle(c, tab, st, e, rng, np.linspace(0, 80, n)[t])


def make_vehicles_after_warmup(c, cfg, tab, st, e, rng) -> None:
    """Second half (:260-318): challenger, the controlled vehicles, removal of the traffic within 20 m of each."""
    spawn_vehicle(c, tab, st, e, rng, 60, spawn_probability=1.0, go_straight=True, position_deviation=0.1,
                  speed_deviation=0.0)
    ts = np.array([c.target_speeds[k] for k in range(c.num_target_speeds)])
    for ego_id in range(c.num_agents):  # (:292-318)
        access = lane_index_of(tab, f"o{ego_id % 4}", f"ir{ego_id % 4}")
        # destination = config["destination"] or "o" + str(np_random.integers(1, 4)): drawn BEFORE the position (:295-300)
        destination = c.destination if c.destination >= 0 else int(rng.integers(1, 4))
        pos = lane_position(tab, access, 60.0 + 5.0 * rng.normal(1.0))
        heading = lane_heading_at(tab, access, 60.0)
        speed = float(tab["speed_limit"][access])
        i = _n_present(st, e)
        if i >= c.num_vehicles:
            raise RuntimeError("max_vehicles slots exhausted: raise config['max_vehicles']")
        lane = closest_lane(tab, pos, heading)
        st["x"][e, i], st["y"][e, i], st["heading"][e, i], st["speed"][e, i] = pos[0], pos[1], heading, speed
        st["lane"][e, i] = st["target_lane"][e, i] = lane
        xs = (speed - ts[0]) / (ts[-1] - ts[0])
        sidx = int(np.clip(np.round(xs * (ts.size - 1)), 0, ts.size - 1))   # speed_to_index (controller.py:326-344)
        st["speed_index"][e, i] = sidx
        st["target_speed"][e, i] = ts[sidx]
        st["timer"][e, i] = st["delta"][e, i] = st["impact_x"][e, i] = st["impact_y"][e, i] = 0.0
        st["flags"][e, i] = _abi.F_CONTROLLED | _abi.F_CHECK_COLLISIONS
        st["route"][e, i] = route_pack(plan_route(tab, lane, destination))
        present = (st["flags"][e] & _abi.F_ABSENT) == 0
        keep = present.copy()
        for j in np.nonzero(present)[0]:  # "prevent early collisions": the other controlled vehicles stay (:313-318)
            if not (st["flags"][e, j] & _abi.F_CONTROLLED) and np.linalg.norm(np.array([st["x"][e, j], st["y"][e, j]]) - pos) < 20:
                keep[j] = False
        if not (keep == present).all():
            _compact(st, e, keep)


def reset_reference_stream(eng, c, cfg: dict, generators) -> dict:
    """``reset(seed=s)`` of IntersectionEnv on the env's numpy Generators: spawns on the host in the reference's draw
    order, the three simulated seconds in between (intersection_env.py:252-258) on the engine."""
    tab = table_from_config(c)
    E = c.num_envs
    st = _abi.alloc_state_ix(E, c.num_vehicles)
    for e in range(E):
        make_vehicles_before_warmup(c, cfg, tab, st, e, generators[e])
    eng.set_state(st)
    eng.step_frames(None, 3 * int(cfg["simulation_frequency"]))
    st = eng.get_state()
    for e in range(E):
        make_vehicles_after_warmup(c, cfg, tab, st, e, generators[e])
    st["time"][...] = 0.0
    eng.set_state(st)
    return st


def clear_and_spawn_reference_stream(eng, c, cfg: dict, generators) -> None:
    """The tail of IntersectionEnv.step (intersection_env.py:136-140) on the env's numpy Generators."""
    tab = table_from_config(c)
    st = eng.get_state()
    for e in range(c.num_envs):
        clear_vehicles(c, tab, st, e)
        spawn_vehicle(c, tab, st, e, generators[e], spawn_probability=float(cfg["spawn_probability"]))
    eng.set_state(st)
