"""Merge scenarios: config dicts, the lane table of the road network, and the host-side,
stream-identical reset.

Mirrors ``highway_env/envs/merge_env.py``: ``MergeEnv`` (merge-v0, :15-190) and ``MergeGenericEnv``
(merge-generic-v0, :193-371).  The road network is a handful of x-aligned lanes (straight highway
sections a->b->c->d, an access ramp j->k, a ``SineLane`` k->b converging onto the highway and a
parallel acceleration lane on b->c ending in an ``Obstacle``); it is flattened into
``hwy_config.net`` (``include/hwy_engine.h``: ``hwy_lane``) in the iteration order of
``RoadNetwork.get_closest_lane_index`` (road/road.py:55-71).

Slot layout of the vehicle arrays (N = ``hwy_config.num_vehicles`` slots per environment):

    merge-v0          [ego, 3 x IDM, merging IDM, obstacle]                                  N = 6
    merge-generic-v0  [ego, vehicles_count x IDM (HWY_F_ABSENT where the rejection-sampled
                       spawn gave up, merge_env.py:336-352), merging IDM, obstacle]          N = vehicles_count + 3

``controlled_vehicles`` > 1 (BASELINE config 5) is an extension the reference's merge classes cannot
express (``_make_vehicles`` creates one ego): the first A-1 traffic slots are then MDPVehicles at the
position / speed the traffic vehicle would have had.
"""
from __future__ import annotations

import numpy as np

from . import _abi

AMPLITUDE = 3.25  # merge_env.py:128,266


# --------------------------------------------------------------------------- config dicts
def merge_default_config() -> dict:
    """MergeEnv.default_config (merge_env.py:24-38)."""
    cfg = _abi.abstract_default_config()
    cfg.update({
        "collision_reward": -1,
        "right_lane_reward": 0.1,
        "high_speed_reward": 0.2,
        "reward_speed_range": [20, 30],
        "merging_speed_reward": -0.5,
        "lane_change_reward": -0.05,
    })
    return cfg


def merge_generic_default_config() -> dict:
    """MergeGenericEnv.default_config (merge_env.py:215-233)."""
    cfg = merge_default_config()
    cfg.update({
        "lanes_count": 2,
        "vehicles_count": 3,
        "before_merge_length": 150,
        "converge_merge_length": 80,
        "parallel_merge_length": 80,
        "after_merge_length": 150,
    })
    return cfg


def _sections(cfg: dict, generic: bool):
    if generic:
        pre, conv, par, after = (cfg["before_merge_length"], cfg["converge_merge_length"],
                                 cfg["parallel_merge_length"], cfg["after_merge_length"])
        # the reference's own asserts (merge_env.py:241-244)
        assert all(s > 0 for s in (pre, conv, par)), "All road segments must have positive length"
        assert after >= 90, "The after merge road segment must have length >= 90"
        return pre, conv, par, after, int(cfg["lanes_count"])
    return 150, 80, 80, 150, 2  # ends = [150, 80, 80, 150], two highway lanes (merge_env.py:98-101)


def ego_road_lanes(cfg: dict, generic: bool) -> int:
    """len(all_side_lanes(ego.lane_index)) at reset: the ego starts on road a->b."""
    return _sections(cfg, generic)[4]


def lane_table(cfg: dict, generic: bool) -> dict:
    """The network as arrays (keys: _abi.LANE_F64 + _abi.LANE_I32), in get_closest_lane_index order."""
    pre, conv, par, after, lanes = _sections(cfg, generic)
    limit = 30.0 if generic else 20.0  # generic passes speed_limit=30; MergeEnv keeps StraightLane's default 20
    w = 4.0
    rows = []  # (road name, x0, y0, length, amplitude, pulsation, phase, forbidden)
    y_par = lanes * w
    sections = [("ab", 0.0, float(pre + conv)), ("bc", float(pre + conv), float(par)),
                ("cd", float(pre + conv + par), float(after))]
    for name, start, length in sections:
        for i in range(lanes):
            rows.append((name, start, i * w, length, 0.0, 0.0, 0.0, 0))
        if name == "bc":  # net.add_lane("b", "c", lbc) appends the acceleration lane as id `lanes`
            if generic:
                y_lbc = float(y_par)
            else:  # lkb.position(ends[1], 0): 6.5 + 4 + 4 - amplitude + amplitude*sin(pulsation*80 + pi/2)
                y_lbc = float((6.5 + 4 + 4) + -AMPLITUDE
                              + AMPLITUDE * np.sin(2 * np.pi / (2 * conv) * conv + np.pi / 2))
            rows.append((name, start, y_lbc, length, 0.0, 0.0, 0.0, 1))
    y_approach = (y_par + 2 * AMPLITUDE) if generic else (6.5 + 4 + 4)
    rows.append(("jk", 0.0, float(y_approach), float(pre), 0.0, 0.0, 0.0, 1))
    y_sine = (y_par + AMPLITUDE) if generic else (y_approach + -AMPLITUDE)
    rows.append(("kb", float(pre), float(y_sine), float(conv), AMPLITUDE, 2 * np.pi / (2 * conv), np.pi / 2, 1))
    successor = {"ab": "bc", "bc": "cd", "cd": None, "jk": "kb", "kb": "bc"}
    names = [r[0] for r in rows]
    road_ids = {n: k for k, n in enumerate(dict.fromkeys(names))}
    tab = {k: np.zeros(len(rows), np.float64) for k in _abi.LANE_F64}
    tab.update({k: np.zeros(len(rows), np.int32) for k in _abi.LANE_I32})
    for k, (name, x0, y0, length, amp, puls, phase, forb) in enumerate(rows):
        tab["x0"][k], tab["y0"][k], tab["length"][k], tab["width"][k] = x0, y0, length, w
        tab["amplitude"][k], tab["pulsation"][k], tab["phase"][k] = amp, puls, phase
        tab["speed_limit"][k] = limit
        tab["forbidden"][k] = forb
        tab["road"][k] = road_ids[name]
        tab["road_first"][k] = names.index(name)
        tab["id"][k] = k - names.index(name)
        tab["road_lanes"][k] = names.count(name)
        nxt = successor[name]
        tab["next_first"][k] = names.index(nxt) if nxt else -1
        tab["next_lanes"][k] = names.count(nxt) if nxt else 0
    return tab


def connected_masks(tab: dict) -> list:
    """hwy_lane.connected: the lanes Road.neighbour_vehicles searches together with lane k when
    neighbour_vehicles_connected_lanes is on (road.py:508-529): lane k, lane `id` (else 0) of the successor road, lane `id`
    (else 0) of every road that ends where k's road starts."""
    n = len(tab["x0"])
    out = []
    for k in range(n):
        m = 1 << k
        _id = int(tab["id"][k])
        if tab["next_first"][k] >= 0:
            m |= 1 << (int(tab["next_first"][k]) + (_id if _id < tab["next_lanes"][k] else 0))
        for q in range(n):   # one visit per road (its lane 0) whose successor is k's road
            if tab["id"][q] == 0 and tab["next_first"][q] == tab["road_first"][k]:
                m |= 1 << (q + (_id if _id < tab["road_lanes"][q] else 0))
        out.append(m)
    return out


def table_from_config(c: _abi.HwyConfig) -> dict:
    n = c.net_lanes
    tab = {k: np.array([getattr(c.net[i], k) for i in range(n)], np.float64) for k in _abi.LANE_F64}
    tab.update({k: np.array([getattr(c.net[i], k) for i in range(n)], np.int32) for k in _abi.LANE_I32})
    return tab


def fill_config(c: _abi.HwyConfig, cfg: dict, generic: bool) -> None:
    """The merge-specific part of _abi.make_config."""
    pre, conv, par, after, lanes = _sections(cfg, generic)
    if not (1 <= lanes and 3 * lanes + 3 <= _abi.HWY_MAX_LANES):
        raise ValueError(f"lanes_count must be in [1, {(_abi.HWY_MAX_LANES - 3) // 3}] for the merge scenarios")
    A = c.num_agents
    n_traffic = int(cfg["vehicles_count"]) if generic else 3
    if A - 1 > n_traffic:
        raise ValueError("controlled_vehicles - 1 must not exceed the number of traffic vehicles")
    c.scenario = _abi.SCENARIO_MERGE_GENERIC if generic else _abi.SCENARIO_MERGE
    c.num_vehicles = 1 + n_traffic + 1 + 1  # ego, traffic, merging vehicle, obstacle
    if c.num_vehicles > 64:
        raise ValueError("the merge scenarios run one wavefront per environment: at most 61 traffic vehicles")
    for a in range(A):
        c.agent_index[a] = a
    c.lanes_count = lanes
    c.duration = float("inf")   # MergeEnv._is_truncated is always False (merge_env.py:81-82)
    c.road_length = float(pre + conv + par + after)
    c.speed_limit = 30.0 if generic else 20.0
    tab = lane_table(cfg, generic)
    c.net_lanes = len(tab["x0"])
    for k in range(c.net_lanes):
        for f in _abi.LANE_F64:
            setattr(c.net[k], f, float(tab[f][k]))
        for f in _abi.LANE_I32:
            setattr(c.net[k], f, int(tab[f][k]))
    for k, m in enumerate(connected_masks(tab)):
        c.net[k].connected = m
    # MergeEnv._rewards tests `vehicle.lane_index == ("b", "c", 2)` literally (merge_env.py:72): the acceleration
    # lane when lanes_count == 2, a HIGHWAY lane of b->c when MergeGenericEnv has more lanes.  a->b holds `lanes`
    # entries and b->c starts right after them.
    c.merge_lane = lanes + 2 if lanes >= 2 else -1
    c.merge_end_x = float(pre + conv + par + after - 90) if generic else 370.0
    c.merging_speed_reward = float(cfg["merging_speed_reward"])
    c.lane_change_reward = float(cfg["lane_change_reward"])


# --------------------------------------------------------------------------- lane geometry on the host
def closest_lane(tab: dict, x: float, y: float, heading: float) -> int:
    """RoadNetwork.get_closest_lane_index (road.py:55-71) with distance_with_heading (lane.py:132-143)."""
    best, bd = 0, None
    for k in range(len(tab["x0"])):
        s = x - tab["x0"][k]
        r = y - tab["y0"][k]
        h_lane = 0.0
        if tab["amplitude"][k] != 0:
            arg = tab["pulsation"][k] * s + tab["phase"][k]
            r = r - tab["amplitude"][k] * np.sin(arg)
            h_lane = 0.0 + np.arctan(tab["amplitude"][k] * tab["pulsation"][k] * np.cos(arg))
        angle = abs(((heading - h_lane) + np.pi) % (2 * np.pi) - np.pi)
        d = abs(r) + max(s - tab["length"][k], 0) + max(0 - s, 0) + 1.0 * angle
        if bd is None or d < bd:
            best, bd = k, d
    return best


# --------------------------------------------------------------------------- reset on numpy's stream
def _new_state(c: _abi.HwyConfig, E: int) -> dict:
    st = _abi.alloc_state(E, c.num_vehicles)
    st["flags"][...] = _abi.F_ABSENT
    return st


def _put_vehicle(c, tab, st, e, i, x, y, speed, controlled, target_speed=None):
    ts = np.array([c.target_speeds[k] for k in range(c.num_target_speeds)])
    st["x"][e, i], st["y"][e, i], st["heading"][e, i], st["speed"][e, i] = x, y, 0.0, speed
    lane = closest_lane(tab, x, y, 0.0)
    st["lane"][e, i] = st["target_lane"][e, i] = lane
    if controlled:  # MDPVehicle ladder snap (controller.py:287-293, 326-344)
        xs = (speed - ts[0]) / (ts[-1] - ts[0])
        sidx = int(np.clip(np.round(xs * (ts.size - 1)), 0, ts.size - 1))
        st["speed_index"][e, i] = sidx
        st["target_speed"][e, i] = ts[sidx]
        st["flags"][e, i] = _abi.F_CONTROLLED | _abi.F_CHECK_COLLISIONS
    else:           # IDMVehicle ctor (behavior.py:46-64): DELTA stays the class default 4.0
        st["target_speed"][e, i] = speed if target_speed is None else target_speed
        st["timer"][e, i] = ((x + y) * np.pi) % 1.0
        st["delta"][e, i] = 4.0
        st["flags"][e, i] = _abi.F_CHECK_COLLISIONS


def _put_obstacle(c, tab, st, e, x, y):
    i = c.num_vehicles - 1
    st["x"][e, i], st["y"][e, i] = x, y
    st["lane"][e, i] = st["target_lane"][e, i] = closest_lane(tab, x, y, 0.0)
    st["flags"][e, i] = _abi.F_OBSTACLE | _abi.F_CHECK_COLLISIONS


def spawn_reference_stream(c: _abi.HwyConfig, cfg: dict, generic: bool, seeds) -> dict:
    """``reset(seed=s)`` of MergeEnv / MergeGenericEnv replayed on ``np.random.default_rng(s)``:
    ``_make_vehicles`` (merge_env.py:162-187 / :320-363) in the reference's draw order."""
    pre, conv, par, after, lanes = _sections(cfg, generic)
    tab = table_from_config(c)
    E, N, A = len(seeds), c.num_vehicles, c.num_agents
    st = _new_state(c, E)
    w = 4.0
    for e, seed in enumerate(seeds):
        rng = seed if isinstance(seed, np.random.Generator) else np.random.default_rng(int(seed))
        _put_vehicle(c, tab, st, e, 0, 30.0, (lanes - 1) * w, 30.0, True)  # ego on ("a","b",lanes-1) at s=30
        slot = 1
        if not generic:
            for position, speed in [(90.0, 29.0), (70.0, 31.0), (5.0, 31.5)]:
                lane = int(rng.integers(2))
                x = position + rng.uniform(-5.0, 5.0)
                speed = speed + rng.uniform(-1.0, 1.0)
                _put_vehicle(c, tab, st, e, slot, x, lane * w, speed, slot < A)
                slot += 1
            merging = (110.0, 6.5 + 4 + 4)
        else:
            max_pos = pre + conv + par
            spawned = {i: [] for i in range(lanes)}
            spawned[lanes - 1].append(30.0)
            for k in range(int(cfg["vehicles_count"])):
                for _ in range(10):
                    lane = int(rng.integers(lanes))
                    longitudinal = rng.uniform(0, max_pos)
                    if all(abs(longitudinal - p) > 15.0 for p in spawned[lane]):
                        spd = 30.0 + rng.uniform(-2.0, 2.0)
                        _put_vehicle(c, tab, st, e, slot, longitudinal, lane * w, spd, slot < A)
                        spawned[lane].append(longitudinal)
                        slot += 1
                        break
            # vehicles are appended compactly in the reference; the unused slots stay HWY_F_ABSENT
            merging = (30.0 + 30, lanes * w + 2 * AMPLITUDE)
        _put_vehicle(c, tab, st, e, N - 2, merging[0], merging[1], 20.0, False, target_speed=30.0)
        _put_obstacle(c, tab, st, e, float(pre + conv + par), float(tab["y0"][2 * lanes]))  # end of ("b","c",lanes)
    return st
