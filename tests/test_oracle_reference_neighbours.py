"""The reference's own unit tests of Road.neighbour_vehicles, /root/reference/tests/road/test_neighbour_vehicles.py
(SURVEY.md section 8c), restated case by case against the C restatements the parity tests use as their checker:
`oracle/hwy_oracle_net.c` (x-aligned lane tables) for the straight fixtures and `oracle/hwy_oracle_ix.c` (general lanes)
for the straight + CircularLane fixture.  Same fixtures (segment lengths, longitudinal positions), same assertions:
the IDENTITY of the front / rear vehicle, or None.  CPU only; the oracle is test infrastructure.
"""
import ctypes as C

import numpy as np
import pytest

from highwayenv_amd import _abi, merge
from oracle import oracle, oracle_ix


# ---- fixtures (test_neighbour_vehicles.py:40-152) as lane tables ---------------------------------------------------------
def _net_config(roads, connected, n_slots=4):
    """roads: [(name, x0, length, [y of every lane]), ...] in add_lane order; successor = the road starting where it ends."""
    rows = [(name, x0, y, length) for name, x0, length, ys in roads for y in ys]
    cfg = merge.merge_default_config()
    cfg["neighbour_vehicles_connected_lanes"] = connected
    c = _abi.make_config(cfg, 1, scenario="merge")
    names = [r[0] for r in rows]
    nxt = {name: next((m for m, x1, _, _ in roads if x1 == x0 + length), None) for name, x0, length, _ in roads}
    c.net_lanes = len(rows)
    tab = {k: np.zeros(len(rows)) for k in _abi.LANE_F64 + _abi.LANE_I32}
    for k, (name, x0, y, length) in enumerate(rows):
        first = names.index(name)
        vals = dict(x0=x0, y0=y, length=length, width=4.0, amplitude=0.0, pulsation=0.0, phase=0.0, speed_limit=20.0,
                    road=list(dict.fromkeys(names)).index(name), id=k - first, road_first=first, road_lanes=names.count(name),
                    next_first=names.index(nxt[name]) if nxt[name] else -1, next_lanes=names.count(nxt[name]) if nxt[name] else 0,
                    forbidden=0)
        for f, v in vals.items():
            tab[f][k] = v
            setattr(c.net[k], f, type(getattr(c.net[k], f))(v))
    for k, m in enumerate(merge.connected_masks({f: tab[f].astype(int) if f in _abi.LANE_I32 else tab[f] for f in tab})):
        c.net[k].connected = m
    c.num_vehicles = n_slots
    index = {(name, i): names.index(name) + i for name in dict.fromkeys(names) for i in range(names.count(name))}
    return c, index, {k: (rows[k][1], rows[k][2]) for k in range(len(rows))}


class NetRoad:
    """Road + `_make_vehicle(road, net, lane_index, longitudinal)` of the reference test (positions on the lane centre)."""

    def __init__(self, roads, connected):
        self.cfg, self.index, self.origin = _net_config(roads, connected)
        self.st = _abi.alloc_state(1, self.cfg.num_vehicles)
        self.st["flags"][...] = _abi.F_ABSENT
        self.n = 0

    def make_vehicle(self, road, lane_id, longitudinal):
        k = self.index[(road, lane_id)]
        x0, y0 = self.origin[k]
        i = self.n
        self.st["x"][0, i], self.st["y"][0, i], self.st["speed"][0, i] = x0 + longitudinal, y0, 10.0
        self.st["lane"][0, i] = self.st["target_lane"][0, i] = k
        self.st["flags"][0, i] = _abi.F_CHECK_COLLISIONS
        self.n += 1
        return i

    def neighbour_vehicles(self, vehicle, road, lane_id):
        return oracle.net_neighbours(self.cfg, self.st, 0, vehicle, self.index[(road, lane_id)])


AB, BC, CD = ("ab", 0.0, 50.0, [0.0]), ("bc", 50.0, 50.0, [0.0]), ("cd", 100.0, 50.0, [0.0])
straight_connected_road = [AB, BC]
three_segment_road = [AB, BC, CD]
multi_lane_road = [("ab", 0.0, 50.0, [0.0, 4.0]), ("bc", 50.0, 50.0, [0.0, 4.0])]


# ---- TestSameSegmentNeighbours (:159-206) ----------------------------------------------------------------------------------
def test_front_and_rear_on_same_segment():
    r = NetRoad(straight_connected_road, False)
    ego, front, rear = r.make_vehicle("ab", 0, 25), r.make_vehicle("ab", 0, 40), r.make_vehicle("ab", 0, 10)
    assert r.neighbour_vehicles(ego, "ab", 0) == (front, rear)


def test_no_neighbours():
    r = NetRoad(straight_connected_road, False)
    ego = r.make_vehicle("ab", 0, 25)
    assert r.neighbour_vehicles(ego, "ab", 0) == (None, None)


def test_only_front():
    r = NetRoad(straight_connected_road, False)
    ego, front = r.make_vehicle("ab", 0, 10), r.make_vehicle("ab", 0, 40)
    assert r.neighbour_vehicles(ego, "ab", 0) == (front, None)


def test_only_rear():
    r = NetRoad(straight_connected_road, False)
    ego, rear = r.make_vehicle("ab", 0, 40), r.make_vehicle("ab", 0, 10)
    assert r.neighbour_vehicles(ego, "ab", 0) == (None, rear)


def test_connected_segments_ignored_by_default():
    r = NetRoad(straight_connected_road, False)
    ego = r.make_vehicle("ab", 0, 48)
    r.make_vehicle("bc", 0, 5)
    # (the reference asserts (None, None): 5 m into b->c is s = 55 on a->b, exactly `length + VEHICLE_LENGTH`, and
    #  on_lane's upper bound is strict)
    assert r.neighbour_vehicles(ego, "ab", 0) == (None, None)


# ---- TestConnectedLaneNeighbours (:213-296) --------------------------------------------------------------------------------
def test_front_on_next_segment():
    r = NetRoad(straight_connected_road, True)
    ego, front = r.make_vehicle("ab", 0, 48), r.make_vehicle("bc", 0, 5)
    assert r.neighbour_vehicles(ego, "ab", 0)[0] == front


def test_rear_on_previous_segment():
    r = NetRoad(straight_connected_road, True)
    ego, rear = r.make_vehicle("bc", 0, 5), r.make_vehicle("ab", 0, 45)
    assert r.neighbour_vehicles(ego, "bc", 0)[1] == rear


def test_closer_same_segment_preferred_over_next_segment():
    r = NetRoad(straight_connected_road, True)
    ego, close_front = r.make_vehicle("ab", 0, 30), r.make_vehicle("ab", 0, 45)
    r.make_vehicle("bc", 0, 10)  # farther vehicle on the next segment
    assert r.neighbour_vehicles(ego, "ab", 0)[0] == close_front


def test_both_connected_front_and_rear():
    r = NetRoad(three_segment_road, True)
    rear, ego, front = r.make_vehicle("ab", 0, 45), r.make_vehicle("bc", 0, 5), r.make_vehicle("cd", 0, 5)
    assert r.neighbour_vehicles(ego, "bc", 0) == (front, rear)


def test_multi_lane_same_lane_id():
    r = NetRoad(multi_lane_road, True)
    ego, front_lane0 = r.make_vehicle("ab", 0, 48), r.make_vehicle("bc", 0, 5)
    r.make_vehicle("bc", 1, 3)  # lane 1 of the next segment: a different lane
    assert r.neighbour_vehicles(ego, "ab", 0)[0] == front_lane0


# ---- TestEdgeCases (:303-366) ----------------------------------------------------------------------------------------------
def test_no_next_segment():
    r = NetRoad([AB], True)
    ego = r.make_vehicle("ab", 0, 48)
    assert r.neighbour_vehicles(ego, "ab", 0) == (None, None)


def test_no_previous_segment():
    r = NetRoad([BC], True)
    ego = r.make_vehicle("bc", 0, 5)
    assert r.neighbour_vehicles(ego, "bc", 0) == (None, None)


def test_vehicle_far_on_next_segment_detected():
    r = NetRoad(straight_connected_road, True)
    ego = r.make_vehicle("ab", 0, 25)
    far = r.make_vehicle("bc", 0, 40)
    assert r.neighbour_vehicles(ego, "ab", 0)[0] == far


# ---- test_front_on_curve_segment (:237-247): StraightLane a->b followed by a CircularLane b->c ------------------------------
def test_lane_index_none_returns_none():
    """test_neighbour_vehicles.py:358-366: a vehicle without a lane index has no neighbours (road.py:499-501); with
    lane_index=None and a lane index the vehicle's own lane is searched."""
    r = NetRoad(straight_connected_road, False)
    ego, front = r.make_vehicle("ab", 0, 25), r.make_vehicle("ab", 0, 40)
    assert oracle.net_neighbours(r.cfg, r.st, 0, ego, None) == (front, None)
    r.st["lane"][0, ego] = -1  # ego.lane_index = None
    assert oracle.net_neighbours(r.cfg, r.st, 0, ego, None) == (None, None)


def test_front_on_curve_segment():
    c = oracle_ix.IxConfig()
    c.num_envs, c.n_slots, c.n_lanes, c.n_route, c.connected_lanes = 1, 4, 2, 4, 1
    a, b = c.lanes[0], c.lanes[1]
    a.kind, a.from_node, a.to_node, a.id = 0, 0, 1, 0
    a.sx, a.sy, a.ex, a.ey, a.heading, a.dirx, a.diry, a.length, a.width, a.speed_limit = 0, 0, 50, 0, 0, 1, 0, 50, 4, 20
    # CircularLane(center=[50, -20], radius=20, start_phase=90 deg, end_phase=0, clockwise=False) (lane.py:311-339)
    b.kind, b.direction, b.from_node, b.to_node, b.id = 1, -1, 1, 2, 0
    b.cx, b.cy, b.radius, b.start_phase, b.end_phase = 50, -20, 20, np.deg2rad(90), np.deg2rad(0)
    b.length, b.width, b.speed_limit = 20 * (np.deg2rad(0) - np.deg2rad(90)) * -1, 4, 20
    st = oracle_ix.alloc_state(1, 4)
    # ego 48 m along a->b; the other car 5 m along the arc: position(5, 0) (lane.py:341-345)
    phi = -1 * 5 / 20 + np.deg2rad(90)
    for i, (x, y, lane) in enumerate([(48.0, 0.0, 0), (50 + 20 * np.cos(phi), -20 + 20 * np.sin(phi), 1)]):
        st["present"][0, i], st["x"][0, i], st["y"][0, i], st["speed"][0, i] = 1, x, y, 10.0
        st["lane"][0, i] = st["target_lane"][0, i] = lane
    assert oracle_ix.neighbours(c, st, 0, 0, 0)[0] == 1


def test_rotated_rectangles_intersect():
    """/root/reference/tests/test_utils.py:19-28: the five known-answer vectors of utils.rotated_rectangles_intersect, the
    predicate RegulatedRoad.is_conflict_possible rests on (regulation.py:103-108)."""
    lib = oracle.lib()
    lib.orc_ix_rotated_rectangles_intersect.argtypes = [C.c_double] * 10
    lib.orc_ix_rotated_rectangles_intersect.restype = C.c_int

    def rri(r1, r2):
        (c1, l1, w1, a1), (c2, l2, w2, a2) = r1, r2
        return bool(lib.orc_ix_rotated_rectangles_intersect(c1[0], c1[1], l1, w1, a1, c2[0], c2[1], l2, w2, a2))

    assert rri(([12.86076812, 28.60182391], 5.0, 2.0, -0.4675779906495494), ([9.67753944, 28.90585412], 5.0, 2.0, -0.3417019364473201))
    assert rri(([0, 0], 2, 1, 0), ([0, 1], 2, 1, 0))
    assert not rri(([0, 0], 2, 1, 0), ([0, 2.1], 2, 1, 0))
    assert not rri(([0, 0], 2, 1, 0), ([1, 1.1], 2, 1, 0))
    assert rri(([0, 0], 2, 1, np.pi / 4), ([1, 1.1], 2, 1, 0))
