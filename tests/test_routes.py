"""Planned routes of the intersection scenario: the 64-bit route word, and the host-side breadth-first planner
(highwayenv_amd/intersection.py: shortest_path / plan_route / route_table) that fills hwy_config.gnet_routes -- the kernel
only looks routes up, so any route RoadNetwork.shortest_path (road.py:159-188) can produce on this network is supported."""
import numpy as np
import pytest

from highwayenv_amd import _abi
from highwayenv_amd import intersection as hix
from oracle import ref_stub


def _table():
    c = _abi.make_config(hix.intersection_default_config(), 1, scenario="intersection")
    return c, hix.table_from_config(c)


def test_route_word_round_trip():
    for lanes in ([], [3], [0, 7, 19], list(range(11)), [31] * 11):
        assert hix.route_unpack(hix.route_pack(lanes)) == lanes
    with pytest.raises(AssertionError):
        hix.route_pack(list(range(12)))
    assert hix.route_pack([1, 2]) == (2 << 56) | 1 | (2 << 5)


def test_route_table_of_the_default_network():
    c, tab = _table()
    names = hix.NODE_NAMES
    for q in range(4):  # from an access lane ("o" + q, "ir" + q) every OTHER exit is two roads away ...
        access = hix.lane_index_of(tab, f"o{q}", f"ir{q}")
        for k in range(4):
            roads = hix.plan_route(tab, access, k)
            assert roads[0] == access and names[tab["to_node"][roads[-1]]] == f"o{k}"
            for a, b in zip(roads[:-1], roads[1:]):  # consecutive roads are connected
                assert tab["to_node"][a] == tab["from_node"][b]
            assert len(roads) == (3 if k != q else 6)  # ... and its own exit needs a loop through another arm (6 roads)
            assert hix.route_unpack(int(c.gnet_routes[access][k])) == roads[1:]


@pytest.mark.reference
@pytest.mark.skipif(not ref_stub.reference_available(), reason="needs the reference package (build container)")
def test_planner_equals_the_reference_shortest_path():
    ref_stub.install()
    from highway_env.envs.intersection_env import IntersectionEnv
    env = IntersectionEnv()
    env.reset(seed=0)
    net = env.road.network
    _, tab = _table()
    for lane in range(len(tab["kind"])):
        t = hix.NODE_NAMES[tab["to_node"][lane]]
        for k in range(4):
            ref = net.shortest_path(t, f"o{k}")
            mine = hix.plan_route(tab, lane, k)
            nodes = [t] + [hix.NODE_NAMES[tab["to_node"][l]] for l in mine[1:]]
            assert (ref or [t]) == nodes, (lane, k, ref, nodes)
