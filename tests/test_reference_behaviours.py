"""The reference's own behavioural tests for this path, restated against the product through
the C-ABI (``emu`` = the kernel source on the CPU, ``hip`` = libhwy_engine.so on the MI355X).  They are loose on
purpose -- the reference's assertions are (``pytest.approx`` on positions and speeds after a few simulated seconds);
the bit-level pins are the golden-trace tests.

    /root/reference/tests/vehicle/test_dynamics.py:11-18, 66-73     a free vehicle keeps its speed; two overlapping cars crash
    /root/reference/tests/vehicle/test_control.py:21-54             LANE_RIGHT reaches the next lane; FASTER raises the speed
    /root/reference/tests/vehicle/test_behavior.py:11-27            an IDM vehicle stops DISTANCE_WANTED short of an obstacle
    /root/reference/tests/envs/test_gym.py:65-91                    whole episodes: observations stay inside the space
"""
import numpy as np
import pytest

from highwayenv_amd import _abi
from tests.backends import BACKENDS, make_engine

FPS = 15
pytestmark = pytest.mark.parametrize("backend", BACKENDS)


def _highway(E, vehicles_count, lanes, controlled=1):
    cfg = _abi.highway_default_config()   # simulation 15 Hz, policy 1 Hz like the reference's FPS = 15 loops
    cfg.update({"vehicles_count": vehicles_count, "lanes_count": lanes, "controlled_vehicles": controlled, "duration": 100})
    return cfg, _abi.make_config(cfg, E)


def _one_controlled(st, e, i, x, y, speed, lane, target_speed, speed_index):
    st["x"][e, i], st["y"][e, i], st["speed"][e, i] = x, y, speed
    st["lane"][e, i] = st["target_lane"][e, i] = lane
    st["target_speed"][e, i], st["speed_index"][e, i] = target_speed, speed_index
    st["flags"][e, i] = _abi.F_CONTROLLED | _abi.F_CHECK_COLLISIONS


def test_step_free_vehicle_keeps_its_speed(backend):
    """test_dynamics.py:11-18 / test_control.py:11-18: speed 20 for 2 s -> x = 40, y = 0, heading 0."""
    cfg, c = _highway(1, 0, 1)
    eng = make_engine(backend, c)
    st = _abi.alloc_state(1, c.num_vehicles)
    _one_controlled(st, 0, 0, 0.0, 0.0, 20.0, 0, 20.0, 0)
    eng.set_state(st)
    eng.step_frames(None, 2 * FPS)
    got = eng.get_state()
    assert got["x"][0, 0] == pytest.approx(40) and got["y"][0, 0] == pytest.approx(0)
    assert got["speed"][0, 0] == pytest.approx(20) and got["heading"][0, 0] == pytest.approx(0)
    eng.close()


def test_lane_change(backend):
    """test_control.py:21-37: LANE_RIGHT, 3 s later the vehicle sits on lane 1 (y = 4 +- 1) at the same speed."""
    cfg, c = _highway(1, 0, 2)
    eng = make_engine(backend, c)
    st = _abi.alloc_state(1, c.num_vehicles)
    _one_controlled(st, 0, 0, 0.0, 0.0, 20.0, 0, 20.0, 0)
    eng.set_state(st)
    eng.step_frames(np.array([[2]], np.int32), 3 * FPS)   # 2 == LANE_RIGHT (action.py:204)
    got = eng.get_state()
    assert got["speed"][0, 0] == pytest.approx(20)
    assert got["y"][0, 0] == pytest.approx(4.0, abs=1.0)
    assert got["lane"][0, 0] == 1
    eng.close()


def test_speed_control(backend):
    """test_control.py:40-54: FASTER, after 3 * TAU_ACC seconds the speed has risen by DELTA_SPEED (5) within 0.5."""
    cfg, c = _highway(1, 0, 1)
    eng = make_engine(backend, c)
    st = _abi.alloc_state(1, c.num_vehicles)
    _one_controlled(st, 0, 0, 0.0, 0.0, 20.0, 0, 20.0, 0)
    eng.set_state(st)
    eng.step_frames(np.array([[3]], np.int32), int(3 * 0.6 * FPS))   # 3 == FASTER
    got = eng.get_state()
    assert got["speed"][0, 0] == pytest.approx(25.0, abs=0.5)
    assert got["y"][0, 0] == pytest.approx(0) and got["lane"][0, 0] == 0
    eng.close()


def test_collision(backend):
    """test_dynamics.py:66-73: two vehicles 4 m apart on the same lane have crashed after one call."""
    cfg, c = _highway(1, 1, 1)
    eng = make_engine(backend, c)
    st = _abi.alloc_state(1, c.num_vehicles)
    _one_controlled(st, 0, 0, 0.0, 0.0, 10.0, 0, 20.0, 0)
    st["x"][0, 1], st["speed"][0, 1], st["target_speed"][0, 1], st["delta"][0, 1] = 4.0, 20.0, 20.0, 4.0
    st["flags"][0, 1] = _abi.F_CHECK_COLLISIONS
    eng.set_state(st)
    eng.step_frames(None, 1)
    got = eng.get_state()
    assert (got["flags"][0] & _abi.F_CRASHED).all()
    eng.close()


def test_stop_before_obstacle(backend):
    """test_behavior.py:11-27: an IDM vehicle at 20 m/s, 80 m behind a stationary obstacle on a one-lane road, stands
    DISTANCE_WANTED (10 m) short of it 10 s later, not crashed.  The straight-road scenarios have no Obstacle class
    (the merge networks do, but there MOBIL would simply change lane), so the obstacle is a stationary wreck; an IDM
    follower treats both alike (behavior.py:150-190)."""
    cfg, c = _highway(1, 2, 1)
    eng = make_engine(backend, c)
    st = _abi.alloc_state(1, c.num_vehicles)
    _one_controlled(st, 0, 0, 500.0, 0.0, 25.0, 0, 25.0, 1)           # the ego, far ahead and out of the way
    st["x"][0, 1], st["speed"][0, 1], st["target_speed"][0, 1], st["delta"][0, 1] = 0.0, 20.0, 20.0, 4.0
    st["flags"][0, 1] = _abi.F_CHECK_COLLISIONS
    st["x"][0, 2], st["speed"][0, 2], st["target_speed"][0, 2], st["delta"][0, 2] = 80.0, 0.0, 0.0, 4.0
    st["flags"][0, 2] = _abi.F_CHECK_COLLISIONS | _abi.F_CRASHED   # stationary: a crashed vehicle brakes to 0 and stays
    eng.set_state(st)
    eng.step_frames(None, 10 * FPS)
    got = eng.get_state()
    assert not (got["flags"][0, 1] & _abi.F_CRASHED)
    assert got["x"][0, 1] == pytest.approx(got["x"][0, 2] - 10.0, abs=1)
    assert got["y"][0, 1] == pytest.approx(0)
    assert got["speed"][0, 1] == pytest.approx(0, abs=1)
    assert got["heading"][0, 1] == pytest.approx(0)
    eng.close()


@pytest.mark.parametrize("spec", ["highway-v0", "highway-fast-v0", "merge-v0", "merge-v1", "merge-generic-v0", "merge-generic-v1",
                                  "intersection-v0", "intersection-v2"])
def test_env_step(backend, spec):
    """test_gym.py:65-91: reset, then random actions until terminated / truncated; every observation is finite, has the
    space's shape and (normalised + clipped features) stays in [-1, 1]."""
    from highwayenv_amd import envs
    from tests.test_envs_host import _emu_factory
    base = {"highway-v0": envs.BatchedHighwayEnv, "highway-fast-v0": envs.BatchedHighwayEnvFast,
            "merge-v0": envs.BatchedMergeEnv, "merge-v1": envs.BatchedConnectedLaneMergeEnv,
            "merge-generic-v0": envs.BatchedMergeGenericEnv, "merge-generic-v1": envs.BatchedConnectedLaneMergeGenericEnv,
            "intersection-v0": envs.BatchedIntersectionEnv, "intersection-v2": envs.BatchedConnectedLaneIntersectionEnv}[spec]
    cls = base if backend == "hip" else type("Emu" + base.__name__, (base,), {"_engine_factory": staticmethod(_emu_factory)})
    env = cls(num_envs=1)
    obs, info = env.reset(seed=3)
    shape = env.single_observation_shape
    rng = np.random.default_rng(0)
    done, steps = False, 0
    while not done and steps < 60:
        assert obs.shape == (1, *shape) and np.isfinite(obs).all() and (np.abs(obs) <= 1 + 1e-6).all()
        obs, reward, term, trunc, info = env.step(rng.integers(0, env.single_action_space.n, size=1))
        assert np.isfinite(reward).all()
        done = bool(term[0] or trunc[0])
        steps += 1
    assert done
    env.close()


@pytest.mark.parametrize("connected", [False, True], ids=["v0", "v1-connected"])
def test_neighbours_across_connected_segments(backend, connected):
    """/root/reference/tests/road/test_neighbour_vehicles.py:199-296, restated through what the neighbours DO (the
    engine does not expose them): on the merge network a->b (x 0..230) -> b->c (230..310),

      env 0  test_front_on_next_segment / test_connected_segments_ignored_by_default: a slow car 15 m ahead but on the
             NEXT segment makes an IDM follower brake only with neighbour_vehicles_connected_lanes;
      env 1  test_multi_lane_same_lane_id: the same car on lane 1 of the next segment is nobody's leader on lane 0;
      env 2  test_closer_same_segment_preferred_over_next_segment: the closer same-segment leader decides, flag or not;
      env 3  test_rear_on_previous_segment: MOBIL refuses a lane change that would make a fast car on the PREVIOUS
             segment brake hard -- without the flag that car is invisible and the change goes ahead.

    The oracle must agree on every number; it is also what says which way each case goes."""
    from highwayenv_amd import merge
    from oracle import oracle
    config = merge.merge_default_config()
    config["neighbour_vehicles_connected_lanes"] = connected
    E = 4
    c = _abi.make_config(config, E, scenario="merge")
    assert bool(c.flags & _abi.C_CONNECTED_LANES) is connected
    st = merge.spawn_reference_stream(c, config, False, np.arange(E))
    AB0, BC0, BC1, CD1 = 0, 2, 3, 6   # [ab0 ab1 | bc0 bc1 lbc | cd0 cd1 | jk | kb]

    def car(e, i, x, y, v, lane, target=20.0, timer=0.0):
        st["x"][e, i], st["y"][e, i], st["heading"][e, i], st["speed"][e, i] = x, y, 0.0, v
        st["lane"][e, i] = st["target_lane"][e, i] = lane
        st["target_speed"][e, i], st["timer"][e, i], st["delta"][e, i] = target, timer, 4.0
        st["flags"][e, i] = _abi.F_CHECK_COLLISIONS

    for e in range(E):
        st["flags"][e, 1:5] = _abi.F_ABSENT
        st["x"][e, 0], st["y"][e, 0], st["lane"][e, 0], st["target_lane"][e, 0] = 420.0, 4.0, CD1, CD1  # ego: far ahead
    car(0, 1, 225.0, 0.0, 15.0, AB0); car(0, 2, 240.0, 0.0, 2.0, BC0, target=2.0)
    car(1, 1, 225.0, 0.0, 15.0, AB0); car(1, 2, 240.0, 4.0, 2.0, BC1, target=2.0)
    car(2, 1, 200.0, 0.0, 15.0, AB0); car(2, 2, 212.0, 0.0, 8.0, AB0, target=8.0); car(2, 3, 240.0, 0.0, 0.0, BC0, target=0.0)
    # env 3: slot 1 on bc lane 1 behind a slow leader, about to decide (timer due); a fast car on ab lane 0 right behind
    car(3, 1, 236.0, 4.0, 12.0, BC1, timer=1.01); car(3, 2, 250.0, 4.0, 3.0, BC1, target=3.0)
    car(3, 3, 222.0, 0.0, 20.0, AB0)   # s = -8 on bc0: beyond on_lane's VEHICLE_LENGTH margin
    ref = _abi.copy_state(st)
    eng = make_engine(backend, c)
    eng.set_state(st)
    eng.step_frames(None, 1)
    oracle.frames(c, ref, None, 1)
    got = eng.get_state()
    for k in ["x", "y", "heading", "speed"]:
        np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=1e-9, err_msg=k)
    np.testing.assert_array_equal(got["target_lane"], ref["target_lane"])
    v = got["speed"][:, 1]
    assert bool(v[0] < 15.0) == connected and bool(v[0] > 15.0) != connected   # brakes only when the next segment is searched
    assert v[1] > 15.0                                                       # lane 1 of the next segment: never a leader
    assert v[2] < 15.0                                                       # same-segment leader decides either way
    assert bool(got["target_lane"][3, 1] == BC1) == connected                 # the unseen follower lets the change through
    assert bool(got["target_lane"][3, 1] == BC0) != connected
    eng.close()


def test_network(backend):
    """/root/reference/tests/road/test_road.py:10-41: a ControlledVehicle without a route on a diamond of five straight lanes
    (0->1, 1->2, 2->0, 1->3, 3->0) keeps choosing the closest lane leaving each node (RoadNetwork.next_lane, road.py:
    73-133): its target lane changes at least 3 times in 20 s at 15 Hz.  The lane table of the intersection scenario is
    replaced by the diamond through the ABI's general lane table (hwy_config.gnet); the oracle must follow frame by frame."""
    from highwayenv_amd import intersection as hix
    from oracle import oracle_ix
    from tests.golden_util import ix_oracle_config, ix_oracle_state
    cfg = hix.intersection_default_config()
    cfg.update({"host_traffic": True, "max_vehicles": 4})
    c = _abi.make_config(cfg, 1, scenario="intersection")
    oc = ix_oracle_config(cfg, c, 1)
    diamond = [((0, 0), (10, 0), 0, 1), ((10, 0), (5, 5), 1, 2), ((5, 5), (0, 0), 2, 0), ((10, 0), (5, -5), 1, 3),
               ((5, -5), (0, 0), 3, 0)]
    c.gnet_lanes = oc.n_lanes = len(diamond)
    for q in range(4):  # (the spawn entries of the intersection: unused with host traffic, but validated by hwy_create)
        c.access_lane[q] = c.exit_of[q] = 0
    for k, (s, e, f, t) in enumerate(diamond):
        d = np.asarray(e, float) - np.asarray(s, float)
        length = float(np.linalg.norm(d))
        d = d / length
        for g in (c.gnet[k], oc.lanes[k]):
            g.kind, g.direction, g.priority, g.forbidden, g.from_node, g.to_node, g.exit_lane = 0, 0, 0, 0, f, t, 0
            g.sx, g.sy, g.heading, g.dirx, g.diry = float(s[0]), float(s[1]), float(np.arctan2(d[1], d[0])), float(d[0]), float(d[1])
            g.length, g.width, g.speed_limit = length, 4.0, 20.0   # StraightLane defaults (lane.py:150-181)
        oc.lanes[k].id = 0
    st = _abi.alloc_state_ix(1, c.num_vehicles)
    # ControlledVehicle(road, [5, 0], heading=0, target_speed=2): speed 0, lane index (0, 1, 0), no route
    st["x"][0, 0], st["y"][0, 0], st["target_speed"][0, 0] = 5.0, 0.0, 2.0
    st["flags"][0, 0] = _abi.F_CONTROLLED | _abi.F_CHECK_COLLISIONS
    ref = ix_oracle_state(st, c)
    eng = make_engine(backend, c)
    eng.set_state(st)
    assert eng.get_state()["lane"][0, 0] == 0
    lane_changes, tgt = 0, 0
    for _ in range(int(20 * FPS)):
        eng.step_frames(None, 1)
        oracle_ix.frames(oc, ref, None, 1)
        got = eng.get_state()
        assert abs(got["x"][0, 0] - ref["x"][0, 0]) < 1e-7 and abs(got["y"][0, 0] - ref["y"][0, 0]) < 1e-7
        assert got["target_lane"][0, 0] == ref["target_lane"][0, 0]
        if got["target_lane"][0, 0] != tgt:
            tgt = got["target_lane"][0, 0]
            lane_changes += 1
    assert lane_changes >= 3
    eng.close()


def test_occupancy_grid_shape_no_uint8_overflow(backend):
    """/root/reference/tests/envs/test_observations.py:27-43: a grid with more than 255 cells along an axis keeps its shape
    (the reference once cast grid_shape to uint8): highway-v0 with a 600 m x 20 m grid of 2 m cells -> (4, 300, 10)."""
    from highwayenv_amd import envs
    from tests.test_envs_host import _emu_factory
    base = envs.HighwayEnv
    cls = base if backend == "hip" else type("EmuHighwayEnv", (base,), {"_engine_factory": staticmethod(_emu_factory)})
    env = cls({"observation": {"type": "OccupancyGrid", "grid_size": [[-300, 300], [-10, 10]], "grid_step": [2, 2]}})
    obs, _ = env.reset(seed=0)
    assert env.single_observation_shape == (4, 300, 10) and obs.shape == (4, 300, 10) and obs.dtype == np.float32
    assert obs[0].sum() >= 1 and obs[3].sum() > 0   # the ego's own cell; the road layer
    obs, reward, terminated, truncated, info = env.step(1)
    assert obs.shape == (4, 300, 10) and np.isfinite(obs).all()
    env.close()
