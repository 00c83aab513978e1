"""Pile-ups: more close pairs than one pass of the pair list holds (64 pairs per wavefront per pass -- csrc/hwy_wave.h,
hwy_device.h, hwy_net.h: collision candidates; csrc/hwy_ix.h: ix_for_pairs for collisions, regulation conflicts and arc
projections).  The bodies are placed by hand so that hundreds of pairs pass the sphere pre-check and the separation test in
ONE frame; the kernels must agree with the oracle's literal all-pairs loops (road.py:477-481, regulation.py:36-68) on every
flag, on the magnitude of every pending impact (its sign is the reference's knife edge, tests/golden_util.py) and, one frame
later, on what the impacts did to the positions wherever the sign is not on that edge.
"""
import numpy as np
import pytest

from highwayenv_amd import _abi, spawn
from oracle import oracle
from tests.backends import BACKENDS, make_engine
from tests.golden_util import assert_state_close


def _pile(st, rng, lanes, x0=300.0, pitch=0.3):
    E, N = st["x"].shape
    for e in range(E):
        order = rng.permutation(N)  # list order != order along the road
        for r, i in enumerate(order):
            lane = int(rng.integers(0, lanes))
            st["x"][e, i] = x0 + pitch * r + rng.uniform(-0.2, 0.2)
            st["y"][e, i] = 4.0 * lane + rng.uniform(-1.2, 1.2)
            st["heading"][e, i] = rng.uniform(-0.3, 0.3)
            st["speed"][e, i] = rng.uniform(15.0, 30.0)
            st["lane"][e, i] = st["target_lane"][e, i] = lane


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("n_vehicles", [50, 100])  # one-wavefront kernel (hwy_wave.h) / two wavefronts per env (hwy_device.h)
def test_highway_pileup_needs_several_passes(backend, n_vehicles):
    cfg_d = _abi.highway_default_config()
    cfg_d.update({"vehicles_count": n_vehicles, "lanes_count": 4})
    E = 3
    cfg = _abi.make_config(cfg_d, E, fast=False)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 7, cfg_d["ego_spacing"], cfg_d["vehicles_density"], cfg_d["initial_lane_id"])
    _pile(st, np.random.default_rng(5), 4)
    dx, dy = st["x"][:, :, None] - st["x"][:, None, :], st["y"][:, :, None] - st["y"][:, None, :]
    overlapping = (np.abs(dx) < 4.5) & (np.abs(dy) < 1.5)  # (a lower bound of the pairs that reach the SAT)
    assert ((overlapping.sum((1, 2)) - st["x"].shape[1]) // 2 > 100).all()  # more than one pass holds, per environment
    ref = _abi.copy_state(st)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    eng.step_frames(None, 1)
    oracle.frames(cfg, ref, None, 1)
    got = eng.get_state()
    n_hit = int(((ref["flags"] & _abi.F_HAS_IMPACT) != 0).sum())
    assert n_hit > 0.9 * st["x"].size  # nearly every vehicle is hit, by several others: a few hundred pairs per environment
    assert_state_close(got, ref, atol=1e-9, what="pile-up, frame 1")
    # the impacts are applied by the next frame's integration: compare where both agree on the sign of the impact
    same_sign = (np.abs(got["impact_x"] - ref["impact_x"]) < 1e-9) & (np.abs(got["impact_y"] - ref["impact_y"]) < 1e-9)
    assert same_sign.mean() > 0.95
    eng.step_frames(None, 1)
    oracle.frames(cfg, ref, None, 1)
    got2 = eng.get_state()
    np.testing.assert_array_equal(got2["flags"] & _abi.F_CRASHED, ref["flags"] & _abi.F_CRASHED)
    for k in ("x", "y", "heading", "speed"):
        np.testing.assert_allclose(got2[k][same_sign], ref[k][same_sign], rtol=0, atol=1e-9, err_msg=k)
    eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_merge_pileup_needs_several_passes(backend):
    from highwayenv_amd import merge
    cfg_d = merge.merge_generic_default_config()
    cfg_d.update({"lanes_count": 4, "vehicles_count": 40, "controlled_vehicles": 2,
                  "action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
                  "observation": {"type": "MultiAgentObservation", "observation_config": {"type": "Kinematics"}}})
    E = 3
    cfg = _abi.make_config(cfg_d, E, scenario="merge-generic")
    eng = make_engine(backend, cfg)
    eng.reset(seeds=np.arange(E, dtype=np.uint64) + 3)
    st = eng.get_state()
    rng = np.random.default_rng(9)
    veh = ((st["flags"] & (_abi.F_ABSENT | _abi.F_OBSTACLE)) == 0)
    for e in range(E):
        idx = np.nonzero(veh[e])[0]
        for r, i in enumerate(rng.permutation(idx)):
            lane = int(rng.integers(0, 4))
            st["x"][e, i] = 60.0 + 0.3 * r + rng.uniform(-0.2, 0.2)   # on the first segment ("a" -> "b")
            st["y"][e, i] = 4.0 * lane + rng.uniform(-1.2, 1.2)
            st["heading"][e, i] = rng.uniform(-0.3, 0.3)
            st["speed"][e, i] = rng.uniform(15.0, 30.0)
            st["lane"][e, i] = st["target_lane"][e, i] = lane
    ref = _abi.copy_state(st)
    eng.set_state(st)
    eng.step_frames(None, 1)
    oracle.frames(cfg, ref, None, 1)
    got = eng.get_state()
    assert int(((ref["flags"] & _abi.F_HAS_IMPACT) != 0).sum()) > 0.9 * int(veh.sum())
    pres = (ref["flags"] & _abi.F_ABSENT) == 0
    np.testing.assert_array_equal(got["flags"][pres], ref["flags"][pres])
    np.testing.assert_array_equal(got["lane"][pres], ref["lane"][pres])
    for k in ("x", "y", "heading", "speed"):
        np.testing.assert_allclose(got[k][pres], ref[k][pres], rtol=0, atol=1e-9, err_msg=k)
    for k in ("impact_x", "impact_y"):
        np.testing.assert_allclose(np.abs(got[k][pres]), np.abs(ref[k][pres]), rtol=0, atol=1e-9, err_msg=k)
    eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("slots,tuning", [(30, None), (30, {"ix_no_helpers": 1}), (48, None)],
                         ids=["helper-lanes", "32-threads", "64-slots"])  # the three builds of csrc/hwy_ix.h
def test_intersection_pileup_needs_several_passes(backend, slots, tuning):
    """A crowd inside the junction: hundreds of collision candidates, regulation conflicts and (vehicle, arc) projections in
    one frame -- the frame is a regulation frame (RegulatedRoad.steps % 7 == 6 before it)."""
    from highwayenv_amd import intersection as hix
    from oracle import oracle_ix
    from tests.golden_util import ix_oracle_config, ix_oracle_state
    cfg_d = hix.intersection_default_config()
    cfg_d.update(max_vehicles=slots, initial_vehicle_count=slots - 4, spawn_probability=1.0, host_traffic=True)
    E = 3
    c = _abi.make_config(cfg_d, E, scenario="intersection", tuning=tuning)
    eng = make_engine(backend, c)
    eng.reset(seeds=np.arange(E, dtype=np.uint64) + 21)
    st = eng.get_state()
    rng = np.random.default_rng(4)
    tab = hix.table_from_config(c)
    filled = 0
    for e in range(E):
        n0 = int(((st["flags"][e] & _abi.F_ABSENT) == 0).sum())
        ego = int(np.argmax((st["flags"][e] & _abi.F_CONTROLLED) != 0))
        # fill the empty slots with copies of present IDM vehicles (same lanes / routes), then crowd everybody
        for i in range(n0, slots - 1):
            src = int(rng.integers(0, n0))
            src = src if src != ego else (src + 1) % n0
            for k, a in st.items():
                if a.ndim == 2:
                    a[e, i] = a[e, src]
            filled += 1
        n = slots - 1
        st["x"][e, :n] = rng.uniform(-9.0, 9.0, n)
        st["y"][e, :n] = rng.uniform(-9.0, 9.0, n)
        st["heading"][e, :n] = rng.uniform(-np.pi, np.pi, n)
        st["speed"][e, :n] = rng.uniform(2.0, 9.0, n)
        for i in range(n):  # a consistent lane index for the new pose (on_state_update of the previous frame)
            st["lane"][e, i] = hix.closest_lane(tab, (st["x"][e, i], st["y"][e, i]), st["heading"][e, i])
    st["road_steps"][...] = 6
    assert filled > 0
    eng.set_state(st)
    oc = ix_oracle_config(cfg_d, c, E)
    ost = ix_oracle_state(st, c)
    eng.step_frames(None, 1)
    oracle_ix.frames(oc, ost, None, 1)
    got = ix_oracle_state(eng.get_state(), c)
    pres = ost["present"] != 0
    assert int((ost["has_impact"][pres] != 0).sum()) > 0.8 * pres.sum() and int(ost["is_yielding"][pres].sum()) > E
    for k in ("present", "lane", "target_lane", "crashed", "has_impact", "is_yielding"):
        np.testing.assert_array_equal(got[k][pres], ost[k][pres], err_msg=k)
    for k in ("x", "y", "heading", "speed", "target_speed"):
        np.testing.assert_allclose(got[k][pres], ost[k][pres], rtol=0, atol=1e-9, err_msg=k)
    for k in ("impact_x", "impact_y"):
        np.testing.assert_allclose(np.abs(got[k][pres]), np.abs(ost[k][pres]), rtol=0, atol=1e-9, err_msg=k)
    eng.close()
