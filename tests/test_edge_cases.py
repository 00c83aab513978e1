"""Edge cases of the hot path against the oracle: degenerate sizes, maximum sizes, unusual frequencies,
custom speed ladders, road ends, off-road termination.  Same backends as the parity tests."""
import numpy as np
import pytest

from highwayenv_amd import _abi, spawn
from oracle import oracle
from tests.backends import BACKENDS, make_engine
from tests.golden_util import assert_obs_close, assert_state_close


def rollout(backend, cfg_d, fast, E, steps, seed, mutate=None, actions=None, compare_wrecks=False):
    cfg = _abi.make_config(cfg_d, E, fast=fast)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 100 * seed, cfg_d["ego_spacing"], cfg_d["vehicles_density"],
                                      cfg_d["initial_lane_id"])
    if mutate:
        mutate(st)
    ref = _abi.copy_state(st)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    rng = np.random.default_rng(seed)
    live = np.ones(E, bool)
    for t in range(steps):
        acts = (rng.integers(0, _abi.num_actions(cfg), size=(E, cfg.num_agents)) if actions is None else np.full((E, cfg.num_agents), actions[t % len(actions)])).astype(np.int32)
        obs, reward, term, trunc, info = eng.step(acts)
        with oracle.impact_margins(cfg) as m:
            o2, r2, te2, tr2, i2 = oracle.step(cfg, ref, acts)
        wreck = ((ref["flags"] & (_abi.F_CRASHED | _abi.F_HAS_IMPACT)) != 0).any(1)
        # the step of an env's first collision is compared like any other (terminal observation, reward, positions) unless a
        # push direction sits on the knife edge (|d.normal| < 1e-9, utils.py:232-236); assert_state_close compares |impact|
        # (free-running: the engine's state may differ from the oracle's by the 1e-7 the previous steps are held to, so a push
        #  direction decided by |d.normal| below 1e-6 can flip -- two cars tracking one lane centre are that close laterally)
        ok = live & (~wreck | compare_wrecks | (m.margin.min(1) >= 1e-6))
        what = f"step {t}"
        np.testing.assert_array_equal(term[live], te2[live], err_msg=what)
        np.testing.assert_array_equal(trunc, tr2, err_msg=what)
        np.testing.assert_allclose(reward[ok], r2[ok], rtol=0, atol=1e-9, err_msg=what)
        got = eng.get_state()
        # free-running (no re-synchronisation of live environments): every step without a collision at 1e-7; the step of a
        # collision at 2.5e-6 -- the minimum-translation vector is a difference of projected corner coordinates, i.e. it carries
        # the two bodies' heading differences times a 2.7 m lever arm on top of their position differences (largest seen in 20 000
        # random configurations: 1.2e-6 m); un-normalised relative features are differences of two such positions
        for rows, atol, atol_obs in ((ok & ~wreck, 1e-7, 1e-6), (ok & wreck, 2.5e-6, 5e-6)):
            assert_obs_close(obs[rows], o2[rows], bool(cfg.flags & _abi.C_GRID_IMAGE), what, atol=atol_obs)
            assert_state_close({k: v[rows] for k, v in got.items()}, {k: v[rows] for k, v in ref.items()}, atol=atol, what=what)
        live &= ~wreck
        if not live.all():  # keep dead envs in lock-step with the oracle so that live ones stay comparable
            for k in got:
                got[k][~live] = ref[k][~live]
            eng.set_state(got)
    eng.close()
    return ref


@pytest.mark.parametrize("backend", BACKENDS)
def test_ego_alone_on_the_road(backend):
    cfg = _abi.highway_fast_default_config()
    cfg.update({"vehicles_count": 0, "lanes_count": 2, "duration": 6})
    ref = rollout(backend, cfg, True, 3, 8, seed=1)
    assert ref["x"].shape == (3, 1)


@pytest.mark.parametrize("backend", BACKENDS)
def test_single_lane_no_lane_changes_possible(backend):
    cfg = _abi.highway_fast_default_config()
    cfg.update({"vehicles_count": 15, "lanes_count": 1})
    ref = rollout(backend, cfg, True, 4, 10, seed=2)
    assert (ref["lane"] == 0).all()


@pytest.mark.parametrize("backend", BACKENDS)
def test_maximum_sizes(backend):
    """N = 256 vehicles (4 wavefronts per env) on 16 lanes, full pairwise collisions."""
    cfg = _abi.highway_default_config()
    cfg.update({"vehicles_count": 255, "lanes_count": 16, "simulation_frequency": 5, "duration": 10})
    rollout(backend, cfg, False, 1 if backend == "emu" else 8, 2 if backend == "emu" else 6, seed=3)


@pytest.mark.parametrize("backend", BACKENDS)
def test_exactly_64_and_65_vehicles(backend):
    """The wavefront boundary: N = 64 runs the one-wavefront kernel with no idle lane, N = 65 the workgroup kernel."""
    for count in (63, 64):
        cfg = _abi.highway_fast_default_config()
        cfg.update({"vehicles_count": count, "lanes_count": 4})
        rollout(backend, cfg, True, 2 if backend == "emu" else 32, 3 if backend == "emu" else 12, seed=4 + count)


@pytest.mark.parametrize("backend", BACKENDS)
def test_unusual_frequencies_and_custom_speed_ladder(backend):
    cfg = _abi.highway_default_config()
    cfg.update({"vehicles_count": 20, "lanes_count": 3, "simulation_frequency": 12, "policy_frequency": 4, "duration": 3,
                "action": {"type": "DiscreteMetaAction", "target_speeds": [10, 17.5, 25, 32.5, 40]}})
    ref = rollout(backend, cfg, False, 4, 14, seed=5, actions=[3, 3, 3, 3, 4, 0, 2, 1])
    assert _abi.make_config(cfg, 1).frames_per_step == 3 and _abi.make_config(cfg, 1).num_target_speeds == 5
    assert (ref["time"] == 14 * 0.25).all()


@pytest.mark.parametrize("backend", BACKENDS)
def test_offroad_terminal_and_unnormalised_reward(backend):
    """Ego steered off the road: on_road_reward zeroes the reward and offroad_terminal terminates."""
    cfg = _abi.highway_fast_default_config()
    cfg.update({"vehicles_count": 6, "lanes_count": 2, "offroad_terminal": True, "normalize_reward": False,
                "collision_reward": -2.5, "right_lane_reward": 0.3, "high_speed_reward": 0.7, "reward_speed_range": [15, 35]})

    def shove(st):
        st["y"][:, 0] = [-1.9, -2.1, 5.9, 6.1]      # lane centres 0 and 4, width 4: |lat| <= 2 is on the road
        st["x"][:, 1:] += 500                       # traffic out of the way

    cfg_ = _abi.make_config(cfg, 4, fast=True)
    st = spawn.spawn_reference_stream(cfg_, np.arange(4), 1.5, 1.0)
    shove(st)
    eng = make_engine(backend, cfg_)
    eng.set_state(st)
    ref = _abi.copy_state(st)
    obs, reward, term, trunc, info = eng.step(np.ones((4, 1), np.int32))
    o2, r2, te2, tr2, _ = oracle.step(cfg_, ref, np.ones((4, 1), np.int32))
    np.testing.assert_array_equal(term, te2)
    np.testing.assert_allclose(reward, r2, rtol=0, atol=1e-9)
    np.testing.assert_allclose(obs, o2, rtol=0, atol=1e-6)
    eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_vehicles_leaving_the_road_end(backend):
    """Past x = length + 5 a vehicle is on no lane (AbstractLane.on_lane, lane.py:98-101): it drops out of every
    neighbour search and is unreachable for lane changes; lane index keeps following the closest lane."""
    cfg = _abi.highway_fast_default_config()
    cfg.update({"vehicles_count": 12, "lanes_count": 3})

    def near_the_end(st):
        st["x"] += 10000 - 40 - st["x"][:, -1:]  # the last vehicle sits 40 m before the end of the road

    rollout(backend, cfg, True, 3, 6, seed=7, mutate=near_the_end, actions=[1])


@pytest.mark.parametrize("backend", BACKENDS)
def test_truncation_fires_exactly_at_duration(backend):
    cfg = _abi.highway_fast_default_config()
    cfg.update({"vehicles_count": 5, "duration": 4, "policy_frequency": 2, "simulation_frequency": 10})
    c = _abi.make_config(cfg, 2, fast=True)
    eng = make_engine(backend, c)
    st = spawn.spawn_reference_stream(c, [0, 1], 1.5, 1.0)
    st["x"][:, 1:] += 2000
    eng.set_state(st)
    flags = [eng.step(np.ones((2, 1), np.int32))[3].copy() for _ in range(9)]
    # time += 0.5 per step; truncated <=> time >= 4  <=> from the 8th step on
    assert [bool(f.all()) for f in flags] == [False] * 7 + [True] * 2
    eng.close()


@pytest.mark.gpu
def test_nonfinite_state_is_counted():
    """NaN guard (include/hwy_engine.h: HWY_CTR_NONFINITE_STORES): a healthy run counts nothing; a NaN handed in through set_state is
    counted where the step writes the state back (and spreads to its lane mates through the gap terms)."""
    from highwayenv_amd.engine import Engine
    cfg_d = _abi.highway_fast_default_config()
    cfg = _abi.make_config(cfg_d, 8, fast=True)
    eng = Engine(cfg)
    eng.reset(base_seed=3)
    acts = np.ones((8, 1), np.int32)
    for _ in range(3):
        eng.step(acts)
    assert eng.counters()["nonfinite_stores"] == 0
    st = eng.get_state()
    st["x"][5, 7] = np.nan
    eng.set_state(st)
    eng.step(acts)
    n = eng.counters(reset=True)["nonfinite_stores"]
    assert n >= 1
    assert np.isnan(eng.get_state()["x"][5]).any() and not np.isnan(eng.get_state()["x"][[0, 1, 2, 3, 4, 6, 7]]).any()
    assert eng.counters()["nonfinite_stores"] == 0
    eng.close()
