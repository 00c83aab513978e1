"""Pin the intersection C oracle (oracle/hwy_oracle_ix.c) against traces of the unmodified reference's
IntersectionEnv (tests/golden/intersection_*.npz from tests/golden/make_golden_intersection.py).

CPU only.  This is the checker of the NEXT hot-path row (CircularLane, routes,
RegulatedRoad, dynamic spawn / clear); the HIP kernel for it does not exist yet.
Tolerances as in test_oracle_golden.py: f64 with glibc libm vs numpy's libm; flags / lane indices / routes exact.
"""
import numpy as np
import pytest

from oracle import oracle
from tests.golden_util import (KNIFE, INTERSECTION_CRASH, INTERSECTION, INTERSECTION_GRID, INTERSECTION_INTENTIONS, INTERSECTION_MA, INTERSECTION_MA_FRAMES,
                               GoldenIntersection,
                               assert_ix_state_close)


def _sub(st, sel):
    return {k: np.ascontiguousarray(v[sel]) for k, v in st.items()}


def check_teacher_forced_frames(g, coverage=True):
    """Every single frame (meta-action on the first one, Road.act, RegulatedRoad.step incl. the regulation every
    7th frame), started from the reference's own state.  Impacts are compared SIGNED wherever the collision is well
    conditioned (oracle.impact_margins >= KNIFE)."""
    name = g.name
    ix = g.ix
    Ef = g.frames_for
    cfg = g.ix_config(Ef)
    envs = slice(0, Ef)
    steps0 = g.z["road_steps0"][:Ef]
    n_yield = 0
    for step in range(g.steps):
        for fr in range(g.T):
            k = step * g.T + fr
            if fr == 0:
                st = g.state("init", envs=envs) if step == 0 else g.state("next", step - 1, envs=envs)
            else:
                st = g.state("frame", k - 1)
            st["road_steps"][...] = steps0 + k
            acts = g.actions[step, :Ef] if fr == 0 else None
            with oracle.impact_margins(cfg) as m:
                ix.frames(cfg, st, acts, 1)
            want = g.state("frame", k)
            assert_ix_state_close(st, want, atol=1e-10, what=f"{name} step {step} frame {fr}", signed=m.margin >= KNIFE)
            n_yield += int(want["is_yielding"].sum())
    assert n_yield > 0 or not coverage  # the fixtures do exercise the regulation


@pytest.mark.parametrize("name", INTERSECTION + INTERSECTION_MA_FRAMES + INTERSECTION_CRASH)
def test_oracle_teacher_forced_frames(name):
    check_teacher_forced_frames(GoldenIntersection(name))


def check_steps_observation_reward_and_clear_spawn(g, coverage=True):
    """Whole policy steps from the reference's state at the start of each step: state before clear/spawn, obs, reward,
    terminated / truncated, info; then _clear_vehicles + _spawn_vehicle replayed on the recorded draws.  Steps WITH a
    collision are compared like any other (terminal observation and reward included, impacts signed) unless one of the
    collisions is on the knife edge (oracle.impact_margins < KNIFE)."""
    name = g.name
    ix = g.ix
    cfg = g.ix_config()
    steps0 = g.z["road_steps0"]
    np.testing.assert_allclose(ix.observe(cfg, g.state("init")), g.z["obs0"], rtol=0, atol=1e-6)
    live = np.ones(g.E, bool)
    n_spawned = n_cleared = n_wreck = n_wreck_full = 0
    for t in range(g.steps):
        st = g.state("init") if t == 0 else g.state("next", t - 1)
        st["road_steps"][...] = steps0 + t * g.T
        st["time"][...] = float(t)
        v0 = st["speed"].copy()
        with oracle.impact_margins(cfg) as m:
            obs, reward, term, trunc, info = ix.step(cfg, st, g.actions[t])
        want = g.state("step", t)
        what = f"{name} step {t}"
        wreck = ((want["present"] != 0) & ((want["crashed"] != 0) | (want["has_impact"] != 0))).any(1)
        clean = live & (m.margin.min(1) >= KNIFE)
        n_wreck += int((live & wreck).sum())
        n_wreck_full += int((clean & wreck).sum())
        # (1e-8 after 15 free-running frames; 1e-6 for a car below 2 m/s at either end of the step: steering_control divides by
        #  not_zero(speed) twice -- live-reference case 229: 2e-8 on a car pulling away at 1 m/s)
        assert_ix_state_close(_sub(st, clean), _sub(want, clean), atol=1e-8, what=what, signed=(m.margin >= KNIFE)[clean],
                              slow_atol=1e-6, slow_start=v0[clean])
        np.testing.assert_array_equal(term[live], g.z["terminated"][t].astype(bool)[live], err_msg=what)
        np.testing.assert_array_equal(trunc[live], g.z["truncated"][t].astype(bool)[live], err_msg=what)
        np.testing.assert_array_equal(info["crashed"][live], g.z["info_crashed"][t].astype(bool)[live], err_msg=what)
        np.testing.assert_allclose(obs[clean], g.z["obs"][t][clean], rtol=0, atol=1e-6, err_msg=what)
        np.testing.assert_allclose(reward[clean], g.z["reward"][t][clean], rtol=0, atol=1e-9, err_msg=what)
        np.testing.assert_allclose(info["speed"][clean], g.z["info_speed"][t][clean], rtol=0, atol=1e-9, err_msg=what)
        if "agents_rewards" in g.z:  # MultiAgentIntersectionEnv's info (intersection_env.py:114-122)
            np.testing.assert_array_equal(info["agents_terminated"][live], g.z["agents_terminated"][t].astype(bool)[live], err_msg=what)
            np.testing.assert_allclose(info["agents_rewards"][clean], g.z["agents_rewards"][t][clean], rtol=0, atol=1e-9, err_msg=what)
        # clear + spawn from the reference's own pre-clear state
        st2 = g.state("step", t)
        used = ix.clear_spawn(cfg, st2, g.z["draws"][t], g.z["n_draws"][t])
        np.testing.assert_array_equal(used, g.z["n_draws"][t], err_msg=f"{what}: draws consumed")
        nxt = g.state("next", t)
        assert_ix_state_close(st2, nxt, atol=1e-12, what=f"{what}: after clear/spawn")
        np.testing.assert_allclose(st2["delta"][nxt["present"] != 0], nxt["delta"][nxt["present"] != 0], rtol=0, atol=0)
        n_spawned += int((nxt["present"].sum(1) > want["present"].sum(1)).sum())
        for e in range(g.E):
            before = set(want["vid"][e][want["present"][e] != 0].tolist())
            after = set(nxt["vid"][e][nxt["present"][e] != 0].tolist())
            n_cleared += len(before - after)
        live &= ~g.z["terminated"][t].astype(bool)
    assert not coverage or (n_spawned > 0 and (n_cleared > 0 or name != "intersection_dense"))
    print(f"\n{name}: {n_wreck} env-steps with a wreck on the road, {n_wreck_full} compared in full")
    if name in INTERSECTION_CRASH:
        assert n_wreck_full >= 0.9 * n_wreck > 0


@pytest.mark.parametrize("name", INTERSECTION + INTERSECTION_GRID + INTERSECTION_MA + INTERSECTION_INTENTIONS + INTERSECTION_CRASH)
def test_oracle_steps_observation_reward_and_clear_spawn(name):
    check_steps_observation_reward_and_clear_spawn(GoldenIntersection(name))


@pytest.mark.parametrize("name", INTERSECTION + INTERSECTION_MA)
def test_oracle_free_running_episodes(name):
    """reset state -> whole episodes on the oracle's own state (steps, clear, spawn on the recorded draws), compared
    while the episode is live, no wreck is on the road and no vehicle has (nearly) stopped:
    steering_control divides by not_zero(speed) twice (controller.py:172-180), so below ~0.5 m/s the heading of a
    braking ego amplifies the last-bit libm differences between numpy and glibc by 1e2..1e4 per frame (measured:
    1e-15 -> 1e-5 in 30 frames at 0.03 m/s) -- every single frame of that regime is still pinned at 1e-10 by the
    teacher-forced test above."""
    g = GoldenIntersection(name)
    ix = g.ix
    cfg = g.ix_config()
    st = g.state("init", road_steps=g.z["road_steps0"])
    live = np.ones(g.E, bool)
    compared = 0
    for t in range(g.steps):
        obs, reward, term, trunc, info = ix.step(cfg, st, g.actions[t])
        want = g.state("step", t)
        what = f"{name} step {t}"
        wreck = ((want["present"] != 0) & ((want["crashed"] != 0) | (want["has_impact"] != 0))).any(1)
        np.testing.assert_array_equal(term[live], g.z["terminated"][t].astype(bool)[live], err_msg=what)
        np.testing.assert_array_equal(trunc[live], g.z["truncated"][t].astype(bool)[live], err_msg=what)
        stalled = ((want["present"] != 0) & (np.abs(want["speed"]) < 0.5)).any(1)
        live &= ~wreck & ~stalled
        assert_ix_state_close(_sub(st, live), _sub(want, live), atol=1e-8, what=what)
        np.testing.assert_allclose(obs[live], g.z["obs"][t][live], rtol=0, atol=1e-6, err_msg=what)
        np.testing.assert_allclose(reward[live], g.z["reward"][t][live], rtol=0, atol=1e-9, err_msg=what)
        compared += int(live.sum())
        ix.clear_spawn(cfg, st, g.z["draws"][t], g.z["n_draws"][t])
        nxt = g.state("next", t)
        assert_ix_state_close(_sub(st, live), _sub(nxt, live), atol=1e-8, what=f"{what}: after clear/spawn")
        live &= ~g.z["terminated"][t].astype(bool)
    assert compared >= 2 * g.E


def test_connected_lanes_flag_is_load_bearing():
    """intersection-v2 (neighbour_vehicles_connected_lanes, road.py:508-529): without the flag the restatement must NOT
    reproduce the reference's ConnectedLaneIntersectionEnv frames, i.e. the fixture does exercise the connected search."""
    g = GoldenIntersection("intersection_v2")
    assert g.config["neighbour_vehicles_connected_lanes"] is True
    Ef = g.frames_for
    cfg = g.ix_config(Ef)
    assert cfg.connected_lanes == 1
    cfg.connected_lanes = 0
    envs = slice(0, Ef)
    steps0 = g.z["road_steps0"][:Ef]
    worst = 0.0
    for step in range(g.steps):
        for fr in range(g.T):
            k = step * g.T + fr
            if fr == 0:
                st = g.state("init", envs=envs) if step == 0 else g.state("next", step - 1, envs=envs)
            else:
                st = g.state("frame", k - 1)
            st["road_steps"][...] = steps0 + k
            g.ix.frames(cfg, st, g.actions[step, :Ef] if fr == 0 else None, 1)
            want = g.state("frame", k)
            pres = want["present"] != 0
            worst = max(worst, float(np.abs(st["speed"] - want["speed"])[pres].max()))
    assert worst > 1e-3
