"""Parity of the road-network step kernel (csrc/hwy_net.h: MergeEnv / MergeGenericEnv) with the reference
(golden traces, tests/golden/merge_*.npz) and with the C oracle (oracle/hwy_oracle_net.c).

Backends: ``emu`` = the kernel source run on the CPU (tests/emu, test infrastructure), ``hip`` = the shipped
libhwy_engine.so on the MI355X (``-m gpu``), called through the C-ABI.  Tolerances as in
test_engine_parity.py: 1e-9 per frame from identical state, 1e-7 over whole episodes, obs (f32) 1e-6, reward
1e-9; lane indices, target lanes, crash / impact flags, terminated: bit-exact.
"""
import numpy as np
import pytest

from highwayenv_amd import _abi, merge
from oracle import oracle
from tests.backends import BACKENDS, make_engine
from tests.golden_util import MERGE, MERGE_GRID, GoldenMerge, assert_net_state_close, assert_obs_close


def _sub(st, sel):
    return {k: np.ascontiguousarray(v[sel]) for k, v in st.items()}


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", MERGE)
def test_teacher_forced_frames_vs_reference(backend, name):
    """Each simulation frame (Road.act + Road.step) from the reference's own state; all recorded frames are
    batched into two engine calls (frames that begin a policy step get the meta-actions)."""
    g = GoldenMerge(name)
    Ef, T = g.frames_for, g.T
    K = g.steps * T
    starts, wants = [], []
    acts = np.ones((K * Ef, g.A), np.int32)
    has_act = np.zeros(K * Ef, bool)
    for k in range(K):
        starts.append(g.state("init", envs=slice(0, Ef)) if k == 0 else g.state("frame", k - 1))
        wants.append(g.state("frame", k))
        if k % T == 0:
            acts[k * Ef:(k + 1) * Ef] = g.actions[k // T, :Ef]
            has_act[k * Ef:(k + 1) * Ef] = True
    cat = lambda sts: {f: np.concatenate([s[f] for s in sts]) for f in sts[0]}  # noqa: E731
    start, want = cat(starts), cat(wants)
    for sel, with_actions in ((has_act, True), (~has_act, False)):
        idx = np.nonzero(sel)[0]
        eng = make_engine(backend, g.hwy_config(len(idx)))
        eng.set_state(_sub(start, idx))
        eng.step_frames(acts[idx] if with_actions else None, 1)
        assert_net_state_close(eng.get_state(), _sub(want, idx), atol=1e-9, what=f"{name} actions={with_actions}")
        eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", MERGE + MERGE_GRID)
def test_free_running_episodes_vs_reference(backend, name):
    """reset state -> whole episodes: obs / reward / terminated / info / state at every step while the episode
    is live and collision-free (flags, termination and reward also on the step of the first crash).  An env leaves
    the 1e-7 state comparison once one of its vehicles crawls below 1 m/s (the merging car queueing behind the
    end-of-lane Obstacle): steering divides by not_zero(speed), which amplifies the ulp-level differences a
    free-running episode has accumulated (the per-frame and per-step teacher-forced tests keep
    those frames at 1e-9)."""
    g = GoldenMerge(name)
    eng = make_engine(backend, g.hwy_config())
    eng.set_state(g.state("init"))
    np.testing.assert_allclose(eng.observe(), g.z["obs0"], rtol=0, atol=1e-6)
    live = np.ones(g.E, bool)
    compared = 0
    crawl_ok = name in ("merge_v1",)
    for t in range(g.steps):
        obs, reward, term, trunc, info = eng.step(g.actions[t])
        what = f"{name} step {t}"
        want = g.state("step", t, time=float(t + 1))
        pres = (want["flags"] & _abi.F_ABSENT) == 0
        wreck_now = (pres & ((want["flags"] & (_abi.F_CRASHED | _abi.F_HAS_IMPACT)) != 0)).any(1)
        if crawl_ok:
            moving = pres & ((want["flags"] & _abi.F_OBSTACLE) == 0)
            live = live & ~(moving & (np.abs(want["speed"]) < 1.0)).any(1)
        T_, L = live, live & ~wreck_now
        compared += int(L.sum())
        np.testing.assert_array_equal(term[T_], g.z["terminated"][t].astype(bool)[T_], err_msg=what)
        assert not trunc.any()
        np.testing.assert_array_equal(info["crashed"][T_, 0], g.z["info_crashed"][t].astype(bool)[T_], err_msg=what)
        np.testing.assert_allclose(obs[L], g.z["obs"][t][L], rtol=0, atol=1e-6, err_msg=what)
        ok = L & ((g.A == 1) | ~np.isin(g.actions[t, :, 0], [0, 2]))  # see test_oracle_golden_merge.py
        np.testing.assert_allclose(reward[ok, 0], g.z["reward"][t][ok], rtol=0, atol=1e-9, err_msg=what)
        np.testing.assert_allclose(info["speed"][L, 0], g.z["info_speed"][t][L], rtol=0, atol=1e-9, err_msg=what)
        got = eng.get_state()
        assert_net_state_close(_sub(got, L), _sub(want, L), atol=1e-7, what=what)
        live = live & ~wreck_now & ~g.z["terminated"][t].astype(bool)
        if not live.all():  # re-synchronise finished episodes from the reference (they are no longer compared)
            for k in got:
                got[k][~live] = want[k][~live]
            eng.set_state(got)
    assert compared > 0
    eng.close()


def _rollout_vs_oracle(backend, config, scenario, E, steps, seed):
    """Free-running engine vs oracle with vector-env semantics: a terminated env is re-spawned (host,
    reference stream) in both; every step of every episode is compared, the terminal one included."""
    generic = scenario == "merge-generic"
    cfg = _abi.make_config(config, E, scenario=scenario)
    st = merge.spawn_reference_stream(cfg, config, generic, np.arange(E) + 1000 * seed)
    ref = _abi.copy_state(st)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    rng = np.random.default_rng(seed)
    n_term = n_crash = 0
    next_seed = 10_000_000 * seed
    for t in range(steps):
        acts = rng.integers(0, _abi.num_actions(cfg), size=(E, cfg.num_agents)).astype(np.int32)
        obs, reward, term, trunc, info = eng.step(acts)
        o2, r2, te2, tr2, i2 = oracle.step(cfg, ref, acts)
        what = f"step {t}"
        pres = (ref["flags"] & _abi.F_ABSENT) == 0
        wreck = (pres & ((ref["flags"] & (_abi.F_CRASHED | _abi.F_HAS_IMPACT)) != 0)).any(1)
        ok = ~wreck
        np.testing.assert_array_equal(term, te2, err_msg=what)
        np.testing.assert_array_equal(trunc, tr2, err_msg=what)
        np.testing.assert_array_equal(info["crashed"], i2["crashed"], err_msg=what)
        assert_obs_close(obs[ok], o2[ok], bool(cfg.flags & _abi.C_GRID_IMAGE), what)
        np.testing.assert_allclose(reward[ok], r2[ok], rtol=0, atol=1e-9, err_msg=what)
        got = eng.get_state()
        assert_net_state_close(_sub(got, ok), _sub(ref, ok), atol=1e-7, what=what)
        n_term += int(term.sum())
        n_crash += int(i2["crashed"].any(1).sum())
        redo = term | trunc | wreck
        if redo.any():  # vector-env reset of finished (or wrecked) episodes, same fresh state in both
            k = int(redo.sum())
            fresh = merge.spawn_reference_stream(_abi.make_config(config, k, scenario=scenario), config, generic,
                                                 next_seed + np.arange(k))
            next_seed += k
            for f in ref:
                ref[f][redo] = fresh[f]
            eng.set_state(ref)
        else:
            # keep the two trajectories from drifting apart through accumulated ulps
            eng.set_state(ref)
    eng.close()
    return n_term, n_crash


@pytest.mark.parametrize("backend", BACKENDS)
def test_rollout_vs_oracle_merge_default(backend):
    n_term, n_crash = _rollout_vs_oracle(backend, merge.merge_default_config(), "merge", E=48, steps=30, seed=1)
    assert n_term > 20


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("generic", [False, True], ids=["merge-v1", "merge-generic-v1"])
def test_rollout_vs_oracle_connected_lanes(backend, generic):
    """neighbour_vehicles_connected_lanes (merge-v1 / merge-generic-v1, road.py:508-529): leaders and followers are
    also looked up across the section boundaries a->b->c->d and on the ramp j->k->b."""
    if generic:
        cfg = merge.merge_generic_default_config()
        cfg.update({"lanes_count": 3, "vehicles_count": 30, "neighbour_vehicles_connected_lanes": True})
        n_term, _ = _rollout_vs_oracle(backend, cfg, "merge-generic", E=8, steps=12, seed=5)
    else:
        cfg = merge.merge_default_config()
        cfg["neighbour_vehicles_connected_lanes"] = True
        n_term, _ = _rollout_vs_oracle(backend, cfg, "merge", E=32, steps=24, seed=6)
        assert n_term > 8


@pytest.mark.parametrize("backend", BACKENDS)
def test_rollout_vs_oracle_merge_generic_multi_agent(backend):
    """BASELINE config 5 shape: 4 lanes, 40 traffic vehicles, 4 controlled agents per environment."""
    cfg = merge.merge_generic_default_config()
    cfg.update({"lanes_count": 4, "vehicles_count": 40, "controlled_vehicles": 4,
                "action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
                "observation": {"type": "MultiAgentObservation", "observation_config": {"type": "Kinematics"}}})
    n_term, n_crash = _rollout_vs_oracle(backend, cfg, "merge-generic", E=12, steps=24, seed=2)
    assert n_term > 5
