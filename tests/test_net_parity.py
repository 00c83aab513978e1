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
from tests.golden_util import (KNIFE, MERGE, MERGE_CRASH, MERGE_GRID, GoldenMerge, assert_net_state_close, assert_obs_close,
                               mask_knife_edge_flags)


def _sub(st, sel):
    return {k: np.ascontiguousarray(v[sel]) for k, v in st.items()}


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", MERGE + MERGE_CRASH)
def test_teacher_forced_frames_vs_reference(backend, name):
    """Each simulation frame (Road.act + Road.step) from the reference's own state; all recorded frames are
    batched into two engine calls (frames that begin a policy step get the meta-actions).  Impacts are compared SIGNED
    wherever the collision is well conditioned (the C oracle's |d.normal| >= KNIFE) -- incl. the vehicle-vs-Obstacle branch
    (objects.py:101-113: the vehicle takes the whole translation), which merge_crash_obstacle exercises."""
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
    ref0_flags = start["flags"]
    n_hit = n_signed = n_obst = n_first = n_first_signed = 0
    for sel, with_actions in ((has_act, True), (~has_act, False)):
        idx = np.nonzero(sel)[0]
        cfg = g.hwy_config(len(idx))
        eng = make_engine(backend, cfg)
        eng.set_state(_sub(start, idx))
        eng.step_frames(acts[idx] if with_actions else None, 1)
        ref = _sub(start, idx)
        with oracle.impact_margins(cfg) as m:
            oracle.frames(cfg, ref, acts[idx] if with_actions else None, 1)
        w = _sub(want, idx)
        # a wreck pushed back by its impact rests EXACTLY touching what it hit: from then on `intersecting` (the partner's
        # crashed flag) is decided by the last bit (oracle.impact_margins.flag_margin == 0): on those two slots the crashed /
        # has-impact BITS are not compared (a flipped decision moves nothing by more than the ~0 distance it hinges on, so
        # positions, speeds and |impact| still are)
        calm = m.flag_margin >= KNIFE
        got = eng.get_state()
        mask_knife_edge_flags(got, w, m.flag_margin)
        hit = np.isfinite(m.margin) & ((w["flags"] & _abi.F_HAS_IMPACT) != 0)
        first = hit & ((ref0_flags[idx] & _abi.F_CRASHED) == 0)          # the frame of the FIRST contact of that vehicle
        n_hit += int(hit.sum())
        n_first += int(first.sum())
        n_signed += int((hit & calm & (m.margin >= KNIFE)).sum())
        n_first_signed += int((first & calm & (m.margin >= KNIFE)).sum())
        obst_env = (((w["flags"] & _abi.F_OBSTACLE) != 0) & np.isfinite(m.margin)).any(1)
        n_obst += int((first & calm & (m.margin >= KNIFE) & obst_env[:, None]).sum())
        assert_net_state_close(got, w, atol=1e-9, what=f"{name} actions={with_actions}", signed=m.margin >= KNIFE)
        eng.close()
    print(f"\n{name} [{backend}]: signed impact of {n_signed} / {n_hit} hit vehicle-frames compared with the reference; of the "
          f"{n_first} FIRST contacts {n_first_signed} ({n_obst} of them against the Obstacle); the rest rest on a knife edge")
    if name in MERGE_CRASH:
        assert n_first_signed >= 0.9 * n_first > 4
    if name == "merge_crash_obstacle":
        assert n_obst >= 4


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", MERGE + MERGE_GRID + MERGE_CRASH)
def test_free_running_episodes_vs_reference(backend, name):
    """reset state -> whole episodes: obs / reward / terminated / info / state at every step while the episode is live --
    INCLUDING the step of the first crash: the observation and reward returned with terminated=True, positions and SIGNED
    impacts (1e-6: the frames after the first contact resolve the wrecks' overlap again and each resolution roughly doubles a
    difference), unless the C oracle, run from the engine's own pre-step state, reports a push on the knife edge
    (|d.normal| < KNIFE, utils.py:232-236).  An env leaves the 1e-7 state comparison once one of its vehicles crawls below
    1 m/s (the merging car queueing behind the end-of-lane Obstacle): steering divides by not_zero(speed), which amplifies the
    ulp-level differences a free-running episode has accumulated (the per-frame and per-step
    teacher-forced tests keep those frames at 1e-9)."""
    g = GoldenMerge(name)
    cfg = g.hwy_config()
    eng = make_engine(backend, cfg)
    eng.set_state(g.state("init"))
    np.testing.assert_allclose(eng.observe(), g.z["obs0"], rtol=0, atol=1e-6)
    live = np.ones(g.E, bool)
    compared = n_col = n_full = 0
    crawl_ok = name in ("merge_v1",)
    for t in range(g.steps):
        before = eng.get_state()
        obs, reward, term, trunc, info = eng.step(g.actions[t])
        with oracle.impact_margins(cfg) as m:
            oracle.step(cfg, before, g.actions[t])
        what = f"{name} step {t}"
        want = g.state("step", t, time=float(t + 1))
        pres = (want["flags"] & _abi.F_ABSENT) == 0
        wreck_now = (pres & ((want["flags"] & (_abi.F_CRASHED | _abi.F_HAS_IMPACT)) != 0)).any(1)
        if crawl_ok:
            moving = pres & ((want["flags"] & _abi.F_OBSTACLE) == 0)
            live = live & ~(moving & (np.abs(want["speed"]) < 1.0)).any(1)
        col = live & wreck_now
        well = col & (m.margin.min(1) >= KNIFE)
        n_col += int(col.sum())
        n_full += int(well.sum())
        T_, L = live, live & ~wreck_now
        compared += int(L.sum())
        np.testing.assert_array_equal(term[T_], g.z["terminated"][t].astype(bool)[T_], err_msg=what)
        assert not trunc.any()
        np.testing.assert_array_equal(info["crashed"][T_, 0], g.z["info_crashed"][t].astype(bool)[T_], err_msg=what)
        got = eng.get_state()
        mask_knife_edge_flags(got, want, m.flag_margin)
        for rows, atol_state in ((L, 1e-7), (well, 1e-6)):
            np.testing.assert_allclose(obs[rows], g.z["obs"][t][rows], rtol=0, atol=1e-6, err_msg=what)
            ok = rows & ((g.A == 1) | ~np.isin(g.actions[t, :, 0], [0, 2]))  # see test_oracle_golden_merge.py
            np.testing.assert_allclose(reward[ok, 0], g.z["reward"][t][ok], rtol=0, atol=1e-9, err_msg=what)
            np.testing.assert_allclose(info["speed"][rows, 0], g.z["info_speed"][t][rows], rtol=0, atol=1e-9, err_msg=what)
            assert_net_state_close(_sub(got, rows), _sub(want, rows), atol=atol_state, what=what,
                                   signed=(m.margin >= KNIFE)[rows])
        live = live & ~wreck_now & ~g.z["terminated"][t].astype(bool)
        if not live.all():  # re-synchronise finished episodes from the reference (they are no longer compared)
            for k in got:
                got[k][~live] = want[k][~live]
            eng.set_state(got)
    assert compared > 0
    eng.close()
    print(f"\n{name} [{backend}]: {n_col} first-collision env-steps, {n_full} compared in full, "
          f"{n_col - n_full} on the knife edge (|d.normal| < {KNIFE})")
    if name in MERGE_CRASH:
        assert n_full >= 0.9 * n_col > 8


def _rollout_vs_oracle(backend, config, scenario, E, steps, seed, sync_every=4):
    """Free-running engine vs oracle with vector-env semantics: a terminated env is re-spawned (host, reference stream) in
    both; every step of every episode is compared, the terminal one included -- a step WITH a collision like any other
    (observation, reward, positions at 1e-6, SIGNED impacts) unless the oracle reports a push on the knife edge.  The engine
    runs on its OWN state for `sync_every` steps between two re-synchronisations from the oracle (an env whose slowest vehicle
    crawls below 1 m/s is re-synchronised every step: steering divides by not_zero(speed))."""
    generic = scenario == "merge-generic"
    cfg = _abi.make_config(config, E, scenario=scenario)
    st = merge.spawn_reference_stream(cfg, config, generic, np.arange(E) + 1000 * seed)
    ref = _abi.copy_state(st)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    rng = np.random.default_rng(seed)
    n_term = n_crash = n_col = n_full = n_edge = n_explained = 0
    edge_log = []
    next_seed = 10_000_000 * seed
    drift = np.zeros(E, np.int64)   # steps since the env was last synchronised
    for t in range(steps):
        acts = rng.integers(0, _abi.num_actions(cfg), size=(E, cfg.num_agents)).astype(np.int32)
        obs, reward, term, trunc, info = eng.step(acts)
        ref0 = _abi.copy_state(ref)   # (the state the step started from: a knife-edge replay below starts there again)
        with oracle.impact_margins(cfg) as m:
            o2, r2, te2, tr2, i2 = oracle.step(cfg, ref, acts)
        what = f"step {t}"
        pres = (ref["flags"] & _abi.F_ABSENT) == 0
        wreck = (pres & ((ref["flags"] & (_abi.F_CRASHED | _abi.F_HAS_IMPACT)) != 0)).any(1)
        # (an environment that has run on its own state since the last re-synchronisation may differ from the oracle's by the
        #  1e-7 its steps are held to: a push direction decided by |d.normal| below 1e-6 can flip there)
        knife = np.where(drift > 0, 1e-6, KNIFE)
        well = m.margin.min(1) >= knife
        n_col += int(wreck.sum())           # wrecked envs are re-spawned below: every wreck is a FIRST collision
        n_full += int((wreck & well).sum())
        np.testing.assert_array_equal(term, te2, err_msg=what)
        np.testing.assert_array_equal(trunc, tr2, err_msg=what)
        np.testing.assert_array_equal(info["crashed"], i2["crashed"], err_msg=what)
        got = eng.get_state()
        got_raw = _abi.copy_state(got)
        mask_knife_edge_flags(got, ref, m.flag_margin)
        # A wreck that rests EXACTLY touching a third body while another pair pushes it: whether the touching pair "will
        # intersect" hinges on a distance of ~0 (flag_margin < KNIFE), and if it does its ~0 translation REPLACES the pending
        # impact of the real collision (the last colliding pair wins, objects.py:104-112) -- the vehicle is then moved by that
        # impact in the next frame or not: a finite difference within the same policy step, either being the reference's answer
        # for one of the two roundings.  Such environments are compared like the others, but a mismatch is counted instead of
        # raised (1 in ~10^5 collision env-steps of the fuzz; they are re-spawned below, so nothing propagates).
        edge_env = wreck & (np.asarray(m.flag_margin) < KNIFE).any(1)
        for ok, atol, strict in ((~wreck, 1e-7, True), (wreck & well & ~edge_env, 1e-6, True), (wreck & well & edge_env, 1e-6, False)):
            for e in (np.flatnonzero(ok) if not strict else [None]):
                sel = ok if strict else (np.arange(E) == e)
                try:
                    assert_obs_close(obs[sel], o2[sel], bool(cfg.flags & _abi.C_GRID_IMAGE), what)
                    np.testing.assert_allclose(reward[sel], r2[sel], rtol=0, atol=1e-9, err_msg=what)
                    assert_net_state_close(_sub(got, sel), _sub(ref, sel), atol=atol, what=what, signed=(m.margin >= knife[:, None])[sel])
                except AssertionError as ex:
                    if strict:
                        raise
                    # The oracle says this env-step hinged on a distance below KNIFE.  "Either answer is the reference's" is
                    # CHECKED, not assumed: the step is replayed on the oracle from the same start with that decision rounded to
                    # either side (oracle.knife_bias), and the engine must reproduce one of the two replays in full.
                    explained = False
                    for bias in (KNIFE, -KNIFE):
                        alt = _abi.copy_state(ref0)
                        with oracle.knife_bias(bias):
                            o3, r3, _, _, _ = oracle.step(cfg, alt, acts)
                        try:
                            assert_obs_close(obs[sel], o3[sel], bool(cfg.flags & _abi.C_GRID_IMAGE), what)
                            np.testing.assert_allclose(reward[sel], r3[sel], rtol=0, atol=1e-9, err_msg=what)
                            g3, a3 = _sub(got_raw, sel), _sub(alt, sel)
                            mask_knife_edge_flags(g3, a3, m.flag_margin[sel])  # (the bits / pending impact of the touching slots)
                            assert_net_state_close(g3, a3, atol=atol, what=what)
                            explained = True
                            break
                        except AssertionError:
                            pass
                    n_explained += int(explained)
                    if not explained:
                        n_edge += 1
                        edge_log.append(f"step {t} env {e}: {str(ex).strip().splitlines()[0:3]}")
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
        # keep the two trajectories from drifting apart through accumulated ulps: every `sync_every` steps, at once for crawlers
        moving = pres & ((ref["flags"] & _abi.F_OBSTACLE) == 0)
        crawl = (moving & (np.abs(ref["speed"]) < 1.0)).any(1)
        drift += 1
        sync = redo | crawl | (drift >= sync_every)
        for f in ref:
            got[f][sync] = ref[f][sync]
        drift[sync] = 0
        eng.set_state(got)
    eng.close()
    print(f"\n{scenario} [{backend}]: {n_col} first-collision env-steps, {n_full - n_edge - n_explained} compared in full"
          + (f"; {n_explained} more equal the oracle's replay with a touching pair's distance rounded the other way" if n_explained else "")
          + (f"; {n_edge} diverged on a touching pair's knife edge and match neither replay (tolerated: at most 1)" if n_edge else ""))
    # Round 5 widened this budget to 2 after ONE chunk of 12 000 (merge fuzz chunk 1215, configuration 3: one lane, three agents
    # steered into each other, 8 first collisions in 10 steps) showed two such env-steps in one call.  Round 6 put it back to 1 and made
    # the exclusion checkable instead: a knife-edge divergence the oracle itself reproduces with the decisive distance rounded to
    # the other side (above) is an explained one and is not counted.  What remains under the budget is a step that hinges on two
    # such distances rounded to DIFFERENT sides -- 9 such env-steps of either kind in 40 000 fuzz configurations;
    # a regression in the collision path shows up as many (tests/test_mutations.py).
    assert n_edge <= 1, "knife-edge divergences must stay rare:\n" + "\n".join(edge_log)
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
