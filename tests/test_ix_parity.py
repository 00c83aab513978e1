"""Parity of the intersection step kernel (csrc/hwy_ix.h: IntersectionEnv) with the reference (golden traces,
tests/golden/intersection_*.npz) and with the C oracle (oracle/hwy_oracle_ix.c).

Backends: ``emu`` = the kernel source run on the CPU (tests/emu, test infrastructure), ``hip`` = the shipped
libhwy_engine.so on the MI355X (``-m gpu``), called through the C-ABI.  Tolerances as in test_engine_parity.py:
1e-9 per frame from identical state, obs (f32) 1e-6, reward 1e-9; lane indices, routes, yielding / crash / impact
flags, terminated / truncated: bit-exact.
"""
import numpy as np
import pytest

from highwayenv_amd import _abi
from tests.backends import BACKENDS, make_engine
from oracle import oracle
from tests.golden_util import (KNIFE, INTERSECTION_CRASH, INTERSECTION, INTERSECTION_GRID, INTERSECTION_INTENTIONS, INTERSECTION_MA, INTERSECTION_MA_FRAMES,
                               GoldenIntersection,
                               assert_ix_engine_state_close,
                               ix_engine_state)


def _hwy_config(g, E, host_traffic=True, tuning=None):
    cfg = dict(g.config)
    cfg["max_vehicles"] = g.N
    cfg["host_traffic"] = host_traffic
    return _abi.make_config(cfg, E, scenario="intersection", tuning=tuning)


def _sub(st, sel):
    return {k: np.ascontiguousarray(v[sel]) for k, v in st.items()}


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", INTERSECTION + INTERSECTION_MA_FRAMES + INTERSECTION_CRASH)
def test_teacher_forced_frames_vs_reference(backend, name):
    """Each simulation frame (meta-action on the first frame of a step, Road.act, RegulatedRoad.step incl. the
    regulation every 7th frame) from the reference's own state; all recorded frames batched into two engine calls.
    Impacts are compared SIGNED wherever the collision is well conditioned (the C oracle's |d.normal| >= KNIFE)."""
    g = GoldenIntersection(name)
    Ef, T = g.frames_for, g.T
    K = g.steps * T
    steps0 = g.z["road_steps0"][:Ef]
    envs = slice(0, Ef)
    starts, wants, acts, has_act = [], [], [], []
    for k in range(K):
        step, fr = divmod(k, T)
        if fr == 0:
            s0 = g.state("init", envs=envs) if step == 0 else g.state("next", step - 1, envs=envs)
        else:
            s0 = g.state("frame", k - 1)
        s0["road_steps"][...] = steps0 + k
        w = g.state("frame", k)
        w["road_steps"][...] = steps0 + k + 1
        starts.append(s0)
        wants.append(w)
        acts.append(g.actions[step, :Ef] if fr == 0 else np.ones((Ef, g.A), np.int32))
        has_act.append(np.full(Ef, fr == 0))
    cat = lambda sts: {f: np.concatenate([s[f] for s in sts]) for f in sts[0]}  # noqa: E731
    start, want = cat(starts), cat(wants)
    acts, has_act = np.concatenate(acts), np.concatenate(has_act)
    n_yield = n_hit = n_signed = 0
    for sel, with_actions in ((has_act, True), (~has_act, False)):
        idx = np.nonzero(sel)[0]
        cfg = _hwy_config(g, len(idx))
        eng = make_engine(backend, cfg)
        eng.set_state(ix_engine_state(g, _sub(start, idx), cfg))
        eng.step_frames(acts[idx] if with_actions else None, 1)
        w = ix_engine_state(g, _sub(want, idx), cfg)
        oc, ost = g.ix_config(len(idx)), _sub(start, idx)
        ost.pop("vid", None)
        with oracle.impact_margins(oc) as m:
            g.ix.frames(oc, ost, acts[idx] if with_actions else None, 1)
        n_hit += int(np.isfinite(m.margin).sum())
        n_signed += int((np.isfinite(m.margin) & (m.margin >= KNIFE)).sum())
        assert_ix_engine_state_close(eng.get_state(), w, atol=1e-9, what=f"{name} actions={with_actions}",
                                     signed=m.margin >= KNIFE)
        n_yield += int(((w["flags"] & _abi.F_YIELDING) != 0).sum())
        eng.close()
    assert n_yield > 0
    print(f"\n{name} [{backend}]: signed impact of {n_signed} / {n_hit} hit vehicle-frames compared with the reference "
          f"({n_hit - n_signed} on the knife edge)")
    if name in INTERSECTION_CRASH:
        assert n_signed >= 0.9 * n_hit > 10


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", INTERSECTION + INTERSECTION_GRID + INTERSECTION_MA + INTERSECTION_INTENTIONS + INTERSECTION_CRASH)
def test_policy_steps_vs_reference(backend, name):
    """Whole policy steps from the reference's state at the start of each step (host-traffic mode: the kernel does not
    clear / spawn): state, obs, reward, terminated / truncated, info -- all steps of all envs in one engine call.
    The *_grid fixtures carry BASELINE config 4's OccupancyGrid observation (on-road layer over straight lanes of any
    direction and circular arcs, world- and vehicle-aligned cells); the *intentions fixtures the cos_d / sin_d features
    (Vehicle.destination_direction) with and without observe_intentions.
    Steps WITH a collision are compared like any other -- the observation and reward returned with terminated=True, positions
    (1e-6: the frames after the first contact resolve the overlap again and each resolution doubles a difference) and SIGNED
    impacts -- unless the C oracle, run from the same state, reports a collision on the knife edge (|d.normal| < KNIFE)."""
    g = GoldenIntersection(name)
    E, S = g.E, g.steps
    steps0 = g.z["road_steps0"]
    starts = []
    for t in range(S):
        s0 = g.state("init") if t == 0 else g.state("next", t - 1)
        s0["road_steps"][...] = steps0 + t * g.T
        s0["time"][...] = float(t)
        starts.append(s0)
    cat = lambda sts: {f: np.concatenate([s[f] for s in sts]) for f in sts[0]}  # noqa: E731
    cfg = _hwy_config(g, E * S)
    eng = make_engine(backend, cfg)
    eng.set_state(ix_engine_state(g, cat(starts), cfg))
    obs, reward, term, trunc, info = eng.step(g.actions.reshape(E * S, g.A))
    got = eng.get_state()
    oc, ost = g.ix_config(E * S), cat(starts)
    ost.pop("vid", None)
    with oracle.impact_margins(oc) as m:
        g.ix.step(oc, ost, g.actions.reshape(E * S, g.A))
    live = np.ones(E, bool)
    n_wreck = n_full = 0
    for t in range(S):
        rows = slice(t * E, (t + 1) * E)
        want = g.state("step", t)
        want["road_steps"][...] = steps0 + (t + 1) * g.T
        want["time"][...] = float(t + 1)
        wreck = ((want["present"] != 0) & ((want["crashed"] != 0) | (want["has_impact"] != 0))).any(1)
        well = m.margin[rows] >= KNIFE
        clean = live & well.all(1)
        n_wreck += int((live & wreck).sum())
        n_full += int((clean & wreck).sum())
        what = f"{name} step {t}"
        for sel, atol in ((clean & ~wreck, 1e-8), (clean & wreck, 1e-6)):
            sub_cfg = _hwy_config(g, int(sel.sum()))
            signed = np.zeros((int(sel.sum()), sub_cfg.num_vehicles), bool)
            signed[:, :g.N] = well[sel]
            start = ix_engine_state(g, _sub(_sub(cat(starts), rows), sel), sub_cfg)["speed"]
            # (1e-8 after 15 free-running frames for cars that drive; 1e-6 for a car below 2 m/s at either end of the step -- the
            #  rule of tests/test_fuzz_configs.py: on the GPU's fused multiply-adds one yielding car of intersection_multi_agent3
            #  ends step 7 4.8e-8 from the reference's numpy double-rounded trace, 1e-9 per frame from the same state)
            assert_ix_engine_state_close(_sub(_sub(got, rows), sel), ix_engine_state(g, _sub(want, sel), sub_cfg), atol=atol,
                                         what=what, signed=signed, slow_atol=1e-6, slow_start=start)
        np.testing.assert_array_equal(term[rows][live], g.z["terminated"][t].astype(bool)[live], err_msg=what)
        np.testing.assert_array_equal(trunc[rows][live], g.z["truncated"][t].astype(bool)[live], err_msg=what)
        np.testing.assert_array_equal(info["crashed"][rows][live, 0], g.z["info_crashed"][t].astype(bool)[live], err_msg=what)
        np.testing.assert_allclose(obs[rows][clean], g.z["obs"][t][clean].reshape(obs[rows][clean].shape), rtol=0, atol=1e-6,
                                   err_msg=what)
        if g.A == 1:
            np.testing.assert_allclose(reward[rows][clean, 0], g.z["reward"][t][clean], rtol=0, atol=1e-9, err_msg=what)
        else:  # MultiAgentIntersectionEnv: per-agent rewards, "crashed or arrived" per agent (intersection_env.py:114-122)
            np.testing.assert_allclose(reward[rows][clean], g.z["agents_rewards"][t][clean], rtol=0, atol=1e-9, err_msg=what)
            np.testing.assert_allclose(reward[rows][clean].sum(1) / g.A, g.z["reward"][t][clean], rtol=0, atol=1e-9, err_msg=what)
            np.testing.assert_array_equal((info["crashed"] | info["arrived"])[rows][live],
                                          g.z["agents_terminated"][t].astype(bool)[live], err_msg=what)
        np.testing.assert_allclose(info["speed"][rows][clean, 0], g.z["info_speed"][t][clean], rtol=0, atol=1e-9, err_msg=what)
        live &= ~g.z["terminated"][t].astype(bool)
    eng.close()
    print(f"\n{name} [{backend}]: {n_wreck} env-steps with a wreck on the road, {n_full} compared in full")
    if name in INTERSECTION_CRASH:
        assert n_full >= 0.9 * n_wreck > 0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", ["intersection_default", "intersection_dense"])
def test_helper_lanes_are_bit_identical_to_the_serial_loops(backend, name):
    """N <= 32: threads 32..63 of the wavefront split the lane-table walk, the collision partners and the regulation
    samples / partners with the vehicle's own thread (hwy_ix.h, IxSharedT).  Same arithmetic, same tie rules: the state
    after whole policy steps (device clear / spawn included) must equal the 32-thread build's (hwy_config.tune_ix_no_helpers) BIT FOR BIT."""
    g = GoldenIntersection(name)
    E = g.E
    rng = np.random.default_rng(3)
    acts = rng.integers(0, 3, size=(6, E, 1)).astype(np.int32)
    out = []
    for no_helpers in (0, 1):
        cfg = _hwy_config(g, E, host_traffic=False, tuning={"ix_no_helpers": no_helpers})
        eng = make_engine(backend, cfg)
        eng.set_state(ix_engine_state(g, g.state("init"), cfg))
        rows = []
        for t in range(acts.shape[0]):
            obs, reward, term, trunc, info = eng.step(acts[t])
            st = eng.get_state()
            rows.append((obs.copy(), reward.copy(), term.copy(), {k: v.copy() for k, v in st.items()}))
        out.append(rows)
        eng.close()
    for (o1, r1, t1, s1), (o0, r0, t0, s0) in zip(*out):
        np.testing.assert_array_equal(o1, o0)
        np.testing.assert_array_equal(r1, r0)
        np.testing.assert_array_equal(t1, t0)
        for k in s1:
            np.testing.assert_array_equal(s1[k], s0[k], err_msg=k)
