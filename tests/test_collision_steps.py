"""Collision steps compared IN FULL, and the collision scan under speeds nothing clamps.

Round-1 parity tests compared flags / termination / reward on the step in which a collision happens but dropped
positions and observations there, because the reference orients the collision push with ``d.dot(normal) > 0``
(highway_env/utils.py:232-236) and for two cars on one lane centre that dot product is rounding noise.  That argument
only covers collisions that ARE on the knife edge.  Here the C oracle reports, per vehicle, the smallest |d.normal| among
the impacts it assigned (``oracle.impact_margins``), and every collision step whose margins are all >= 1e-9 is compared
like any other step: the observation returned WITH ``terminated=True`` (what a user actually sees), positions, signed
impacts.  The sign of an impact may differ from the reference's ONLY where the margin is below 1e-9.

Backends as in test_engine_parity.py (emu = kernel source on the CPU, hip = the product on the MI355X).
"""
import numpy as np
import pytest

from highwayenv_amd import _abi, merge, spawn
from oracle import oracle
from tests.backends import BACKENDS, make_engine
from tests.golden_util import ALL, Golden, assert_state_close

KNIFE = 1e-9
CRASH_GOLDENS = ["crash_many_fast", "crash_many_v0"]


def _signed_impacts_equal(got, want, rows, atol=1e-9):
    return (np.abs(got["impact_x"][rows] - want["impact_x"][rows]) <= atol) & \
           (np.abs(got["impact_y"][rows] - want["impact_y"][rows]) <= atol)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", ALL + CRASH_GOLDENS)
def test_collision_steps_vs_reference(backend, name):
    """Free-running episodes against the reference's traces.  On the step in which an episode's first collision happens
    the terminal observation, reward, positions and SIGNED impacts must match the reference whenever the collision is
    well conditioned (all margins >= 1e-9); knife-edge collisions keep the round-1 comparison (flags, |impact|)."""
    g = Golden(name)
    cfg = _abi.make_config(g.config, g.E, fast=g.fast)
    eng = make_engine(backend, cfg)
    eng.set_state(g.state("init"))
    live = np.ones(g.E, bool)
    n_collision = n_full = n_knife = 0
    for t in range(g.steps):
        before = eng.get_state()
        obs, reward, term, trunc, info = eng.step(g.actions[t])
        want = g.state("step", t, time=float(t + 1))
        ref = _abi.copy_state(before)
        with oracle.impact_margins(cfg) as m:
            oracle.step(cfg, ref, g.actions[t])
        wreck_now = ((want["flags"] & (_abi.F_CRASHED | _abi.F_HAS_IMPACT)) != 0).any(1)
        col = live & wreck_now
        well = col & (m.margin.min(1) >= KNIFE)
        n_collision += int(col.sum())
        n_full += int(well.sum())
        n_knife += int((col & ~well).sum())
        got = eng.get_state()
        what = f"{name} step {t} (collision steps)"
        np.testing.assert_array_equal(term[col], g.z["terminated"][t].astype(bool)[col], err_msg=what)
        np.testing.assert_array_equal(info["crashed"][col, 0], g.z["info_crashed"][t].astype(bool)[col], err_msg=what)
        if well.any():
            np.testing.assert_allclose(obs[well, 0], g.z["obs"][t][well], rtol=0, atol=1e-6, err_msg=what)
            np.testing.assert_allclose(reward[well, 0], g.z["reward"][t][well], rtol=0, atol=1e-9, err_msg=what)
            np.testing.assert_allclose(info["speed"][well, 0], g.z["info_speed"][t][well], rtol=0, atol=1e-9, err_msg=what)
            # 1e-6: the frames after the first contact resolve the wrecks' overlap again and again (differences double)
            assert_state_close({k: v[well] for k, v in got.items()}, {k: v[well] for k, v in want.items()}, atol=1e-6, what=what)
            assert _signed_impacts_equal(got, want, well, 1e-6).all(), what + ": impact sign"
        live = live & ~wreck_now
        for k in got:
            got[k][~live] = want[k][~live]
        eng.set_state(got)
    eng.close()
    print(f"\n{name} [{backend}]: {n_collision} first-collision env-steps, {n_full} compared in full, "
          f"{n_knife} on the knife edge (|d.normal| < {KNIFE})")
    if name in CRASH_GOLDENS:
        assert n_full >= 10, "the crash-rich fixtures must exercise the full comparison"


@pytest.mark.parametrize("backend", BACKENDS)
def test_impact_sign_agreement_rate(backend):
    """Every recorded FRAME of the crash-rich traces, teacher-forced: the signed impact of every vehicle that was hit in
    that frame, engine vs reference.  Prints the agreement rate; a disagreement is allowed only on the knife edge."""
    total = agree = knife = 0
    for name in ("dense_crash", "crash_many_v0"):
        g = Golden(name)
        Ef, T = g.frames_for, g.T
        K = g.steps * T
        for with_actions in (True, False):
            ks = [k for k in range(K) if (k % T == 0) == with_actions]
            if backend == "emu":
                # the CPU emulation replays the frames of a FIRST contact (a vehicle is hit that was not a wreck before) and
                # every eighth of the frames in which the wrecks keep colliding (the MI355X run replays all frames)
                hit = [k for k in ks if (g.state("frame", k)["flags"] & _abi.F_HAS_IMPACT).any()]

                def first_contact(k):
                    now = g.state("frame", k)["flags"]
                    before = (g.state("frame", k - 1) if k else g.state("init", envs=slice(0, Ef)))["flags"]
                    return (((now & _abi.F_HAS_IMPACT) != 0) & ((before & _abi.F_CRASHED) == 0)).any()
                ks = sorted(set([k for k in hit if first_contact(k)] + hit[::8]))
            if not ks:
                continue
            start = {f: np.concatenate([(g.state("init", envs=slice(0, Ef)) if k == 0 else g.state("frame", k - 1))[f]
                                        for k in ks]) for f in _abi.STATE_F64 + _abi.STATE_I32 + ["time"]}
            want = {f: np.concatenate([g.state("frame", k)[f] for k in ks]) for f in start}
            acts = np.concatenate([g.actions[k // T, :Ef] for k in ks]).reshape(-1, 1).astype(np.int32)
            cfg = _abi.make_config(g.config, len(ks) * Ef, fast=g.fast)
            eng = make_engine(backend, cfg)
            eng.set_state(start)
            eng.step_frames(acts if with_actions else None, 1)
            got = eng.get_state()
            eng.close()
            ref = _abi.copy_state(start)
            with oracle.impact_margins(cfg) as m:
                oracle.frames(cfg, ref, acts if with_actions else None, 1)
            hit = (want["flags"] & _abi.F_HAS_IMPACT) != 0
            np.testing.assert_array_equal((got["flags"] & _abi.F_HAS_IMPACT) != 0, hit)
            same = _signed_impacts_equal(got, want, hit)
            mg = m.margin[hit]
            total += int(hit.sum())
            agree += int(same.sum())
            knife += int((mg < KNIFE).sum())
            assert (same | (mg < KNIFE)).all(), f"{name}: impact sign differs on a well-conditioned collision"
    print(f"\nimpact sign agreement [{backend}]: {agree} / {total} hit vehicle-frames agree with the reference "
          f"({100.0 * agree / max(total, 1):.1f} %), {knife} of them on the knife edge")
    assert total > 20


def _inject_fast_bodies(st, rng, n_vehicles, speeds, lanes_y=None):
    """Give a few vehicles per env an extreme speed and put each `gap` metres behind another vehicle ON ITS LANE, so
    that it reaches it within one frame although the two start far apart."""
    E = st["x"].shape[0]
    n_hits = 0
    for e in range(E):
        pres = np.nonzero((st["flags"][e] & (_abi.F_ABSENT | _abi.F_OBSTACLE)) == 0)[0]
        pres = pres[pres < n_vehicles]
        for v in rng.choice(speeds, size=2, replace=False):
            a, b = rng.choice(pres[1:], size=2, replace=False)
            st["speed"][e, a] = v
            st["heading"][e, a] = 0.0
            st["x"][e, a] = st["x"][e, b] - rng.uniform(0.55, 1.0) * v * (1 / 15) - 6.0
            st["y"][e, a] = st["y"][e, b]
            st["lane"][e, a] = st["target_lane"][e, a] = st["lane"][e, b]
            n_hits += 1
    return n_hits


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("kernel", ["wave", "block", "two_waves"])
def test_fast_bodies_are_not_missed_by_the_bounded_scan(backend, kernel):
    """Road.step tests ALL pairs (road.py:477-481) and nothing clamps a speed (clip_actions only pulls it back,
    kinematics.py:155-168; hwy_set_state accepts any).  The kernels walk forward in rank order up to a reach derived from the
    frame's ACTUAL largest displacement and speed (hwy_device.h: reach_from_keys; rounds 2-5: from 50 m/s, with an all-pairs
    fallback): bodies at 60..600 m/s that close 8..46 m within one frame must be found.  Flags exact vs the oracle
    (tests/test_mutations.py: a reach that forgets the frame's motion fails here)."""
    cfg_d = _abi.highway_default_config()
    cfg_d.update({"vehicles_count": 100 if kernel == "two_waves" else 40, "lanes_count": 4})
    if kernel == "block":
        cfg_d["tuning"] = {"block_kernel": 1}
    E = (2 if kernel == "two_waves" else 4) if backend == "emu" else 64
    cfg = _abi.make_config(cfg_d, E, fast=False)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 31, cfg_d["ego_spacing"], cfg_d["vehicles_density"], None)
    rng = np.random.default_rng(8)
    _inject_fast_bodies(st, rng, cfg.num_vehicles, [60.0, 80.0, 150.0, 300.0, 600.0])
    ref = _abi.copy_state(st)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    far_hits = 0
    for fr in range(3):
        x0 = ref["x"].copy()
        eng.step_frames(None, 1)
        with oracle.impact_margins(cfg) as m:
            oracle.frames(cfg, ref, None, 1)
        got = eng.get_state()
        for k in ("flags", "lane", "target_lane"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"frame {fr}: {k}")
        calm = m.margin.min(1) >= KNIFE
        assert_state_close({k: v[calm] for k, v in got.items()}, {k: v[calm] for k, v in ref.items()}, atol=1e-6,
                           what=f"frame {fr}")
        # a hit between two bodies that started the frame further apart than the 50 m/s reach: the old bound's blind spot
        hit = (ref["flags"] & _abi.F_HAS_IMPACT) != 0
        reach = (5.5 + 50.0 * cfg.dt) + 2.0 * (50.0 * cfg.dt + 3.0)
        for e in range(E):
            idx = np.nonzero(hit[e])[0]
            if len(idx) >= 2 and (np.abs(x0[e, idx][:, None] - x0[e, idx][None, :]).max() > reach):
                far_hits += 1
    eng.close()
    assert far_hits > 0, "the scenario must contain collisions beyond the 50 m/s reach"


@pytest.mark.parametrize("backend", BACKENDS)
def test_fast_bodies_on_the_merge_network(backend):
    """The same for hwy_net_step_kernel (two-tier reach, 36 / 50 m/s): merge-generic traffic with bodies at 60..600 m/s."""
    cfg_d = merge.merge_generic_default_config()
    cfg_d.update({"lanes_count": 3, "vehicles_count": 25})
    E = 3 if backend == "emu" else 64
    cfg = _abi.make_config(cfg_d, E, scenario="merge-generic")
    st = merge.spawn_reference_stream(cfg, cfg_d, True, np.arange(E) + 5)
    rng = np.random.default_rng(9)
    _inject_fast_bodies(st, rng, cfg.num_vehicles - 2, [60.0, 80.0, 150.0, 300.0, 600.0])
    ref = _abi.copy_state(st)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    n_hit = 0
    for fr in range(3):
        eng.step_frames(None, 1)
        oracle.frames(cfg, ref, None, 1)
        got = eng.get_state()
        pres = (ref["flags"] & _abi.F_ABSENT) == 0
        np.testing.assert_array_equal(got["flags"][pres], ref["flags"][pres], err_msg=f"frame {fr}: flags")
        np.testing.assert_array_equal(got["lane"][pres], ref["lane"][pres], err_msg=f"frame {fr}: lane")
        n_hit += int(((ref["flags"] & _abi.F_HAS_IMPACT) != 0)[pres].sum())
    eng.close()
    assert n_hit > 0


# ---- the road-network families (round 3): merge (incl. vehicle-vs-Obstacle) and intersection -----------------------------
# The whole-step comparisons live next to the other parity tests of each family (test_net_parity.py::
# test_free_running_episodes_vs_reference / _rollout_vs_oracle, test_ix_parity.py::test_policy_steps_vs_reference: every
# collision step in full unless a push is on the knife edge); here: the per-FRAME signed-impact agreement rate with the
# reference's recorded frames, one number per family, like test_impact_sign_agreement_rate above.

def _cat(sts):
    return {f: np.concatenate([s[f] for s in sts]) for f in sts[0]}


def _sub(st, sel):
    return {k: np.ascontiguousarray(v[sel]) for k, v in st.items()}


@pytest.mark.parametrize("backend", BACKENDS)
def test_impact_sign_agreement_rate_merge(backend):
    """Every recorded frame of the crash-rich merge traces, teacher-forced from the reference's own state: the SIGNED impact of
    every vehicle hit in that frame, engine vs reference (objects.py:101-113: halves for two vehicles, the whole translation
    for a vehicle against the Obstacle).  A disagreement is allowed only on the knife edge (|d.normal| < KNIFE)."""
    from tests.golden_util import MERGE_CRASH, GoldenMerge
    total = agree = knife = vs_obstacle = 0
    for name in MERGE_CRASH:
        g = GoldenMerge(name)
        Ef, T = g.frames_for, g.T
        for with_actions in (True, False):
            ks = [k for k in range(g.steps * T) if (k % T == 0) == with_actions]
            ks = [k for k in ks if g.z["frame_has_impact"][k].any()]   # frames in which the reference recorded a hit
            if not ks:
                continue
            start = _cat([g.state("init", envs=slice(0, Ef)) if k == 0 else g.state("frame", k - 1) for k in ks])
            want = _cat([g.state("frame", k) for k in ks])
            acts = np.concatenate([g.actions[k // T, :Ef] for k in ks]).astype(np.int32)
            cfg = g.hwy_config(len(ks) * Ef)
            eng = make_engine(backend, cfg)
            eng.set_state(start)
            eng.step_frames(acts if with_actions else None, 1)
            got = eng.get_state()
            eng.close()
            ref = _abi.copy_state(start)
            with oracle.impact_margins(cfg) as m:
                oracle.frames(cfg, ref, acts if with_actions else None, 1)
            hit = (want["flags"] & _abi.F_HAS_IMPACT) != 0
            np.testing.assert_array_equal((got["flags"] & _abi.F_HAS_IMPACT) != 0, hit, err_msg=name)
            same = _signed_impacts_equal(got, want, hit)
            mg = m.margin[hit]
            total += int(hit.sum())
            agree += int(same.sum())
            knife += int((mg < KNIFE).sum())
            # a vehicle whose impact is the WHOLE translation: its partner was the Obstacle (which never gets one)
            obst = ((want["flags"] & _abi.F_OBSTACLE) != 0) & np.isfinite(m.margin)
            vs_obstacle += int((hit & obst.any(1)[:, None]).sum())
            assert (same | (mg < KNIFE)).all(), f"{name}: impact sign differs on a well-conditioned collision"
    print(f"\nimpact sign agreement, merge family [{backend}]: {agree} / {total} hit vehicle-frames agree with the reference "
          f"({100.0 * agree / max(total, 1):.1f} %), {knife} on the knife edge, {vs_obstacle} in frames where the Obstacle was hit")
    assert total > 100 and vs_obstacle > 20


@pytest.mark.parametrize("backend", BACKENDS)
def test_impact_sign_agreement_rate_intersection(backend):
    """The same for the intersection kernel on the crash-rich IntersectionEnv / MultiAgentIntersectionEnv traces."""
    from tests.golden_util import INTERSECTION_CRASH, GoldenIntersection, ix_engine_state
    total = agree = knife = 0
    for name in INTERSECTION_CRASH + ["intersection_dense"]:
        g = GoldenIntersection(name)
        Ef, T = g.frames_for, g.T
        steps0 = g.z["road_steps0"][:Ef]
        envs = slice(0, Ef)
        for with_actions in (True, False):
            ks = [k for k in range(g.steps * T) if (k % T == 0) == with_actions and g.z["frame_has_impact"][k].any()]
            if not ks:
                continue
            starts, wants = [], []
            for k in ks:
                step, fr = divmod(k, T)
                s0 = (g.state("init", envs=envs) if step == 0 else g.state("next", step - 1, envs=envs)) if fr == 0 \
                    else g.state("frame", k - 1)
                s0["road_steps"][...] = steps0 + k
                w = g.state("frame", k)
                w["road_steps"][...] = steps0 + k + 1
                starts.append(s0)
                wants.append(w)
            start, want = _cat(starts), _cat(wants)
            acts = np.concatenate([g.actions[k // T, :Ef] for k in ks]).astype(np.int32)
            cfg_d = dict(g.config, max_vehicles=g.N, host_traffic=True)
            cfg = _abi.make_config(cfg_d, len(ks) * Ef, scenario="intersection")
            eng = make_engine(backend, cfg)
            eng.set_state(ix_engine_state(g, start, cfg))
            eng.step_frames(acts if with_actions else None, 1)
            got = eng.get_state()
            eng.close()
            oc = g.ix_config(len(ks) * Ef)
            ost = {k_: v for k_, v in start.items() if k_ != "vid"}
            with oracle.impact_margins(oc) as m:
                g.ix.frames(oc, ost, acts if with_actions else None, 1)
            w = ix_engine_state(g, want, cfg)
            hit = (w["flags"] & _abi.F_HAS_IMPACT) != 0
            np.testing.assert_array_equal(((got["flags"] & _abi.F_HAS_IMPACT) != 0) & ((w["flags"] & _abi.F_ABSENT) == 0), hit,
                                          err_msg=name)
            same = _signed_impacts_equal(got, w, hit)
            mg = np.full(hit.shape, np.inf)
            mg[:, :g.N] = m.margin
            total += int(hit.sum())
            agree += int(same.sum())
            knife += int((mg[hit] < KNIFE).sum())
            assert (same | (mg[hit] < KNIFE)).all(), f"{name}: impact sign differs on a well-conditioned collision"
    print(f"\nimpact sign agreement, intersection family [{backend}]: {agree} / {total} hit vehicle-frames agree with the "
          f"reference ({100.0 * agree / max(total, 1):.1f} %), {knife} on the knife edge")
    assert total > 100
