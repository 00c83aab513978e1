"""Parity of the step kernel with the reference (golden traces) and with the C oracle.

Backends: ``emu`` = the kernel source run on the CPU (tests/emu, test infrastructure),
``hip`` = the shipped libhwy_engine.so on the MI355X (``-m gpu``), called through the C-ABI.

Tolerances (f64 state): 1e-9 absolute on positions/speeds/headings/timers/impacts after one
frame from identical state, 1e-7 after whole episodes (libm ulps accumulate; ocml != glibc !=
numpy); obs (f32) 1e-6; reward 1e-9; lane indices, target lanes, crash/impact flags, terminated,
truncated: bit-exact.  north_star asks for 1e-5.
"""
import numpy as np
import pytest

from highwayenv_amd import _abi, spawn
from oracle import oracle
from tests.backends import BACKENDS, make_engine
from tests.golden_util import ALL, WITH_FRAMES, Golden, assert_state_close


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", WITH_FRAMES)
def test_teacher_forced_frames_vs_reference(backend, name):
    """Each simulation frame (Road.act + Road.step) from the reference's own state.
    All frames of all recorded envs are batched into ONE engine call per frame index."""
    g = Golden(name)
    Ef, T = g.frames_for, g.T
    K = g.steps * T
    # env b of the batch = (frame k, recorded env e): start state = state before frame k
    starts, wants, acts = [], [], np.ones((K * Ef, 1), np.int32)
    has_act = np.zeros(K * Ef, bool)
    for k in range(K):
        starts.append(g.state("init", envs=slice(0, Ef)) if k == 0 else g.state("frame", k - 1))
        wants.append(g.state("frame", k))
        if k % T == 0:
            acts[k * Ef:(k + 1) * Ef, 0] = g.actions[k // T, :Ef]
            has_act[k * Ef:(k + 1) * Ef] = True
    cat = lambda sts: {f: np.concatenate([s[f] for s in sts]) for f in sts[0]}
    start, want = cat(starts), cat(wants)
    # two launches: frames that begin a policy step (meta-action applied) and the others
    for sel, with_actions in ((has_act, True), (~has_act, False)):
        idx = np.nonzero(sel)[0]
        cfg = _abi.make_config(g.config, len(idx), fast=g.fast)
        eng = make_engine(backend, cfg)
        eng.set_state({f: np.ascontiguousarray(v[idx]) for f, v in start.items()})
        eng.step_frames(acts[idx] if with_actions else None, 1)
        got = eng.get_state()
        assert_state_close(got, {f: v[idx] for f, v in want.items()}, atol=1e-9, what=f"{name} actions={with_actions}")
        eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", ALL)
def test_free_running_episodes_vs_reference(backend, name):
    """reset state -> whole episodes: obs/reward/terminated/truncated/info/state every step, for as long
    as the episode is live (up to and INCLUDING the step in which the first crash happens).  The golden
    traces keep stepping after `terminated`; from then on a wreck rests against the cars it hit and
    the reference itself is ill-conditioned at the ulp level (test_ulp_noise_only_moves_post_termination_wrecks),
    so those rows are re-synchronised from the reference instead of compared."""
    g = Golden(name)
    cfg = _abi.make_config(g.config, g.E, fast=g.fast)
    eng = make_engine(backend, cfg)
    eng.set_state(g.state("init"))
    np.testing.assert_allclose(eng.observe()[:, 0], g.z["obs0"], rtol=0, atol=1e-6)
    live = np.ones(g.E, bool)
    compared = 0
    for t in range(g.steps):
        obs, reward, term, trunc, info = eng.step(g.actions[t])
        what = f"{name} step {t}"
        want = g.state("step", t, time=float(t + 1))
        wreck_now = ((want["flags"] & (_abi.F_CRASHED | _abi.F_HAS_IMPACT)) != 0).any(1)
        T_ = live            # episodes live at the start of the step: flags / termination / reward
        L = live & ~wreck_now  # ... and collision-free during it: everything (see _random_rollout_vs_oracle)
        compared += int(L.sum())
        np.testing.assert_array_equal(term[T_], g.z["terminated"][t].astype(bool)[T_], err_msg=what)
        np.testing.assert_array_equal(trunc, g.z["truncated"][t].astype(bool), err_msg=what)
        np.testing.assert_array_equal(info["crashed"][T_, 0], g.z["info_crashed"][t].astype(bool)[T_], err_msg=what)
        np.testing.assert_allclose(reward[T_, 0], g.z["reward"][t][T_], rtol=0, atol=1e-6, err_msg=what)
        np.testing.assert_allclose(obs[L, 0], g.z["obs"][t][L], rtol=0, atol=1e-6, err_msg=what)
        np.testing.assert_allclose(reward[L, 0], g.z["reward"][t][L], rtol=0, atol=1e-9, err_msg=what)
        np.testing.assert_allclose(info["speed"][L, 0], g.z["info_speed"][t][L], rtol=0, atol=1e-9, err_msg=what)
        got = eng.get_state()
        assert_state_close({k: v[L] for k, v in got.items()}, {k: v[L] for k, v in want.items()}, atol=1e-7, what=what)
        live = live & ~wreck_now
        if not live.all():
            for k in got:
                got[k][~live] = want[k][~live]
            eng.set_state(got)
    assert compared > 0
    eng.close()


def _random_rollout_vs_oracle(backend, config, fast, E, steps, seed):
    """Free-running engine vs oracle with vector-env semantics: an env that terminates/truncates is
    re-spawned (host, reference stream) in both.  Every step of every episode is compared, the
    terminal step included.  (What is NOT compared is a wreck left to rot after `terminated`: there
    the reference's own contact dynamics are ill-conditioned at the ulp level -- see
    test_ulp_noise_only_moves_post_termination_wrecks.)"""
    cfg = _abi.make_config(config, E, fast=fast)
    spawn_args = (config["ego_spacing"], config["vehicles_density"], config["initial_lane_id"])
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 1000 * seed, *spawn_args)
    ref = _abi.copy_state(st)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    rng = np.random.default_rng(seed)
    n_term = n_trunc = n_crashed_vehicles = n_collision_steps = n_collision_full = 0
    next_seed = 10_000_000 * seed
    for t in range(steps):
        acts = rng.integers(0, 5, size=(E, cfg.num_agents)).astype(np.int32)
        obs, reward, term, trunc, info = eng.step(acts)
        with oracle.impact_margins(cfg) as margins:
            o2, r2, te2, tr2, i2 = oracle.step(cfg, ref, acts)
        what = f"step {t}"
        # Envs in which a collision happened during this step are compared IN FULL (terminal observation, positions,
        # signed impacts) unless the collision sits on the knife edge: when two cars collide while tracking the same
        # lane centre, the sign of the reference's minimum-translation vector is decided by rounding noise in
        # d.normal ~ 1e-16 (utils.py:232-236) and the 1 m lateral push it encodes can differ between two libm's.
        # There (oracle.impact_margins: |d.normal| < 1e-9) flags / termination / reward are compared, positions are not.
        wreck = ((ref["flags"] & (_abi.F_CRASHED | _abi.F_HAS_IMPACT)) != 0).any(1)
        ok = ~wreck | (margins.margin.min(1) >= 1e-9)
        n_collision_steps += int(wreck.sum())
        n_collision_full += int((wreck & ok).sum())
        np.testing.assert_array_equal(term, te2, err_msg=what)
        np.testing.assert_array_equal(trunc, tr2, err_msg=what)
        np.testing.assert_array_equal(info["crashed"], i2["crashed"], err_msg=what)
        np.testing.assert_allclose(reward, r2, rtol=0, atol=1e-6, err_msg=what)
        np.testing.assert_allclose(obs[ok], o2[ok], rtol=0, atol=1e-6, err_msg=what)
        np.testing.assert_allclose(reward[ok], r2[ok], rtol=0, atol=1e-9, err_msg=what)
        np.testing.assert_allclose(info["speed"][ok], i2["speed"][ok], rtol=0, atol=1e-9, err_msg=what)
        got = eng.get_state()
        # (collision steps at 1e-6: every frame after the first contact resolves the overlap of the two wrecks again, and
        # each resolution roughly doubles a difference -- 1e-8 becomes a few 1e-7 within the step, measured on the GPU)
        clean, hit = ok & ~wreck, ok & wreck
        assert_state_close({k: v[clean] for k, v in got.items()}, {k: v[clean] for k, v in ref.items()}, atol=1e-7, what=what)
        assert_state_close({k: v[hit] for k, v in got.items()}, {k: v[hit] for k, v in ref.items()}, atol=1e-6, what=what + " (collision)")
        for k in ("impact_x", "impact_y"):  # signed, where the push direction is well conditioned
            np.testing.assert_allclose(got[k][ok], ref[k][ok], rtol=0, atol=1e-6, err_msg=f"{what}: {k} (signed)")
        np.testing.assert_array_equal((got["flags"] & _abi.F_CRASHED)[:, 0], (ref["flags"] & _abi.F_CRASHED)[:, 0], err_msg=what)
        # an env holding a wreck (possible without `terminated` when the crash does not involve agent 0:
        # IDM-IDM pile-ups in highway-v0, secondary agents) is retired too: resting contact is the one
        # regime where the reference itself is ill-conditioned (see the ulp-noise test below)
        done = term | trunc | wreck
        n_term += int(term.sum())
        n_trunc += int(trunc.sum())
        n_crashed_vehicles += int(((ref["flags"][done] & _abi.F_CRASHED) != 0).sum())
        if done.any():
            idx = np.nonzero(done)[0]
            sub = _abi.make_config(config, len(idx), fast=fast)
            fresh = spawn.spawn_reference_stream(sub, next_seed + np.arange(len(idx)), *spawn_args)
            next_seed += len(idx)
            for k in fresh:
                got[k][idx] = fresh[k]
                ref[k][idx] = fresh[k]
            eng.set_state(got)
    eng.close()
    return {"terminated": n_term, "truncated": n_trunc, "crashed_vehicles": n_crashed_vehicles,
            "collision_steps": n_collision_steps, "collision_steps_compared_in_full": n_collision_full}


@pytest.mark.parametrize("backend", BACKENDS)
def test_random_rollout_vs_oracle_fast(backend):
    """highway-fast-v0 semantics, N=51 x 4 lanes (per-env workload of BASELINE config 2)."""
    cfg = _abi.highway_fast_default_config()
    cfg.update({"vehicles_count": 50, "lanes_count": 4})
    E = 8 if backend == "emu" else 512
    stats = _random_rollout_vs_oracle(backend, cfg, True, E, 8 if backend == "emu" else 40, seed=1)
    if backend == "hip":
        assert stats["terminated"] > 50 and stats["truncated"] > 0  # crashes, resets and time limits exercised
        assert stats["collision_steps_compared_in_full"] > 25, stats  # terminal observations really compared


@pytest.mark.parametrize("backend", BACKENDS)
def test_random_rollout_vs_oracle_v0(backend):
    """highway-v0 semantics: 15 Hz, full pairwise collisions; N=101 => two wavefronts per env."""
    cfg = _abi.highway_default_config()
    cfg.update({"vehicles_count": 100})
    E = 2 if backend == "emu" else 64
    _random_rollout_vs_oracle(backend, cfg, False, E, 2 if backend == "emu" else 10, seed=2)


@pytest.mark.parametrize("backend", BACKENDS)
def test_multi_agent_vs_oracle(backend):
    """controlled_vehicles=3 (MultiAgentAction/MultiAgentObservation layout: A obs blocks per env)."""
    cfg = _abi.highway_default_config()
    cfg.update({"vehicles_count": 20, "controlled_vehicles": 3, "lanes_count": 3, "simulation_frequency": 5,
                "duration": 15})
    _random_rollout_vs_oracle(backend, cfg, False, 4 if backend == "emu" else 128, 6 if backend == "emu" else 16, seed=3)


@pytest.mark.parametrize("backend", BACKENDS)
def test_block_kernel_for_small_envs_matches_oracle(backend):
    """N <= 64 normally runs the one-wavefront-per-env kernel (hwy_wave.h); the generic workgroup
    kernel (hwy_device.h, the N > 64 path; hwy_config.tune_block_kernel) must give the same answers on the same inputs."""
    cfg = _abi.highway_fast_default_config()
    cfg.update({"vehicles_count": 50, "lanes_count": 4, "tuning": {"block_kernel": 1}})
    _random_rollout_vs_oracle(backend, cfg, True, 6 if backend == "emu" else 256, 6 if backend == "emu" else 30, seed=4)
    cfg = _abi.highway_default_config()
    cfg.update({"vehicles_count": 30, "duration": 20, "tuning": {"block_kernel": 1}})
    _random_rollout_vs_oracle(backend, cfg, False, 3 if backend == "emu" else 64, 2 if backend == "emu" else 8, seed=5)


@pytest.mark.parametrize("backend", BACKENDS)
def test_equal_x_ties_take_the_literal_scan_path(backend):
    """Two vehicles at the same longitudinal coordinate: Road.neighbour_vehicles tie rules
    (front: last in list wins; rear: first wins, road/road.py:539-544) must hold exactly."""
    cfg_d = _abi.highway_fast_default_config()
    cfg_d.update({"vehicles_count": 7, "lanes_count": 3})
    cfg = _abi.make_config(cfg_d, 1, fast=True)
    st = spawn.spawn_reference_stream(cfg, [5], 1.5, 1.0)
    # put vehicles 2,3 and 5,6 at identical x (different / same lanes), a leader group ahead of vehicle 1
    st["x"][0, 3] = st["x"][0, 2]
    st["x"][0, 6] = st["x"][0, 5]
    st["y"][0, 2] = st["y"][0, 3] = 4.0
    st["lane"][0, 2] = st["lane"][0, 3] = st["target_lane"][0, 2] = st["target_lane"][0, 3] = 1
    st["y"][0, 1] = 4.0
    st["lane"][0, 1] = st["target_lane"][0, 1] = 1
    st["timer"][0, :] = 1.5  # everybody takes a MOBIL decision this frame
    ref = _abi.copy_state(st)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    for _ in range(3):
        eng.step_frames(None, 1)
        oracle.frames(cfg, ref, None, 1)
        assert_state_close(eng.get_state(), ref, atol=1e-9, what="ties")
    eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_observation_variants_vs_oracle(backend):
    """KinematicObservation options: features, absolute, see_behind, no clip / no normalize."""
    base = _abi.highway_fast_default_config()
    variants = [
        {"type": "Kinematics", "vehicles_count": 7,
         "features": ["presence", "x", "y", "vx", "vy", "heading", "cos_h", "sin_h", "cos_d", "sin_d",
                      "long_off", "lat_off", "ang_off"], "absolute": True, "normalize": False},
        {"type": "Kinematics", "vehicles_count": 3, "see_behind": True, "clip": False},
        {"type": "Kinematics", "vehicles_count": 30, "features": ["x", "vy", "presence"],
         "features_range": {"x": [-100, 100], "vy": [-10, 10]}},
    ]
    for ov in variants:
        cfg_d = dict(base, observation=ov, vehicles_count=12)
        cfg = _abi.make_config(cfg_d, 6, fast=True)
        st = spawn.spawn_reference_stream(cfg, np.arange(6) + 40, 1.5, 1.0)
        eng = make_engine(backend, cfg)
        eng.set_state(st)
        acts = np.array([0, 1, 2, 3, 4, 2]).reshape(6, 1)
        eng.step(acts)
        ref = _abi.copy_state(st)
        o2 = oracle.step(cfg, ref, acts)[0]
        np.testing.assert_allclose(eng.observe(), o2, rtol=0, atol=1e-6)
        np.testing.assert_allclose(eng.observe(), oracle.observe(cfg, ref), rtol=0, atol=1e-6)
        eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_invalid_action_raises_keyerror(backend):
    cfg = _abi.make_config(_abi.highway_fast_default_config(), 2, fast=True)
    eng = make_engine(backend, cfg)
    eng.set_state(spawn.spawn_reference_stream(cfg, [0, 1], 1.5, 1.0))
    with pytest.raises(KeyError):
        eng.step([[1], [5]])
    eng.close()


def test_ulp_noise_only_moves_post_termination_wrecks():
    """Explains the one class of divergence between two correct f64 implementations with different
    libm (numpy / glibc / ocml): after `terminated` a wreck stays in resting contact with the cars
    it hit; impact resolution leaves the rectangles exactly touching, so the next frames' SAT
    `distance > 0` tests sit on a knife edge.  Perturbing every libm result of the kernel by +-1 ulp
    (tests/emu, -DHWY_EMU_ULP_NOISE) never changes a live episode (flags exact, floats 1e-7) but does
    move rotting wrecks.  Episodes end at `terminated`, so this regime is never observed by a user."""
    import ctypes as C
    import subprocess
    import tests.emu.emu as emu
    src = emu.os.path.join(emu._HERE, "emu_engine.cpp")
    noisy = emu.os.path.join(emu._HERE, "_build", "libhwy_emu_noisy.so")
    emu.build()
    subprocess.run(["g++", "-std=c++20", "-O1", "-pthread", "-fPIC", "-shared", "-ffp-contract=off",
                    "-DHWY_EMU_ULP_NOISE", "-o", noisy, src], check=True, capture_output=True)
    lib = C.CDLL(noisy)
    lib.emu_config_size.restype = C.c_size_t
    saved, emu._lib = emu._lib, lib
    try:
        cfg_d = _abi.highway_fast_default_config()
        cfg_d.update({"vehicles_count": 50, "lanes_count": 4})
        E = 96
        cfg = _abi.make_config(cfg_d, E, fast=True)
        st = spawn.spawn_reference_stream(cfg, np.arange(E) + 1000, 1.5, 1.0)
        ref = _abi.copy_state(st)
        eng = emu.EmuEngine(cfg)
        rng = np.random.default_rng(1)
        dead = np.zeros(E, bool)
        live_bad = wreck_bad = 0
        for t in range(14):
            acts = rng.integers(0, 5, size=(E, 1)).astype(np.int32)
            eng.set_state(ref)  # teacher-forced: isolate the flips of THIS step
            eng.step(acts)
            _, _, te, tr, _ = oracle.step(cfg, ref, acts)
            got = eng.get_state()
            bad = np.zeros(E, bool)
            for k in ("x", "y", "heading", "speed"):
                bad |= (np.abs(got[k] - ref[k]) > 1e-7).any(1)
            for k in ("lane", "target_lane", "flags"):
                bad |= (got[k] != ref[k]).any(1)
            live_bad += int((bad & ~dead).sum())
            wreck_bad += int((bad & dead).sum())
            dead |= te | tr
        assert dead.sum() > 20          # plenty of terminated episodes in the sample
        assert live_bad == 0            # live episodes: immune to ulp noise
        print(f"ulp-noise probe: live mismatches {live_bad}, post-termination wreck mismatches {wreck_bad}")
    finally:
        emu._lib = saved
