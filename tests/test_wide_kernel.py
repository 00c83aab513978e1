"""The K-vehicles-per-thread one-wavefront kernel (highwayenv_amd/csrc/hwy_wave2.h: K = 2 for 64 < N <= 128, BASELINE config 3's
N = 101; K = 3 / 4 for N <= 192 / 256 since round 5) against the workgroup kernel it replaces there (hwy_device.h,
`tuning.block_kernel`) and against the C oracle.

Both kernels call the same per-vehicle device functions on the same source expressions, so the comparison between them is
BIT FOR BIT (state planes, rewards, flags; observations to an f64 ulp before their rounding to f32) -- across free-running episodes with device auto-resets, the
ego-only and the full-pairwise collision paths, several controlled vehicles, equal-x ties and the multi-step launch.
"""
import numpy as np
import pytest

from highwayenv_amd import _abi, spawn
from oracle import oracle
from tests.backends import BACKENDS, make_engine
from tests.golden_util import assert_state_close
from tests.test_engine_parity import _random_rollout_vs_oracle

STATE_KEYS = ("x", "y", "heading", "speed", "timer", "target_speed", "delta", "impact_x", "impact_y", "lane", "target_lane",
              "speed_index", "flags")


def _pair(backend, cfg_d, E, fast):
    # (block_kernel=2: the one-wavefront kernels wherever they exist -- the engine's own choice beyond N = 128 is the workgroup kernel)
    wide = make_engine(backend, _abi.make_config(dict(cfg_d, tuning={"block_kernel": 2}), E, fast=fast))
    block = make_engine(backend, _abi.make_config(dict(cfg_d, tuning={"block_kernel": 1}), E, fast=fast))
    return wide, block


def _assert_same(a, b, what):
    for k in STATE_KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=f"{what}: {k}")


CASES = [
    # (config overrides, fast, envs emu / hip, steps emu / hip)
    pytest.param({"vehicles_count": 100}, False, (2, 96), (3, 30), id="v0_n101_full_pairwise"),
    pytest.param({"vehicles_count": 99, "lanes_count": 4, "duration": 6}, True, (3, 128), (8, 40), id="fast_n100_ego_only"),
    pytest.param({"vehicles_count": 70, "controlled_vehicles": 3, "lanes_count": 3, "duration": 8}, False, (2, 64), (3, 24),
                 id="v0_n73_three_agents"),
    pytest.param({"vehicles_count": 127, "lanes_count": 5, "vehicles_density": 2.0}, False, (1, 32), (2, 12), id="v0_n128_dense"),
    pytest.param({"vehicles_count": 64, "lanes_count": 2, "vehicles_density": 2.5, "duration": 5}, True, (2, 64), (6, 30),
                 id="fast_n65_two_lanes"),
    # three / four vehicles per thread: the N > 128 path (round 5)
    pytest.param({"vehicles_count": 150, "lanes_count": 4}, False, (2, 48), (3, 20), id="v0_n151_three_per_thread"),
    pytest.param({"vehicles_count": 200, "lanes_count": 5, "vehicles_density": 1.5, "duration": 8}, False, (1, 32), (3, 16), id="v0_n201_four_per_thread"),
    pytest.param({"vehicles_count": 253, "controlled_vehicles": 3, "lanes_count": 6, "vehicles_density": 2.0, "duration": 6}, False, (1, 24), (2, 12),
                 id="v0_n256_three_agents"),
    pytest.param({"vehicles_count": 140, "lanes_count": 3, "duration": 5}, True, (2, 48), (7, 30), id="fast_n141_ego_only"),
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("over,fast,envs,steps", CASES)
def test_wide_kernel_bit_identical_to_workgroup_kernel(backend, over, fast, envs, steps):
    cfg_d = _abi.highway_fast_default_config() if fast else _abi.highway_default_config()
    cfg_d.update(over)
    E, T = (envs[0], steps[0]) if backend == "emu" else (envs[1], steps[1])
    wide, block = _pair(backend, cfg_d, E, fast)
    cfg = _abi.make_config(cfg_d, E, fast=fast)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 77, cfg_d["ego_spacing"], cfg_d["vehicles_density"], cfg_d["initial_lane_id"])
    for eng in (wide, block):
        eng.set_state(_abi.copy_state(st))
        eng.set_autoreset(True, base_seed=1234, ego_spacing=cfg_d["ego_spacing"], vehicles_density=cfg_d["vehicles_density"],
                          initial_lane_id=-1 if cfg_d["initial_lane_id"] is None else cfg_d["initial_lane_id"])
    rng = np.random.default_rng(5)
    n_done = n_crash = 0
    for t in range(T):
        acts = rng.integers(0, 5, size=(E, cfg.num_agents)).astype(np.int32)
        ow, rw, tew, trw, iw = wide.step(acts)
        ob, rb, teb, trb, ib = block.step(acts)
        what = f"step {t}"
        # (the observation's lmap: the one-wavefront kernels multiply by host-computed reciprocals of the feature ranges, the
        # workgroup kernel divides -- an f64 ulp apart before the rounding to f32)
        np.testing.assert_allclose(ow, ob, rtol=0, atol=1e-7, err_msg=what)
        np.testing.assert_array_equal(rw, rb, err_msg=what)
        np.testing.assert_array_equal(tew, teb, err_msg=what)
        np.testing.assert_array_equal(trw, trb, err_msg=what)
        np.testing.assert_array_equal(iw["crashed"], ib["crashed"], err_msg=what)
        np.testing.assert_array_equal(iw["speed"], ib["speed"], err_msg=what)
        _assert_same(wide.get_state(), block.get_state(), what)
        n_done += int((tew | trw).sum())
        n_crash += int(tew.sum())
    if backend == "hip":
        assert n_done > 0, "no episode ended: the auto-reset path of the kernel was not exercised"
    print(f"wide == workgroup kernel over {T} steps x {E} envs ({n_done} episode ends, {n_crash} crashes)")
    wide.close()
    block.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_kernel_frames_only_and_observe(backend):
    """hwy_step_frames (no observation / reward, no auto-reset) and hwy_observe after it."""
    cfg_d = _abi.highway_default_config()
    cfg_d.update({"vehicles_count": 100})
    E = 2 if backend == "emu" else 32
    wide, block = _pair(backend, cfg_d, E, False)
    cfg = _abi.make_config(cfg_d, E, fast=False)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 5, cfg_d["ego_spacing"], cfg_d["vehicles_density"], cfg_d["initial_lane_id"])
    acts = np.full((E, 1), 3, np.int32)
    for eng in (wide, block):
        eng.set_state(_abi.copy_state(st))
        eng.step_frames(acts, 4)
        eng.step_frames(None, 3)
    _assert_same(wide.get_state(), block.get_state(), "frames only")
    np.testing.assert_allclose(wide.observe(), block.observe(), rtol=0, atol=1e-7)
    wide.close()
    block.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_kernel_equal_x_ties(backend):
    """Vehicles at identical longitudinal coordinates, in both slots of a thread and across them: the literal-scan path."""
    cfg_d = _abi.highway_default_config()
    cfg_d.update({"vehicles_count": 89, "lanes_count": 3})
    E = 2
    cfg = _abi.make_config(cfg_d, E, fast=False)
    st = spawn.spawn_reference_stream(cfg, [3, 4], cfg_d["ego_spacing"], cfg_d["vehicles_density"], cfg_d["initial_lane_id"])
    for a, b in ((2, 3), (10, 70), (66, 67), (63, 64)):  # same slot, across slots, second slot, the slot boundary
        st["x"][:, b] = st["x"][:, a]
        lane_b = (st["lane"][:, a] + 1) % 3
        st["lane"][:, b] = st["target_lane"][:, b] = lane_b
        st["y"][:, b] = lane_b * 4.0
    st["timer"][:, :] = 1.5  # everybody takes a MOBIL decision in the first frame
    ref = _abi.copy_state(st)
    wide, block = _pair(backend, cfg_d, E, False)
    for eng in (wide, block):
        eng.set_state(_abi.copy_state(st))
    for f in range(3):
        wide.step_frames(None, 1)
        block.step_frames(None, 1)
        oracle.frames(cfg, ref, None, 1)
        _assert_same(wide.get_state(), block.get_state(), f"ties, frame {f}")
        assert_state_close(wide.get_state(), ref, atol=1e-9, what=f"ties, frame {f}")
    wide.close()
    block.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_kernel_vs_oracle(backend):
    """Free-running episodes against the C oracle (host re-spawns on the reference's stream), both collision modes."""
    cfg = _abi.highway_default_config()
    cfg.update({"vehicles_count": 100})
    _random_rollout_vs_oracle(backend, cfg, False, 2 if backend == "emu" else 96, 2 if backend == "emu" else 12, seed=11)
    cfg = _abi.highway_fast_default_config()
    cfg.update({"vehicles_count": 90, "lanes_count": 4})
    _random_rollout_vs_oracle(backend, cfg, True, 3 if backend == "emu" else 128, 6 if backend == "emu" else 30, seed=12)


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_kernel_rollout_equals_single_steps(backend):
    """hwy_rollout: K policy steps in one launch == K launches, bit for bit (auto-resets in between included)."""
    cfg_d = _abi.highway_default_config()
    cfg_d.update({"vehicles_count": 100, "duration": 3})
    E, K = (2, 4) if backend == "emu" else (48, 8)
    cfg = _abi.make_config(cfg_d, E, fast=False)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 9, cfg_d["ego_spacing"], cfg_d["vehicles_density"], cfg_d["initial_lane_id"])
    one, many = make_engine(backend, cfg), make_engine(backend, _abi.make_config(cfg_d, E, fast=False))
    for eng in (one, many):
        eng.set_state(_abi.copy_state(st))
        eng.set_autoreset(True, base_seed=99, ego_spacing=cfg_d["ego_spacing"], vehicles_density=cfg_d["vehicles_density"],
                          initial_lane_id=-1 if cfg_d["initial_lane_id"] is None else cfg_d["initial_lane_id"])
    acts = np.random.default_rng(1).integers(0, 5, size=(K, E, 1)).astype(np.int32)
    obs, reward, term, trunc, info = many.rollout(acts)
    for k in range(K):
        o, r, te, tr, i = one.step(acts[k])
        np.testing.assert_array_equal(obs[k], o, err_msg=f"step {k}")
        np.testing.assert_array_equal(reward[k], r, err_msg=f"step {k}")
        np.testing.assert_array_equal(term[k], te, err_msg=f"step {k}")
        np.testing.assert_array_equal(trunc[k], tr, err_msg=f"step {k}")
    _assert_same(one.get_state(), many.get_state(), "after the rollout")
    one.close()
    many.close()


def test_abort_chain_on_the_emulator():
    """The abort rule of ongoing lane changes runs per thread in rank space and is resolved to its fixed point by ballots
    (hwy_wave2.h section D) -- against the workgroup kernel's literal link-by-link chain on dense traffic with many
    simultaneous lane changes; a mutant that never applies a verdict must fail the same comparison."""
    import ctypes as C
    import tests.emu.emu as emu

    def soak(E, T):
        cfg_d = _abi.highway_default_config()
        cfg_d.update({"vehicles_count": 110, "lanes_count": 3, "vehicles_density": 2.2, "duration": 12})
        wide, block = _pair("emu", cfg_d, E, False)
        cfg = _abi.make_config(cfg_d, E, fast=False)
        st = spawn.spawn_reference_stream(cfg, np.arange(E) + 31, cfg_d["ego_spacing"], cfg_d["vehicles_density"], cfg_d["initial_lane_id"])
        for eng in (wide, block):
            eng.set_state(_abi.copy_state(st))
            eng.set_autoreset(True, base_seed=7, ego_spacing=cfg_d["ego_spacing"], vehicles_density=cfg_d["vehicles_density"])
        rng = np.random.default_rng(3)
        changing = 0
        for t in range(T):
            acts = rng.integers(0, 5, size=(E, 1)).astype(np.int32)
            wide.step(acts)
            block.step(acts)
            a = wide.get_state()
            _assert_same(a, block.get_state(), f"step {t}")
            changing += int((a["lane"] != a["target_lane"]).sum())
        wide.close()
        block.close()
        return changing

    emu.build()
    assert soak(6, 14) > 20  # ongoing lane changes at the step boundaries alone (the mutant below shows that links do abort)
    src = emu.os.path.join(emu._HERE, "emu_engine.cpp")
    saved = emu._lib
    try:
        mutant = emu.os.path.join(emu._HERE, "_build", "libhwy_emu_noabort.so")
        emu.compile_emulator(src, mutant, ["-DHWY_WIDE_MUTANT_NO_ABORT=1"])
        emu._lib = C.CDLL(mutant)
        emu._lib.emu_config_size.restype = C.c_size_t
        with pytest.raises(AssertionError):
            soak(6, 14)
    finally:
        emu._lib = saved


def test_one_wavefront_abort_chain_on_the_emulator():
    """The same per-thread rank-space chain in the N <= 64 kernel (hwy_wave.h section D: the headline path and highway-v0), against
    the workgroup kernel's literal link-by-link chain, bit for bit, on dense three-lane traffic; and its own no-abort mutant."""
    import ctypes as C
    import tests.emu.emu as emu

    def soak(E, T):
        cfg_d = _abi.highway_default_config()
        cfg_d.update({"vehicles_count": 60, "lanes_count": 3, "vehicles_density": 2.2, "duration": 12})
        wave, block = _pair("emu", cfg_d, E, False)
        cfg = _abi.make_config(cfg_d, E, fast=False)
        st = spawn.spawn_reference_stream(cfg, np.arange(E) + 41, cfg_d["ego_spacing"], cfg_d["vehicles_density"], cfg_d["initial_lane_id"])
        for eng in (wave, block):
            eng.set_state(_abi.copy_state(st))
            eng.set_autoreset(True, base_seed=8, ego_spacing=cfg_d["ego_spacing"], vehicles_density=cfg_d["vehicles_density"])
        rng = np.random.default_rng(4)
        changing = 0
        for t in range(T):
            acts = rng.integers(0, 5, size=(E, 1)).astype(np.int32)
            wave.step(acts)
            block.step(acts)
            a = wave.get_state()
            _assert_same(a, block.get_state(), f"step {t}")
            changing += int((a["lane"] != a["target_lane"]).sum())
        wave.close()
        block.close()
        return changing

    emu.build()
    assert soak(12, 16) > 20
    src = emu.os.path.join(emu._HERE, "emu_engine.cpp")
    saved = emu._lib
    try:
        mutant = emu.os.path.join(emu._HERE, "_build", "libhwy_emu_wave_noabort.so")
        emu.compile_emulator(src, mutant, ["-DHWY_WAVE_MUTANT_NO_ABORT=1"])
        emu._lib = C.CDLL(mutant)
        emu._lib.emu_config_size.restype = C.c_size_t
        with pytest.raises(AssertionError):
            soak(12, 16)
    finally:
        emu._lib = saved


def test_workgroup_kernel_abort_chain_against_its_literal_build():
    """Round 6: the workgroup kernel (hwy_device.h section D, the N > 128 path) resolves the abort rule per thread in rank space too
    -- one workgroup ballot per round instead of one barrier per changer.  Against a build of the same kernel with the literal
    link-by-link chain (-DHWY_BLOCK_LITERAL_CHAIN), bit for bit, on dense traffic at N = 201 (four wavefronts) and N = 90 (two);
    a build that never applies a verdict must fail the same comparison."""
    import ctypes as C
    import tests.emu.emu as emu

    def soak(E, T, n, lanes):
        cfg_d = _abi.highway_default_config()
        cfg_d.update({"vehicles_count": n, "lanes_count": lanes, "vehicles_density": 2.2, "duration": 12, "tuning": {"block_kernel": 1}})
        block = make_engine("emu", _abi.make_config(cfg_d, E, fast=False))
        cfg = _abi.make_config(cfg_d, E, fast=False)
        st = spawn.spawn_reference_stream(cfg, np.arange(E) + 51, cfg_d["ego_spacing"], cfg_d["vehicles_density"], cfg_d["initial_lane_id"])
        block.set_state(_abi.copy_state(st))
        block.set_autoreset(True, base_seed=9, ego_spacing=cfg_d["ego_spacing"], vehicles_density=cfg_d["vehicles_density"])
        rng = np.random.default_rng(6)
        trace = []
        for t in range(T):
            block.step(rng.integers(0, 5, size=(E, 1)).astype(np.int32))
            trace.append(block.get_state())
        block.close()
        return trace

    def same(a, b):
        for t, (x, y) in enumerate(zip(a, b)):
            _assert_same(x, y, f"step {t}")

    emu.build()
    runs = ((1, 9, 200, 4), (3, 10, 89, 3))
    ours = [soak(*r) for r in runs]
    assert sum(int((s["lane"] != s["target_lane"]).sum()) for tr in ours for s in tr) > 20
    src = emu.os.path.join(emu._HERE, "emu_engine.cpp")
    saved = emu._lib
    try:
        for flag, lib, must_match in (("-DHWY_BLOCK_LITERAL_CHAIN=1", "libhwy_emu_block_literal.so", True),
                                      ("-DHWY_BLOCK_MUTANT_NO_ABORT=1", "libhwy_emu_block_noabort.so", False)):
            path = emu.os.path.join(emu._HERE, "_build", lib)
            emu.compile_emulator(src, path, [flag])
            emu._lib = C.CDLL(path)
            emu._lib.emu_config_size.restype = C.c_size_t
            theirs = [soak(*r) for r in runs]
            if must_match:
                for a, b in zip(ours, theirs):
                    same(a, b)
            else:
                with pytest.raises(AssertionError):
                    for a, b in zip(ours, theirs):
                        same(a, b)
    finally:
        emu._lib = saved


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("n_other,kernel", [(8, 0), (8, 1), (80, 2), (80, 1), (140, 1)], ids=["wave", "block_small", "wide", "block_2w", "block_3w"])
def test_abort_chain_of_depth_two_vs_oracle(backend, n_other, kernel):
    """A directed chain of depth two (behavior.py:229-244): r0 (lane 2 -> 1, nobody ahead) keeps going; c1 (lane 0 -> 1, 30 m behind
    r0) is blocked by r0; c2 (lane 2 -> 1, 35 m behind c1, 65 m behind r0) is blocked by c1 ONLY while c1 still heads for lane 1.
    In list order r0, c1, c2 the literal chain lets c1 abort first and c2 go on; in list order c2 < c1 the rule reads c1's
    frame-start target and c2 aborts too.  Every kernel family resolves this per thread by a fixed-point iteration -- all six list
    orders against the oracle's literal loop, and the two outcomes spelled out."""
    import itertools
    cfg_d = _abi.highway_default_config()
    cfg_d.update({"vehicles_count": n_other, "lanes_count": 4, "tuning": {"block_kernel": kernel}})
    perms = list(itertools.permutations((1, 2, 3)))
    E = len(perms)
    cfg = _abi.make_config(cfg_d, E, fast=False)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 9, cfg_d["ego_spacing"], cfg_d["vehicles_density"], cfg_d["initial_lane_id"])
    N = st["x"].shape[1]
    # everybody else: far ahead on lane 3, one behind the other, no decision due
    st["x"][:, :] = 2000.0 + 40.0 * np.arange(N)[None, :]
    st["y"][:, :] = 12.0
    st["heading"][:, :] = 0.0
    st["speed"][:, :] = 20.0
    st["target_speed"][:, :] = 20.0
    st["lane"][:, :] = 3
    st["target_lane"][:, :] = 3
    st["timer"][:, :] = 0.0
    st["impact_x"][:, :] = 0.0
    st["impact_y"][:, :] = 0.0
    st["flags"][:, 1:] &= ~(_abi.F_CRASHED | _abi.F_HAS_IMPACT)
    for e, (ir0, ic1, ic2) in enumerate(perms):
        for idx, x, lane in ((ir0, 130.0, 2), (ic1, 100.0, 0), (ic2, 65.0, 2)):
            st["x"][e, idx] = x
            st["y"][e, idx] = 4.0 * lane
            st["lane"][e, idx] = lane
            st["target_lane"][e, idx] = 1
    ref = _abi.copy_state(st)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    eng.step_frames(np.full((E, 1), 1, np.int32), 1)
    oracle.frames(cfg, ref, np.full((E, 1), 1, np.int32), 1)
    got = eng.get_state()
    np.testing.assert_array_equal(got["target_lane"], ref["target_lane"])
    assert_state_close(got, ref, atol=1e-9, what="depth-two chain")
    for e, (ir0, ic1, ic2) in enumerate(perms):
        assert got["target_lane"][e, ir0] == 1 and got["target_lane"][e, ic1] == 0, (e, got["target_lane"][e, :4])
        # c2 goes on exactly when c1 acted (and aborted) before it
        assert got["target_lane"][e, ic2] == (1 if ic1 < ic2 else 2), (e, got["target_lane"][e, :4])
    eng.close()


def test_abort_chain_window_bound_dominates_the_desired_gap():
    """The wide kernel's abort chain stops walking the vehicles ahead of a changer at
    bound = (10 + 1.5 v + v (v + 5) / (2 sqrt(ab))) (1 + 1e-6) + 1e-6, claimed to dominate IDMVehicle.desired_gap(changer, rival)
    (behavior.py:192-217) for EVERY rival while no vehicle of the environment drives backwards (v cos h >= 0) or sideways faster
    than 5 m/s (|v sin h| <= 5) and the changer itself has v >= 0, cos h >= 0 -- the kernel checks exactly these conditions and walks
    without a bound otherwise (hwy_wave2.h section D).  Checked here on the formula the oracle evaluates, random and extreme cases."""
    rng = np.random.default_rng(0)
    n = 2_000_000
    v = np.concatenate([rng.uniform(0, 60, n), [0.0, 40.0, 1e-9, 100.0]])
    h = np.concatenate([rng.uniform(-np.pi / 2, np.pi / 2, n), [0.0, np.pi / 2, -np.pi / 2, 0.3]])
    m = v.size
    # rivals: forward component in [0, 80], lateral component in [-5, 5] (the `sane` condition), any mix of speed and heading
    fx = np.concatenate([rng.uniform(0, 80, m - 3), [0.0, 0.0, 80.0]])
    fy = np.concatenate([rng.uniform(-5, 5, m - 3), [5.0, -5.0, 5.0]])
    ce, se = np.cos(h), np.sin(h)
    dv = (v * ce - fx) * ce + (v * se - fy) * se
    d_star = 10.0 + v * 1.5 + (v * dv) * 0.12909944487358055
    bound = (10.0 + v * 1.5 + v * (v + 5.0) * 0.12909944487358055) * (1.0 + 1e-6) + 1e-6
    assert (d_star <= bound).all(), float((d_star - bound).max())
