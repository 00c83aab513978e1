"""Device-side reset (hwy_reset / auto-reset) against the spawn RULE pinned to the reference.

The device spawn draws from Philox-4x32-10 (counter-based, one stream per env), NOT numpy's PCG64
stream; what must match the reference is the rule that turns draws into traffic
(Vehicle.create_random, highway_env/vehicle/kinematics.py:50-104).  ``spawn.spawn_from_draws`` is
that rule, verified against the reference's own reset states in tests/test_spawn.py; here the
kernel is checked against it on the same Philox uniforms, computed independently in Python.
"""
import numpy as np
import pytest

from highwayenv_amd import _abi, spawn
from oracle import oracle
from tests.backends import BACKENDS, make_engine

M32 = 0xFFFFFFFF


def philox_uniform2(seed, vehicle, episode, draw):
    """Philox-4x32-10 (Salmon et al. 2011), counter (vehicle, episode, draw, 'HWY1'), key = seed."""
    c = [vehicle & M32, episode & M32, draw & M32, 0x48575931]
    k0, k1 = seed & M32, (seed >> 32) & M32
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c[3] ^ k1) & M32, p0 & M32]
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    a, b = (c[0] << 32) | c[1], (c[2] << 32) | c[3]
    return (a >> 11) / 9007199254740992.0, (b >> 11) / 9007199254740992.0


def expected_state(cfg, seeds, episode, ego_spacing, density, initial_lane_id):
    E, N, L = len(seeds), cfg.num_vehicles, cfg.lanes_count
    lane = np.zeros((E, N), np.int64)
    us, up, ud = np.zeros((E, N)), np.zeros((E, N)), np.zeros((E, N))
    ctrl = spawn.controlled_mask(cfg)
    for e, sd in enumerate(seeds):
        for i in range(N):
            u_lane, u_speed = philox_uniform2(int(sd), i, episode, 0)
            u_pos, u_delta = philox_uniform2(int(sd), i, episode, 1)
            lane[e, i] = min(int(u_lane * L), L - 1)
            if ctrl[i] and initial_lane_id >= 0:
                lane[e, i] = initial_lane_id
            us[e, i], up[e, i], ud[e, i] = u_speed, u_pos, u_delta
    return spawn.spawn_from_draws(cfg, lane, us, up, ud, ego_spacing, density)


def assert_spawn_equal(got, want, rows=slice(None)):
    for k in ("lane", "target_lane", "flags", "speed_index"):
        np.testing.assert_array_equal(got[k][rows], want[k][rows], err_msg=k)
    for k in ("x", "y", "heading", "speed", "target_speed", "timer", "delta"):
        np.testing.assert_allclose(got[k][rows], want[k][rows], rtol=0, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("fast,controlled,lane_id", [(True, 1, -1), (False, 1, 2), (False, 3, -1)])
def test_device_reset_follows_the_reference_spawn_rule(backend, fast, controlled, lane_id):
    cfg_d = _abi.highway_fast_default_config() if fast else _abi.highway_default_config()
    cfg_d.update({"vehicles_count": 24, "lanes_count": 4, "controlled_vehicles": controlled})
    E = 5
    cfg = _abi.make_config(cfg_d, E, fast=fast)
    eng = make_engine(backend, cfg)
    seeds = np.array([3, 2**40 + 17, 99, 12345678901234567, 0], np.uint64)
    obs = eng.reset(seeds=seeds, ego_spacing=1.7, vehicles_density=1.3, initial_lane_id=lane_id)
    got = eng.get_state()
    want = expected_state(cfg, seeds, 0, 1.7, 1.3, lane_id)
    assert_spawn_equal(got, want)
    assert (got["time"] == 0).all()
    # first observation == KinematicObservation of the spawned state
    np.testing.assert_allclose(obs, oracle.observe(cfg, want), rtol=0, atol=1e-6)
    # x strictly increasing in creation order (every vehicle is placed ahead of the previous ones)
    assert (np.diff(got["x"], axis=1) > 0).all()
    eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_masked_reset_leaves_other_envs_untouched(backend):
    cfg_d = _abi.highway_fast_default_config()
    cfg = _abi.make_config(cfg_d, 4, fast=True)
    eng = make_engine(backend, cfg)
    eng.reset(seeds=np.arange(4, dtype=np.uint64))
    eng.step(np.ones((4, 1), np.int32))
    before = eng.get_state()
    eng.reset(seeds=np.arange(10, 14, dtype=np.uint64), mask=np.array([0, 1, 0, 1], np.uint8))
    after = eng.get_state()
    for k in before:
        np.testing.assert_array_equal(after[k][[0, 2]], before[k][[0, 2]], err_msg=k)
    want = expected_state(cfg, np.arange(10, 14), 0, 2.0, 1.0, -1)
    assert_spawn_equal(after, want, rows=[1, 3])
    assert (after["time"][[1, 3]] == 0).all() and (after["time"][[0, 2]] == 1).all()
    eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_autoreset_next_step_semantics(backend):
    """gymnasium vector 'next-step' auto-reset: the step after terminated|truncated re-spawns the env
    (new Philox episode), ignores the action, returns the reset observation with reward 0 and both
    flags False; envs that are not done keep stepping exactly like the oracle."""
    cfg_d = _abi.highway_fast_default_config()
    cfg_d.update({"duration": 3, "vehicles_count": 12})
    E = 6
    cfg = _abi.make_config(cfg_d, E, fast=True)
    eng = make_engine(backend, cfg)
    base = 4242
    eng.reset(seeds=np.uint64(base) + np.arange(E, dtype=np.uint64), ego_spacing=1.5)
    eng.set_autoreset(True, base_seed=base, ego_spacing=1.5, vehicles_density=1.0)
    ref = eng.get_state()
    episode = np.zeros(E, np.int64)
    done_prev = np.zeros(E, bool)
    rng = np.random.default_rng(0)
    n_resets = 0
    for t in range(9):
        acts = rng.integers(0, 5, size=(E, 1)).astype(np.int32)
        obs, reward, term, trunc, info = eng.step(acts)
        # oracle on the envs that were live; fresh spawn for the ones that were done
        o2, r2, te2, tr2, _ = oracle.step(cfg, ref, acts)
        for e in np.nonzero(done_prev)[0]:
            episode[e] += 1
            fresh = expected_state(cfg, [base + e], int(episode[e]), 1.5, 1.0, -1)
            for k in ref:
                ref[k][e] = fresh[k][0] if k != "time" else 0.0
            sub = _abi.make_config(cfg_d, 1, fast=True)
            o2[e] = oracle.observe(sub, fresh)[0]
            r2[e], te2[e], tr2[e] = 0.0, False, False
            n_resets += 1
        np.testing.assert_array_equal(term, te2, err_msg=f"step {t}")
        np.testing.assert_array_equal(trunc, tr2, err_msg=f"step {t}")
        np.testing.assert_allclose(reward, r2, rtol=0, atol=1e-9, err_msg=f"step {t}")
        np.testing.assert_allclose(obs, o2, rtol=0, atol=1e-6, err_msg=f"step {t}")
        got = eng.get_state()
        np.testing.assert_allclose(got["x"], ref["x"], rtol=0, atol=1e-7)
        np.testing.assert_array_equal(got["time"], ref["time"])
        done_prev = term | trunc
    assert n_resets >= E  # duration 3 => every env was truncated and re-spawned at least once
    eng.close()
