"""hwy_rollout_device: K policy steps with pre-staged actions == K calls of hwy_step_device, BIT FOR BIT -- observations,
rewards, flags, info and the state afterwards, auto-reset in between included.  On the one-wavefront kernel (N <= 64, straight
road) the K steps are ONE launch (hwy_rollout_wave_kernel); the workgroup kernel and the road-network kernels run K launches.

Backends as everywhere: emu = the kernel source on the CPU, hip = the product on the MI355X through the C-ABI."""
import numpy as np
import pytest

from highwayenv_amd import _abi, merge, spawn
from tests.backends import BACKENDS, make_engine


def _compare(backend, cfg, st, K, n_actions, autoreset_kw):
    rng = np.random.default_rng(11)
    acts = rng.integers(0, n_actions, size=(K, cfg.num_envs, cfg.num_agents)).astype(np.int32)
    one, many = make_engine(backend, cfg), make_engine(backend, cfg)
    for eng in (one, many):
        eng.set_state(_abi.copy_state(st))
        eng.set_autoreset(True, base_seed=123, **autoreset_kw)
    outs = [one.step(acts[k]) for k in range(K)]
    obs, reward, term, trunc, info = many.rollout(acts)
    n_done = 0
    for k in range(K):
        o, r, te, tr, inf = outs[k]
        np.testing.assert_array_equal(obs[k], o, err_msg=f"obs, step {k}")
        np.testing.assert_array_equal(reward[k], r, err_msg=f"reward, step {k}")
        np.testing.assert_array_equal(term[k], te, err_msg=f"terminated, step {k}")
        np.testing.assert_array_equal(trunc[k], tr, err_msg=f"truncated, step {k}")
        np.testing.assert_array_equal(info["speed"][k], inf["speed"], err_msg=f"info speed, step {k}")
        np.testing.assert_array_equal(info["crashed"][k], inf["crashed"], err_msg=f"info crashed, step {k}")
        n_done += int((te | tr).sum())
    s1, sk = one.get_state(), many.get_state()
    for f in s1:
        np.testing.assert_array_equal(sk[f], s1[f], err_msg=f"state {f}")
    for eng in (one, many):
        eng.close()
    return n_done


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("fast", [True, False], ids=["highway-fast", "highway-v0"])
def test_k_steps_in_one_launch_equal_k_launches(backend, fast):
    """The one-wavefront kernel: ego-only and full pairwise collision builds; short episodes so that environments end and are
    re-spawned INSIDE the multi-step launch."""
    cfg_d = _abi.highway_fast_default_config() if fast else _abi.highway_default_config()
    cfg_d.update({"vehicles_count": 30, "lanes_count": 3, "duration": 4, "vehicles_density": 2.0})
    E, K = (6, 7) if backend == "emu" else (256, 16)
    cfg = _abi.make_config(cfg_d, E, fast=fast)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 40, cfg_d["ego_spacing"], cfg_d["vehicles_density"])
    n_done = _compare(backend, cfg, st, K, 5, {"ego_spacing": cfg_d["ego_spacing"], "vehicles_density": cfg_d["vehicles_density"]})
    assert n_done >= E  # duration 4: every environment was re-spawned at least once within the K steps


@pytest.mark.parametrize("backend", BACKENDS)
def test_k_steps_multi_agent_occupancy_grid(backend):
    """3 controlled vehicles, OccupancyGrid outputs (the K blocks of the observation plane are [E][A][F][W][H] each)."""
    cfg_d = _abi.highway_fast_default_config()
    cfg_d.update({"vehicles_count": 20, "lanes_count": 4, "duration": 5, "controlled_vehicles": 3,
                  "action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
                  "observation": {"type": "MultiAgentObservation", "observation_config": {"type": "OccupancyGrid"}}})
    E, K = (3, 4) if backend == "emu" else (64, 8)
    cfg = _abi.make_config(cfg_d, E, fast=True)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 7, cfg_d["ego_spacing"], cfg_d["vehicles_density"])
    _compare(backend, cfg, st, K, 5, {"ego_spacing": cfg_d["ego_spacing"], "vehicles_density": cfg_d["vehicles_density"]})


@pytest.mark.parametrize("backend", BACKENDS)
def test_k_steps_on_the_workgroup_kernel(backend):
    """N = 101 (two wavefronts per environment): K launches back to back, same contract."""
    cfg_d = _abi.highway_default_config()
    cfg_d.update({"vehicles_count": 100, "duration": 3})
    E, K = (2, 4) if backend == "emu" else (32, 6)
    cfg = _abi.make_config(cfg_d, E, fast=False)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 3, cfg_d["ego_spacing"], cfg_d["vehicles_density"])
    _compare(backend, cfg, st, K, 5, {"ego_spacing": cfg_d["ego_spacing"], "vehicles_density": cfg_d["vehicles_density"]})


@pytest.mark.parametrize("backend", BACKENDS)
def test_k_steps_on_the_merge_kernel(backend):
    cfg_d = merge.merge_generic_default_config()
    cfg_d.update({"lanes_count": 3, "vehicles_count": 20})
    E, K = (3, 5) if backend == "emu" else (64, 12)
    cfg = _abi.make_config(cfg_d, E, scenario="merge-generic")
    st = merge.spawn_reference_stream(cfg, cfg_d, True, np.arange(E) + 9)
    _compare(backend, cfg, st, K, 5, {})


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("slots,grid", [(30, True), (40, False)])  # helper-lane build (N <= 32) and the 64-slot build
def test_k_steps_on_the_intersection_kernel(backend, slots, grid):
    """Device traffic (clear / spawn / re-spawn on Philox inside the launch).  The multi-step launch holds the STEP blocks only: an
    environment that ends in it warms its next episode up inline, where one launch per step pre-warms it in shadow blocks -- WHEN the
    warm-up frames are computed must not change a result."""
    from highwayenv_amd import intersection as hix
    cfg_d = hix.intersection_default_config()
    cfg_d.update({"max_vehicles": slots, "duration": 5})
    if grid:
        cfg_d["observation"] = {"type": "OccupancyGrid"}
    E, K = (3, 7) if backend == "emu" else (64, 14)
    cfg = _abi.make_config(cfg_d, E, scenario="intersection")
    a, b = make_engine(backend, cfg), make_engine(backend, cfg)
    acts = np.random.default_rng(2).integers(0, 3, size=(K, E, 1)).astype(np.int32)
    for eng in (a, b):
        eng.reset(base_seed=21)
        eng.set_autoreset(True, base_seed=22)
    outs = [a.step(acts[k]) for k in range(K)]
    obs, reward, term, trunc, info = b.rollout(acts)
    for k in range(K):
        np.testing.assert_array_equal(obs[k], outs[k][0])
        np.testing.assert_array_equal(reward[k], outs[k][1])
        np.testing.assert_array_equal(term[k], outs[k][2])
    sa, sb = a.get_state(), b.get_state()
    for f in sa:
        np.testing.assert_array_equal(sa[f], sb[f], err_msg=f)
    for eng in (a, b):
        eng.close()


@pytest.mark.gpu
def test_k_steps_refused_with_host_side_traffic():
    """An intersection engine with HWY_C_HOST_TRAFFIC (spawn_mode="reference") has its vehicles cleared / spawned by the host
    between policy steps: a K-step launch would skip K - 1 of those passes, so hwy_rollout* refuse k_steps > 1 there
    (include/hwy_engine.h) instead of returning something that is not "K calls of hwy_step"."""
    from highwayenv_amd import intersection as hix
    from highwayenv_amd.engine import Engine, EngineError
    cfg_d = hix.intersection_default_config()
    cfg_d.update({"max_vehicles": 30, "host_traffic": True})
    cfg = _abi.make_config(cfg_d, 4, scenario="intersection")
    dev = Engine(_abi.make_config(dict(cfg_d, host_traffic=False), 4, scenario="intersection"))
    dev.reset(base_seed=3)
    st = dev.get_state()
    dev.close()
    eng = Engine(cfg)
    eng.set_state(st)
    acts = np.ones((2, 4, 1), np.int32)
    with pytest.raises(EngineError, match="device traffic"):
        eng.rollout(acts)
    obs, *_ = eng.rollout(acts[:1])   # one step is a step
    assert obs.shape[0] == 1
    eng.close()
