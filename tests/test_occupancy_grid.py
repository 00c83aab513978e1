"""OccupancyGridObservation (highway_env/envs/common/observation.py:279-499) on the straight highway:
the rasterisation kernel against grids recorded from the unmodified reference and against the oracle.
SURVEY.md section 8(f) row 3.  Grid values are f32; cells are compared exactly where the reference
holds 0/1 (presence, on_road, empty) and at 1e-6 elsewhere."""
import numpy as np
import pytest

from highwayenv_amd import _abi, envs, spawn
from oracle import oracle
from tests.backends import BACKENDS, make_engine
from tests.golden_util import GRID, Golden


@pytest.mark.parametrize("name", GRID)
def test_oracle_grid_matches_reference(name):
    g = Golden(name)
    cfg = g.hwy_config()
    np.testing.assert_allclose(oracle.observe(cfg, g.state("init"))[:, 0], g.z["obs0"], rtol=0, atol=1e-7)
    for t in range(g.steps):
        np.testing.assert_allclose(oracle.observe(cfg, g.state("step", t))[:, 0], g.z["obs"][t], rtol=0, atol=1e-7)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", GRID)
def test_engine_grid_matches_reference(backend, name):
    """Teacher-forced on the reference's own states (observe only), then free-running steps."""
    g = Golden(name)
    cfg = _abi.make_config(g.config, g.E, fast=g.fast)
    assert _abi.obs_shape(cfg) == g.z["obs0"].shape[1:]
    eng = make_engine(backend, cfg)
    eng.set_state(g.state("init"))
    np.testing.assert_allclose(eng.observe()[:, 0], g.z["obs0"], rtol=0, atol=1e-6)
    for t in range(g.steps):
        eng.set_state(g.state("step", t))
        np.testing.assert_allclose(eng.observe()[:, 0], g.z["obs"][t], rtol=0, atol=1e-6, err_msg=f"{name} step {t}")
    # free-running from reset: obs returned by step() while the episode is collision-free
    eng.set_state(g.state("init"))
    for t in range(g.steps):
        obs = eng.step(g.actions[t])[0]
        want = g.state("step", t)
        ok = ~((want["flags"] & (_abi.F_CRASHED | _abi.F_HAS_IMPACT)) != 0).any(1)
        np.testing.assert_allclose(obs[ok, 0], g.z["obs"][t][ok], rtol=0, atol=1e-6, err_msg=f"{name} free step {t}")
        if not ok.all():
            break
    eng.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_grid_contested_cells_lowest_index_wins_and_multi_agent(backend):
    """Several vehicles in one cell: the reference iterates the list in reverse, so the lowest index
    owns the cell.  Also 2 observers per env and N = 101 (two wavefronts per env)."""
    cfg_d = _abi.highway_default_config()
    cfg_d.update({"vehicles_count": 99, "controlled_vehicles": 2, "lanes_count": 4, "vehicles_density": 3.0,
                  "observation": {"type": "OccupancyGrid", "grid_size": [[-40, 40], [-10, 10]], "grid_step": [8, 5],
                                  "features": ["presence", "x", "vx", "heading", "on_road"]}})
    E = 3
    cfg = _abi.make_config(cfg_d, E)
    st = spawn.spawn_reference_stream(cfg, np.arange(E) + 9, 2.0, 3.0)
    eng = make_engine(backend, cfg)
    eng.set_state(st)
    got = eng.observe()
    want = oracle.observe(cfg, st)
    assert got.shape == (E, 2, 5, 10, 4)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    assert (want[:, :, 0].sum(axis=(2, 3)) >= 2).all()  # grids are populated
    acts = np.ones((E, 2), np.int32)
    ref = _abi.copy_state(st)
    np.testing.assert_allclose(eng.step(acts)[0], oracle.step(cfg, ref, acts)[0], rtol=0, atol=1e-6)
    eng.close()


def test_grid_config_errors_match_reference():
    base = _abi.highway_fast_default_config()
    with pytest.raises(NotImplementedError):  # observation.py:358-359
        _abi.make_config(dict(base, observation={"type": "OccupancyGrid", "absolute": True}), 1)
    with pytest.raises(KeyError):
        _abi.make_config(dict(base, observation={"type": "Kinematics", "features": ["presence", "on_road"]}), 1)
    c = _abi.make_config(dict(base, observation={"type": "OccupancyGrid", "grid_size": [[-10, 20], [-6, 6]], "grid_step": [4, 3]}), 1)
    assert _abi.obs_shape(c) == (4, 7, 4)  # floor(30/4), floor(12/3)


def test_env_api_exposes_grid_shape():
    class Emu(envs.BatchedHighwayEnvFast):
        @staticmethod
        def _engine_factory(cfg, device, stream):
            from tests.emu.emu import EmuEngine
            return EmuEngine(cfg)

    env = Emu({"observation": {"type": "OccupancyGrid"}}, num_envs=2)
    obs, _ = env.reset(seed=0)
    assert obs.shape == (2, 4, 11, 11) and env.single_observation_space.shape == (4, 11, 11)
    g = Golden("grid_default")
    env2 = Emu(g.config, num_envs=3)
    obs, _ = env2.reset(seed=0)
    np.testing.assert_allclose(obs, g.z["obs0"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("half,step", [(15.0, 1.0), (20.0, 1.0), (27.5, 5.0)])
def test_intersection_grid_workspace_paths_agree_with_the_oracle(backend, half, step):
    """csrc/hwy_ix.h keeps the OccupancyGrid's workspace in LDS when it fits and paints only the DISTINCT clipped waypoints of
    every lane from a list that has to fit too: 30 x 30 cells = LDS without the list, 40 x 40 cells = the global workspace
    (the other scenarios' path), 11 x 11 = BASELINE config 4 (LDS + list).  Same cells either way."""
    from highwayenv_amd import intersection as hix
    from oracle import oracle_ix
    from tests.golden_util import ix_oracle_config, ix_oracle_state
    cfg_d = hix.intersection_default_config()
    # (borders off the waypoint lattice: see the knife edge described in tests/test_fuzz_configs.py)
    cfg_d["observation"] = {"type": "OccupancyGrid", "grid_size": [[-half - 0.3, half - 0.3], [-half + 0.2, half + 0.2]],
                            "grid_step": [step, step], "features": ["presence", "vx", "vy", "on_road"],
                            "align_to_vehicle_axes": step == 1.0 and half == 20.0}
    cfg_d.update(max_vehicles=30, host_traffic=True)
    E = 4
    c = _abi.make_config(cfg_d, E, scenario="intersection")
    eng = make_engine(backend, c)
    obs = eng.reset(seeds=np.arange(E, dtype=np.uint64) + 5)
    st = eng.get_state()
    want = oracle_ix.observe(ix_oracle_config(cfg_d, c, E), ix_oracle_state(st, c))
    assert want[:, -1].sum() > 10  # the on-road layer is not empty
    np.testing.assert_allclose(obs[:, 0], want, rtol=0, atol=1e-6)
    for t in range(2):
        obs = eng.step(np.ones((E, 1), np.int32))[0]
        want = oracle_ix.observe(ix_oracle_config(cfg_d, c, E), ix_oracle_state(eng.get_state(), c))
        np.testing.assert_allclose(obs[:, 0], want, rtol=0, atol=1e-6)
    eng.close()
