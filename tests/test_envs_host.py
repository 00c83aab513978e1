"""Host-side API mirror (highwayenv_amd/envs.py): config semantics, error behaviour, unbatched
drop-in signatures.  CPU: the engine is replaced by the CPU emulation of the same kernel source
(test infrastructure); `-m gpu` runs the same API on the real engine."""
import numpy as np
import pytest

from highwayenv_amd import _abi, envs
from highwayenv_amd.engine import EngineError
from tests.golden_util import Golden


def _emu_factory(cfg, device, stream):
    from tests.emu.emu import EmuEngine
    return EmuEngine(cfg)


class EmuFast(envs.BatchedHighwayEnvFast):
    _engine_factory = staticmethod(_emu_factory)


class EmuSingleFast(envs._SingleEnvMixin, EmuFast):
    pass


def test_default_configs_match_reference_dicts():
    g = Golden("cfg1_fast_default")
    ref = g.config
    mine = envs.BatchedHighwayEnvFast.default_config()
    for k in ["lanes_count", "vehicles_count", "simulation_frequency", "policy_frequency", "duration", "ego_spacing",
              "vehicles_density", "collision_reward", "right_lane_reward", "high_speed_reward", "normalize_reward",
              "offroad_terminal", "reward_speed_range"]:
        assert mine[k] == ref[k], k
    v0 = envs.BatchedHighwayEnv.default_config()
    assert (v0["simulation_frequency"], v0["lanes_count"], v0["vehicles_count"], v0["duration"]) == (15, 4, 50, 40)


def test_unknown_types_raise_like_reference():
    with pytest.raises(ValueError, match="Unknown action type"):
        envs.BatchedHighwayEnv({"action": {"type": "Nope"}})
    with pytest.raises(ValueError, match="Unknown observation type"):
        envs.BatchedHighwayEnv({"observation": {"type": "Nope"}})
    with pytest.raises(NotImplementedError):
        envs.BatchedHighwayEnv({"observation": {"type": "TimeToCollision"}})
    with pytest.raises(NotImplementedError):
        envs.BatchedHighwayEnv({"action": {"type": "ContinuousAction"}})
    with pytest.raises(NotImplementedError):
        envs.BatchedHighwayEnv(render_mode="human")


def test_step_before_reset_raises_notimplemented():
    env = envs.BatchedHighwayEnvFast(num_envs=2)
    with pytest.raises(NotImplementedError):
        env.step([1, 1])


def test_no_gpu_means_loud_failure_not_fallback():
    import highwayenv_amd._lib as L
    if L.load().hwy_device_count() > 0:
        pytest.skip("a GPU is present")
    env = envs.BatchedHighwayEnvFast(num_envs=2)
    with pytest.raises(EngineError, match="no CPU fallback"):
        env.reset(seed=0)


@pytest.mark.parametrize("real", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["emu", "hip"])
def test_single_env_dropin_matches_reference_episode(real):
    """HighwayEnvFast(): reset(seed=0), golden actions -> the reference's obs/reward/flags."""
    g = Golden("cfg1_fast_default")
    env = envs.HighwayEnvFast() if real else EmuSingleFast()
    obs, info = env.reset(seed=int(g.seeds[0]))
    assert obs.shape == (5, 5) and obs.dtype == np.float32
    np.testing.assert_allclose(obs, g.z["obs0"][0], atol=1e-6)
    for t in range(6):
        obs, r, te, tr, info = env.step(int(g.actions[t, 0]))
        assert isinstance(r, float) and isinstance(te, bool) and isinstance(tr, bool)
        np.testing.assert_allclose(obs, g.z["obs"][t, 0], atol=1e-6)
        assert abs(r - g.z["reward"][t, 0]) < 1e-9
        assert te == bool(g.z["terminated"][t, 0]) and tr == bool(g.z["truncated"][t, 0])
        assert abs(info["speed"] - g.z["info_speed"][t, 0]) < 1e-9
        assert set(info["rewards"]) == {"collision_reward", "right_lane_reward", "high_speed_reward", "on_road_reward"}
    assert env.vehicle.lane_index[:2] == ("0", "1")
    assert len(env.road().vehicles) == 21
    with pytest.raises(KeyError):
        env.step(9)


@pytest.mark.parametrize("real", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["emu", "hip"])
def test_batched_reset_seed_convention_and_options_config(real):
    env = (envs.BatchedHighwayEnvFast if real else EmuFast)(num_envs=3)
    obs, info = env.reset(seed=0, options={"config": {"lanes_count": 4, "vehicles_count": 50}})
    assert obs.shape == (3, 5, 5)
    g = Golden("cfg2_fast_n50_l4")  # seeds 0..3 of the same config
    np.testing.assert_allclose(obs, g.z["obs0"][:3], atol=1e-6)
    o, r, te, tr, info = env.step(g.actions[0, :3])
    np.testing.assert_allclose(o, g.z["obs"][0, :3], atol=1e-6)
    np.testing.assert_allclose(r, g.z["reward"][0, :3], atol=1e-9)
    # reset() without a seed continues each env's np_random stream like the reference does
    obs2, _ = env.reset()
    assert not np.allclose(obs2, obs)


@pytest.mark.gpu
def test_batched_env_device_spawn_autoreset_runs_and_resets():
    """spawn_mode='device' + autoreset: the pure-GPU path used by the benchmark, through the env API."""
    env = envs.BatchedHighwayEnvFast({"vehicles_count": 50, "lanes_count": 4, "duration": 5}, num_envs=256,
                                     spawn_mode="device", autoreset=True)
    obs, info = env.reset(seed=7)
    assert obs.shape == (256, 5, 5) and np.isfinite(obs).all()
    rng = np.random.default_rng(0)
    n_done = 0
    for t in range(12):
        obs, r, te, tr, info = env.step(rng.integers(0, 5, 256))
        assert np.isfinite(obs).all() and ((0 <= r) & (r <= 1)).all()
        n_done += int((te | tr).sum())
    assert n_done >= 256  # duration 5 => every env truncated (and was re-spawned) at least once
    env.close()


# ---- merge-v0 / merge-generic-v0 (highway_env/envs/merge_env.py) ------------------------------------------
class EmuMerge(envs._SingleMergeMixin, envs.BatchedMergeEnv):
    _engine_factory = staticmethod(_emu_factory)


class EmuMergeGeneric(envs._SingleMergeMixin, envs.BatchedMergeGenericEnv):
    _engine_factory = staticmethod(_emu_factory)


class EmuBatchedMergeGeneric(envs.BatchedMergeGenericEnv):
    _engine_factory = staticmethod(_emu_factory)


class EmuMergeV1(envs._SingleMergeMixin, envs.BatchedConnectedLaneMergeEnv):
    _engine_factory = staticmethod(_emu_factory)


class EmuMergeGenericV1(envs._SingleMergeMixin, envs.BatchedConnectedLaneMergeGenericEnv):
    _engine_factory = staticmethod(_emu_factory)


@pytest.mark.parametrize("real", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["emu", "hip"])
@pytest.mark.parametrize("name", ["merge_default", "merge_generic_l3", "merge_v1", "merge_generic_v1"])
def test_single_merge_env_dropin_matches_reference_episode(real, name):
    """MergeEnv() / MergeGenericEnv(config) and their ConnectedLane* (merge-v1, merge-generic-v1) variants:
    reset(seed=s), golden actions -> the reference's obs/reward/flags."""
    from tests.golden_util import GoldenMerge
    g = GoldenMerge(name)
    v1 = name.endswith("_v1")
    cls = {(False, False, False): EmuMerge, (False, True, False): EmuMergeGeneric,
           (True, False, False): envs.MergeEnv, (True, True, False): envs.MergeGenericEnv,
           (False, False, True): EmuMergeV1, (False, True, True): EmuMergeGenericV1,
           (True, False, True): envs.ConnectedLaneMergeEnv,
           (True, True, True): envs.ConnectedLaneMergeGenericEnv}[(real, g.generic, v1)]
    over = {k: v for k, v in g.config.items() if cls.default_config().get(k) != v} if g.generic else None
    assert cls.default_config()["neighbour_vehicles_connected_lanes"] is v1
    env = cls(over)
    e = 1
    obs, info = env.reset(seed=int(g.seeds[e]))
    assert obs.shape == (5, 5) and obs.dtype == np.float32
    np.testing.assert_allclose(obs, g.z["obs0"][e, 0], atol=1e-6)
    for t in range(g.steps):
        a = int(g.actions[t, e, 0])
        obs, r, te, tr, info = env.step(a)
        assert isinstance(r, float) and isinstance(te, bool) and tr is False
        np.testing.assert_allclose(obs, g.z["obs"][t, e, 0], atol=1e-6)
        assert abs(r - g.z["reward"][t, e]) < 1e-9
        assert te == bool(g.z["terminated"][t, e])
        assert abs(info["speed"] - g.z["info_speed"][t, e]) < 1e-9
        assert set(info["rewards"]) == {"collision_reward", "right_lane_reward", "high_speed_reward",
                                        "lane_change_reward", "merging_speed_reward"}
        # MergeEnv._reward (merge_env.py:40-60): lmap of the weighted sum of _rewards
        c = env.config
        raw = sum(c.get(k, 0) * v for k, v in info["rewards"].items())
        lo, hi = c["collision_reward"] + c["merging_speed_reward"], c["high_speed_reward"] + c["right_lane_reward"]
        assert abs((raw - lo) / (hi - lo) - r) < 1e-9
        if te:
            break
    assert env.vehicle.crashed in (False, True) and len(env.controlled_vehicles) == 1
    env.close()


def test_batched_merge_generic_multi_agent_shapes_and_errors():
    cfg = {"lanes_count": 4, "vehicles_count": 40, "controlled_vehicles": 4,
           "action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
           "observation": {"type": "MultiAgentObservation", "observation_config": {"type": "Kinematics"}}}
    env = EmuBatchedMergeGeneric(cfg, num_envs=3)
    obs, info = env.reset(seed=5)
    assert obs.shape == (3, 4, 5, 5) and np.isfinite(obs).all()
    o, r, te, tr, info = env.step(np.ones((3, 4), int))
    assert o.shape == (3, 4, 5, 5) and r.shape == (3,) and info["agents_rewards"].shape == (3, 4)
    assert not tr.any()
    with pytest.raises(KeyError):
        env.step(np.full((3, 4), 7))
    env.close()
    with pytest.raises(AssertionError):  # the reference's own assert (merge_env.py:241-244)
        EmuBatchedMergeGeneric({"after_merge_length": 50}, num_envs=1)
    assert EmuBatchedMergeGeneric({"observation": {"type": "OccupancyGrid"}}, num_envs=1).single_observation_shape == (4, 11, 11)
    with pytest.raises(NotImplementedError):  # a heading layer has no value for the end-of-lane Obstacle (objects.py:141-160)
        EmuBatchedMergeGeneric({"observation": {"type": "Kinematics", "features": ["presence", "heading"]}}, num_envs=1)


# ---- intersection-v0 (highway_env/envs/intersection_env.py) -----------------------------------------------------
class EmuIntersection(envs._SingleIntersectionMixin, envs.BatchedIntersectionEnv):
    _engine_factory = staticmethod(_emu_factory)


class EmuBatchedIntersection(envs.BatchedIntersectionEnv):
    _engine_factory = staticmethod(_emu_factory)


class EmuIntersectionV2(envs._SingleIntersectionMixin, envs.BatchedConnectedLaneIntersectionEnv):
    _engine_factory = staticmethod(_emu_factory)


@pytest.mark.parametrize("real", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["emu", "hip"])
@pytest.mark.parametrize("e,v2", [(0, False), (2, False), (0, True), (3, True)])
def test_single_intersection_env_dropin_matches_reference_episode(real, e, v2):
    """IntersectionEnv(): reset(seed=s) -- host spawns on the reference's numpy stream, the three warm-up seconds of
    _make_vehicles on the engine -- then golden actions, clearing and spawning on the same stream: the reference's
    obs / reward / terminated / truncated while the episode is live (stop at the first wreck or near-standstill:
    see tests/test_oracle_golden_intersection.py on the ill-conditioned steering of a stopped car)."""
    from tests.golden_util import GoldenIntersection
    g = GoldenIntersection("intersection_v2" if v2 else "intersection_default")
    if v2:  # intersection-v2 (ConnectedLaneIntersectionEnv): the fixture's traffic settings on top of the class defaults
        cls = envs.ConnectedLaneIntersectionEnv if real else EmuIntersectionV2
        assert cls.default_config()["neighbour_vehicles_connected_lanes"] is True
        env = cls({k: g.config[k] for k in ("initial_vehicle_count", "spawn_probability", "duration")} | {"max_vehicles": g.N})
    else:
        env = envs.IntersectionEnv() if real else EmuIntersection()
    obs, info = env.reset(seed=int(g.z["seeds"][e]))
    assert obs.shape == (15, 7) and obs.dtype == np.float32
    np.testing.assert_allclose(obs, g.z["obs0"][e], atol=1e-6)
    assert abs(info["speed"] - 10.0) < 1e-12
    compared = 0
    for t in range(g.steps):
        want = g.state("step", t)
        pres = want["present"][e] != 0
        if ((want["crashed"][e] != 0) | (want["has_impact"][e] != 0))[pres].any() or (np.abs(want["speed"][e][pres]) < 0.5).any():
            break
        obs, r, te, tr, info = env.step(int(g.actions[t, e, 0]))
        assert isinstance(r, float) and isinstance(te, bool) and isinstance(tr, bool)
        np.testing.assert_allclose(obs, g.z["obs"][t, e], atol=1e-6, err_msg=f"step {t}")
        assert abs(r - g.z["reward"][t, e]) < 1e-9
        assert te == bool(g.z["terminated"][t, e]) and tr == bool(g.z["truncated"][t, e])
        assert abs(info["speed"] - g.z["info_speed"][t, e]) < 1e-9
        assert set(info["rewards"]) == {"collision_reward", "high_speed_reward", "arrived_reward", "on_road_reward"}
        # the vehicle list after clear / spawn is the reference's
        st = env.get_state()
        nxt = g.state("next", t)
        n = int(nxt["present"][e].sum())
        assert int(((st["flags"][0] & _abi.F_ABSENT) == 0).sum()) == n
        np.testing.assert_allclose(st["x"][0, :n], nxt["x"][e, :n], atol=1e-7)
        compared += 1
        if te or tr:
            break
    assert compared >= 3
    with pytest.raises(KeyError):
        env.step(3)  # IntersectionEnv.ACTIONS has three entries
    assert env.vehicle.controlled and len(env.controlled_vehicles) == 1
    env.close()


def test_intersection_config_errors():
    assert EmuBatchedIntersection({"controlled_vehicles": 2}, num_envs=1).single_observation_shape == (2, 15, 7)
    with pytest.raises(ValueError):
        EmuBatchedIntersection({"controlled_vehicles": 5}, num_envs=1)  # one controlled vehicle per access road
    assert EmuBatchedIntersection({"destination": None}, num_envs=1)._hcfg.destination == -1  # random exit per episode
    with pytest.raises(ValueError):
        EmuBatchedIntersection({"destination": "o7"}, num_envs=1)
    with pytest.raises(NotImplementedError):
        EmuBatchedIntersection({"observation": {"type": "OccupancyGrid", "features": ["presence", "lat_off"]}}, num_envs=1)
    kin = dict(EmuBatchedIntersection.default_config()["observation"], observe_intentions=True,
               features=["presence", "x", "y", "cos_d", "sin_d"])
    assert EmuBatchedIntersection({"observation": kin}, num_envs=1)._hcfg.flags & _abi.C_OBS_INTENTIONS
    assert EmuBatchedIntersection({"observation": {"type": "OccupancyGrid"}}, num_envs=1).single_observation_shape == (4, 11, 11)
    env = EmuBatchedIntersection(num_envs=2)
    assert env.single_action_space.n == 3 and env.single_observation_shape == (15, 7)


@pytest.mark.parametrize("real", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["emu", "hip"])
def test_intersection_env_with_occupancy_grid_matches_reference(real):
    """BASELINE config 4: IntersectionEnv({"observation": {"type": "OccupancyGrid"}}) -- reset(seed=s) and the first
    steps give the reference's 4 x 11 x 11 grids (presence, vx, vy, on_road over straight and circular lanes)."""
    from tests.golden_util import GoldenIntersection
    g = GoldenIntersection("intersection_grid")
    cls = envs.IntersectionEnv if real else EmuIntersection
    env = cls({"observation": {"type": "OccupancyGrid"}})
    e = 1
    obs, info = env.reset(seed=int(g.z["seeds"][e]))
    assert obs.shape == (4, 11, 11) and obs.dtype == np.float32
    np.testing.assert_allclose(obs, g.z["obs0"][e], atol=1e-6)
    assert obs[3].sum() > 10  # the on-road layer is painted
    for t in range(4):
        obs, r, te, tr, info = env.step(int(g.actions[t, e, 0]))
        np.testing.assert_allclose(obs, g.z["obs"][t, e], atol=1e-6, err_msg=f"step {t}")
        assert abs(r - g.z["reward"][t, e]) < 1e-9
    env.close()


def test_registry_mirrors_the_reference_ids():
    """highway_env/__init__.py registers id -> entry-point class; the same ids resolve to the drop-in classes here and
    the class names match the reference's entry points."""
    ref = {"highway-v0": "HighwayEnv", "highway-fast-v0": "HighwayEnvFast", "merge-v0": "MergeEnv",
           "merge-v1": "ConnectedLaneMergeEnv", "merge-generic-v0": "MergeGenericEnv",
           "merge-generic-v1": "ConnectedLaneMergeGenericEnv", "intersection-v0": "IntersectionEnv",
           "intersection-v2": "ConnectedLaneIntersectionEnv", "intersection-multi-agent-v0": "MultiAgentIntersectionEnv",
           # the reference registers these two as MultiAgentIntersectionEnv / ConnectedLaneMultiAgentIntersectionEnv UNDER its
           # MultiAgentWrapper (highway_env/__init__.py:76-85); the drop-in classes fold the wrapper in
           "intersection-multi-agent-v1": "MultiAgentIntersectionEnvV1", "intersection-multi-agent-v2": "MultiAgentIntersectionEnvV2"}
    assert {k: v[0].__name__ for k, v in envs.REGISTRY.items()} == ref
    assert issubclass(envs.MultiAgentIntersectionEnvV1, envs.MultiAgentIntersectionEnv)
    assert issubclass(envs.MultiAgentIntersectionEnvV2, envs.ConnectedLaneMultiAgentIntersectionEnv)
    for env_id, (single, batched) in envs.REGISTRY.items():
        # (the wrapped ids: single and batched classes each fold the wrapper over the same batched engine class)
        assert issubclass(single, batched if "agent-v1" not in env_id and "agent-v2" not in env_id else batched.__mro__[2])
        connected = env_id.endswith(("-v1", "-v2")) and env_id != "intersection-multi-agent-v1"
        assert single.default_config()["neighbour_vehicles_connected_lanes"] is connected
    with pytest.raises(KeyError):
        envs.make("parking-v0")
    with pytest.raises(RuntimeError):  # no GPU in the build container: loud, no fallback
        if __import__("torch").cuda.is_available():
            raise RuntimeError("skip")
        envs.make("merge-v1")


def test_gymnasium_registration_when_gymnasium_is_importable():
    """highway_env registers its ids on import (highway_env/__init__.py:22-190).  gymnasium is not installed in the build
    image, so a recording stand-in is put in sys.modules of a fresh interpreter: importing highwayenv_amd must register the
    drop-ins under the "highwayenv_amd/" namespace, entry points must resolve, and the classes must be gymnasium.Env's."""
    import subprocess
    import sys
    code = r'''
import sys, types, importlib, importlib.machinery, importlib.util
gym = types.ModuleType("gymnasium"); gym.__spec__ = importlib.machinery.ModuleSpec("gymnasium", None)
class Env: pass
gym.Env = Env
spaces = types.ModuleType("gymnasium.spaces")
spaces.Discrete = lambda n: ("Discrete", n)
spaces.Box = lambda *a, **k: ("Box", a)
gym.spaces = spaces
envs = types.ModuleType("gymnasium.envs"); reg = types.ModuleType("gymnasium.envs.registration")
reg.registry = {}
def register(id, entry_point, **kw): reg.registry[id] = entry_point
reg.register = register
envs.registration = reg; gym.envs = envs
sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs, "gymnasium.envs.registration": reg})
import highwayenv_amd
from highwayenv_amd import envs as E
want = {"highwayenv_amd/" + k for k in E.REGISTRY}
assert set(reg.registry) == want, (sorted(reg.registry), sorted(want))
for k, ep in reg.registry.items():
    mod, cls = ep.split(":")
    c = getattr(importlib.import_module(mod), cls)
    assert issubclass(c, Env) and c is E.REGISTRY[k.split("/", 1)[1]][0]
assert highwayenv_amd.register_envs() == []                      # idempotent
assert set(E.register_envs(None)) == set(E.REGISTRY)              # the reference's bare ids on request
print("REGISTERED", len(reg.registry))
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "REGISTERED 22" in r.stdout


def test_action_tables_longitudinal_or_lateral_only():
    """DiscreteMetaAction(longitudinal=False) -> 3 ids indexing ACTIONS_LAT; neither -> the reference's ValueError
    (action.py:246-249); id 3 -> KeyError like self.actions[int(action)] (action.py:260)."""
    env = EmuFast({"action": {"type": "DiscreteMetaAction", "longitudinal": False}, "vehicles_count": 8}, num_envs=2)
    env.reset(seed=0)
    assert env.single_action_space.n == 3
    env.step([0, 2])
    with pytest.raises(KeyError):
        env.step([3, 1])
    with pytest.raises(ValueError, match="At least longitudinal or lateral"):
        envs.BatchedHighwayEnv({"action": {"type": "DiscreteMetaAction", "longitudinal": False, "lateral": False}})
    assert _abi.make_config(dict(_abi.highway_default_config(), action={"type": "DiscreteMetaAction", "lateral": False}), 1).action_set == _abi.ACTIONS_SET_LONGI


@pytest.mark.parametrize("real", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["emu", "hip"])
def test_shuffled_observation_order_replays_the_reference_stream(real):
    """KinematicObservation(order="shuffled"): the kernel returns the first vehicles_count - 1 eligible vehicles in LIST
    order (close_objects_to(sort=False)), the host shuffles rows 1.. on each environment's own generator -- the same
    stream the reference's reset(seed=s) spawned the traffic from, so the observations are the reference's row for row."""
    g = Golden("fast_shuffled")
    assert g.config["observation"]["order"] == "shuffled"
    cls = envs.BatchedHighwayEnvFast if real else EmuFast
    env = cls(g.config, num_envs=g.E)
    assert env._hcfg.flags & _abi.C_OBS_UNSORTED
    obs, _ = env.reset(seed=[int(s) for s in g.seeds])
    np.testing.assert_allclose(obs, g.z["obs0"], rtol=0, atol=1e-6)
    live = np.ones(g.E, bool)
    for t in range(g.steps):
        obs, reward, term, trunc, info = env.step(g.actions[t])
        np.testing.assert_allclose(obs[live], g.z["obs"][t][live], rtol=0, atol=1e-6, err_msg=f"step {t}")
        np.testing.assert_allclose(reward[live], g.z["reward"][t][live], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(term[live], g.z["terminated"][t].astype(bool)[live])
        live &= ~(term | trunc)
        if not live.any():
            break
    env.close()


@pytest.mark.parametrize("real", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["emu", "hip"])
def test_random_destination_replays_the_reference_stream(real):
    """config["destination"] = None: the reference draws "o" + str(np_random.integers(1, 4)) for the ego BEFORE its position
    (intersection_env.py:295-300).  reset(seed=s) must give the reference's ego route, first observation and episode."""
    from highwayenv_amd import intersection as hix
    from tests.golden_util import GoldenIntersection
    g = GoldenIntersection("intersection_random_destination")
    assert g.config["destination"] is None
    dests = set()
    for e in range(g.E):
        env = (envs.IntersectionEnv if real else EmuIntersection)({"destination": None})
        obs, info = env.reset(seed=int(g.z["seeds"][e]))
        np.testing.assert_allclose(obs, g.z["obs0"][e], atol=1e-6)
        st = env.get_state()
        ego = int(np.argmax((st["flags"][0] & _abi.F_CONTROLLED) != 0))
        route = hix.route_unpack(int(st["route"][0, ego]))
        tab = hix.table_from_config(env._hcfg)
        init = g.state("init")
        want = [(int(init["route_from"][e, ego, q]), int(init["route_to"][e, ego, q])) for q in range(int(init["route_len"][e, ego]))]
        assert [(int(tab["from_node"][l]), int(tab["to_node"][l])) for l in route] == want
        dests.add(want[-1][1])
        for t in range(3):
            wst = g.state("step", t)
            pres = wst["present"][e] != 0
            if ((wst["crashed"][e] != 0) | (wst["has_impact"][e] != 0))[pres].any() or (np.abs(wst["speed"][e][pres]) < 0.5).any():
                break
            obs, r, te, tr, info = env.step(int(g.actions[t, e, 0]))
            np.testing.assert_allclose(obs, g.z["obs"][t, e], atol=1e-6, err_msg=f"env {e} step {t}")
            assert abs(r - g.z["reward"][t, e]) < 1e-9
            if te or tr:
                break
        env.close()
    assert len(dests) >= 2  # the draw really varies over the seeds


# ---- intersection-multi-agent-v0 (MultiAgentIntersectionEnv, intersection_env.py:348-399) -----------------------------
class EmuMultiAgentIntersection(envs._SingleIntersectionMixin, envs.BatchedMultiAgentIntersectionEnv):
    _engine_factory = staticmethod(_emu_factory)


def test_multi_agent_intersection_default_config_is_the_references():
    from tests.golden_util import GoldenIntersection
    g = GoldenIntersection("intersection_multi_agent")
    mine = envs.BatchedMultiAgentIntersectionEnv.default_config()
    for k, v in g.config.items():  # the reference's MultiAgentIntersectionEnv.default_config(), as recorded
        if k not in ("screen_width", "screen_height", "centering_position", "scaling", "offscreen_rendering"):  # (rendering / test harness)
            assert mine[k] == v, k
    assert envs.REGISTRY["intersection-multi-agent-v0"][0] is envs.MultiAgentIntersectionEnv
    with pytest.raises(ValueError, match="1..4 controlled vehicles"):
        envs.BatchedMultiAgentIntersectionEnv({"controlled_vehicles": 5})


class EmuMultiAgentIntersectionV1(envs._MultiAgentWrapperMixin, EmuMultiAgentIntersection):
    pass


def test_multi_agent_wrapper_ids_hand_out_the_per_agent_tuples():
    """intersection-multi-agent-v1 (MultiAgentWrapper, abstract.py:468-477): reward = info["agents_rewards"], terminated =
    info["agents_terminated"] -- the reference's recorded per-agent values."""
    from tests.golden_util import GoldenIntersection
    g = GoldenIntersection("intersection_multi_agent")
    env = EmuMultiAgentIntersectionV1(dict(g.config))
    env.reset(seed=int(g.z["seeds"][1]))
    obs, reward, terminated, truncated, info = env.step(tuple(int(a) for a in g.actions[0, 1]))
    assert isinstance(reward, tuple) and isinstance(terminated, tuple) and len(reward) == g.A
    np.testing.assert_allclose(reward, g.z["agents_rewards"][0, 1], atol=1e-9)
    assert terminated == tuple(bool(b) for b in g.z["agents_terminated"][0, 1]) and truncated == bool(g.z["truncated"][0, 1])
    assert envs.REGISTRY["intersection-multi-agent-v1"][0].__mro__[1] is envs._MultiAgentWrapperMixin
    env.close()


@pytest.mark.parametrize("real", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["emu", "hip"])
@pytest.mark.parametrize("name", ["intersection_multi_agent", "intersection_multi_agent3"])
def test_multi_agent_intersection_dropin_matches_reference_episode(real, name):
    """MultiAgentIntersectionEnv(config): reset(seed=s) spawns the A controlled vehicles on the reference's numpy stream (agent
    k on road o{k % 4}; with destination None an integers(1, 4) draw before each position), then the golden tuple actions:
    the reference's stacked observations, mean reward, info["agents_rewards"] / ["agents_terminated"], terminated."""
    from tests.golden_util import GoldenIntersection
    g = GoldenIntersection(name)
    A = g.A
    steps_compared = 0
    for e in range(g.E):
        env = (envs.MultiAgentIntersectionEnv if real else EmuMultiAgentIntersection)(dict(g.config))
        obs, info = env.reset(seed=int(g.z["seeds"][e]))
        assert obs.shape == (A, 15, 7) and len(env.controlled_vehicles) == A
        np.testing.assert_allclose(obs, g.z["obs0"][e], atol=1e-6)
        for t in range(g.steps):
            wst = g.state("step", t)
            pres = wst["present"][e] != 0
            stalled = (np.abs(wst["speed"][e][pres]) < 0.5).any()
            wreck = ((wst["crashed"][e] != 0) | (wst["has_impact"][e] != 0))[pres].any()
            if stalled:
                break
            obs, r, te, tr, info = env.step(tuple(int(a) for a in g.actions[t, e]))
            what = f"{name} env {e} step {t}"
            assert te == bool(g.z["terminated"][t, e]) and tr == bool(g.z["truncated"][t, e]), what
            assert info["agents_terminated"] == tuple(bool(b) for b in g.z["agents_terminated"][t, e]), what
            if wreck:
                break
            np.testing.assert_allclose(obs, g.z["obs"][t, e], atol=1e-6, err_msg=what)
            assert abs(r - g.z["reward"][t, e]) < 1e-9, what
            np.testing.assert_allclose(info["agents_rewards"], g.z["agents_rewards"][t, e], atol=1e-9, err_msg=what)
            steps_compared += 1
            if te or tr:
                break
        env.close()
    assert steps_compared >= 2 * g.E


@pytest.mark.parametrize("real", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["emu", "hip"])
@pytest.mark.parametrize("name", ["intersection_intentions", "intersection_no_intentions"])
def test_destination_features_replay_the_reference(real, name):
    """cos_d / sin_d (Vehicle.destination_direction): IntersectionEnv(config).reset(seed=s) and the first steps give the reference's
    observations -- with observe_intentions every observed vehicle's destination, without it only the observer's own."""
    from tests.golden_util import GoldenIntersection
    g = GoldenIntersection(name)
    n_rows = 0
    for e in range(g.E):
        env = (envs.IntersectionEnv if real else EmuIntersection)(dict(g.config))
        obs, info = env.reset(seed=int(g.z["seeds"][e]))
        np.testing.assert_allclose(obs, g.z["obs0"][e], atol=1e-6)
        dest = obs[:, -2:]
        n_rows += int((np.abs(dest).sum(1) > 0).sum())
        for t in range(3):
            wst = g.state("step", t)
            pres = wst["present"][e] != 0
            if ((wst["crashed"][e] != 0) | (wst["has_impact"][e] != 0))[pres].any() or (np.abs(wst["speed"][e][pres]) < 0.5).any():
                break
            obs, r, te, tr, info = env.step(int(g.actions[t, e, 0]))
            np.testing.assert_allclose(obs, g.z["obs"][t, e], atol=1e-6, err_msg=f"env {e} step {t}")
            if te or tr:
                break
        env.close()
    assert n_rows > g.E if g.config["observation"]["observe_intentions"] else n_rows == g.E


def test_occupancy_grid_as_image_is_uint8_like_the_reference():
    """OccupancyGridObservation(as_image=True): Box(0, 255, uint8) cells, ((clip(v, -1, 1) + 1) / 2 * 255).astype(uint8), an
    empty cell 0 (observation.py:330-331, 408-409) -- reset(seed=s) gives the reference's first image."""
    g = Golden("grid_image")
    env = EmuFast(g.config, num_envs=g.E)
    obs, _ = env.reset(seed=[int(s) for s in g.seeds])
    assert obs.dtype == np.uint8 and env.single_observation_space.dtype == np.uint8
    np.testing.assert_array_equal(obs, g.z["obs0"])
    obs, *_ = env.step(g.actions[0])
    np.testing.assert_array_equal(obs, g.z["obs"][0])
    env.close()
