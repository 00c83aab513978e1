"""highwayenv_amd.vector.HighwayVectorEnv: the gymnasium.vector.VectorEnv-shaped front end (the reference's users vectorise with
gym.vector.SyncVectorEnv(..., autoreset_mode="SameStep"), /root/reference/tests/envs/test_gym.py:158-165).

CPU part: the wrapper over the emulated engine (interface, NextStep bookkeeping).  GPU part: the reference's vectorisation test
restated, SameStep vs NextStep on the same seeds, torch-resident outputs."""
import numpy as np
import pytest

from highwayenv_amd import envs, vector


def _emu_factory(cfg, device, stream):
    from tests.emu.emu import EmuEngine
    return EmuEngine(cfg)


class EmuBatchedFast(envs.BatchedHighwayEnvFast):
    _engine_factory = staticmethod(_emu_factory)


def test_interface_and_next_step_autoreset_on_the_emulator():
    cfg = {"vehicles_count": 8, "lanes_count": 2, "duration": 2}
    venv = vector.HighwayVectorEnv(EmuBatchedFast, num_envs=3, config=cfg, autoreset_mode="NextStep")
    assert venv.num_envs == 3 and venv.metadata["autoreset_mode"] == "NextStep" and venv.autoreset_mode == "NextStep"
    assert venv.single_observation_space.shape == (5, 5) and venv.observation_space.shape == (3, 5, 5)
    assert venv.single_action_space.n == 5 and venv.action_space.shape == (3,)
    obs, infos = venv.reset(seed=5)
    assert obs.shape == (3, 5, 5) and obs.dtype == np.float32
    assert np.issubdtype(infos["speed"].dtype, np.floating) and infos["crashed"].dtype == bool  # test_gym.py:166-167
    ended = None
    for t in range(4):
        venv.step_async(np.ones(3, np.int64))
        obs, reward, term, trunc, infos = venv.step_wait()
        assert obs.shape == (3, 5, 5) and reward.shape == (3,) and reward.dtype == np.float64
        assert term.dtype == bool and trunc.dtype == bool and np.issubdtype(infos["speed"].dtype, np.floating)
        if ended is not None:  # the step after an episode ended re-spawns: first observation, reward 0, flags down
            assert (reward[ended] == 0).all() and not term[ended].any() and not trunc[ended].any()
            assert (obs[ended, 0, 0] == 1.0).all()
        ended = term | trunc
        if t == 1:
            assert trunc.all() or term.any()   # duration 2: every surviving episode is truncated at the second step
    assert venv.call("num_envs") == (3, 3, 3)
    venv.close()
    with pytest.raises(ValueError):
        vector.HighwayVectorEnv(EmuBatchedFast, num_envs=2, autoreset_mode="sometimes")
    with pytest.raises(RuntimeError):
        vector.HighwayVectorEnv(EmuBatchedFast, num_envs=2).step_wait()


def _same_step_second_episodes(seed, steps=3):
    cfg = {"vehicles_count": 8, "lanes_count": 2, "duration": 2}
    venv = vector.HighwayVectorEnv(EmuBatchedFast, num_envs=3, config=cfg, autoreset_mode="SameStep", spawn_mode="device")
    out = []
    for _rep in range(2):   # reset(seed) twice in one process: the same continuation both times
        venv.reset(seed=seed)
        seen = []
        for _t in range(steps):
            obs, _r, term, trunc, _i = venv.step(np.ones(3, np.int64))
            if (term | trunc).any():
                seen.append(obs[term | trunc].copy())
        out.append(np.concatenate(seen))
    venv.close()
    np.testing.assert_array_equal(out[0], out[1])
    return out[0]


def test_same_step_respawn_seeds_follow_the_user_seed():
    """reset_done (SameStep): later episodes are a function of the seed given to reset() -- reproducible for one seed (also
    across two reset() calls of one process), different for two seeds, and not a copy of a first episode (round-3 advisor)."""
    a, b = _same_step_second_episodes(11), _same_step_second_episodes(12)
    assert a.shape == b.shape and len(a) >= 3
    assert not np.array_equal(a, b)
    np.testing.assert_array_equal(a, _same_step_second_episodes(11))
    first = vector.HighwayVectorEnv(EmuBatchedFast, num_envs=3, config={"vehicles_count": 8, "lanes_count": 2, "duration": 2},
                                    autoreset_mode="SameStep", spawn_mode="device")
    o11, _ = first.reset(seed=11)
    first.close()
    assert not any(np.array_equal(r, o) for r in a for o in o11)


def test_stream_argument_needs_the_torch_front_end():
    with pytest.raises(ValueError):
        vector.HighwayVectorEnv("highway-fast-v0", num_envs=2, stream=object())


def test_ids_resolve_to_the_batched_classes():
    assert envs.batched_class("highway-fast-v0") is envs.BatchedHighwayEnvFast
    assert envs.batched_class("highwayenv_amd/merge-v0") is envs.BatchedMergeEnv
    assert envs.batched_class("intersection-v0") is envs.BatchedIntersectionEnv


@pytest.mark.gpu
@pytest.mark.parametrize("env_id", ["highway-v0", "highway-fast-v0", "merge-v0", "intersection-v0"])
def test_reference_vectorisation_pattern_same_step(env_id):
    """/root/reference/tests/envs/test_gym.py:158-165 restated: autoreset_mode="SameStep", info dtypes, run until truncation."""
    venv = vector.HighwayVectorEnv(env_id, num_envs=2, config={"duration": 2, "simulation_frequency": 2},
                                   autoreset_mode="SameStep")
    _obs, info = venv.reset(seed=0)
    assert np.issubdtype(info["speed"].dtype, np.floating)
    zero_action = np.zeros(venv.action_space.shape, np.int64)
    seen_final = False
    for _step in range(3):
        obs, _reward, terminated, truncated, info = venv.step(zero_action)
        assert np.issubdtype(info["speed"].dtype, np.floating)
        done = terminated | truncated
        if done.any():
            seen_final = True
            assert (info["_final_obs"] == done).all() and (info["_final_info"] == done).all()
            for e in np.nonzero(done)[0]:
                assert info["final_obs"][e].shape == obs[e].shape and "speed" in info["final_info"][e]
            assert all(info["final_obs"][e] is None for e in np.nonzero(~done)[0])
    assert seen_final or env_id == "merge-v0"   # (MergeEnv never truncates, merge_env.py:85-86: nothing ends within 3 steps)
    venv.close()


@pytest.mark.gpu
def test_same_step_equals_next_step_shifted():
    """The two modes describe the same episodes: SameStep's final_obs at step t == NextStep's observation at step t, and
    NextStep spends one extra call per episode boundary."""
    cfg = {"vehicles_count": 20, "lanes_count": 3, "duration": 3}
    a = vector.HighwayVectorEnv("highway-fast-v0", num_envs=8, config=cfg, autoreset_mode="NextStep")
    b = vector.HighwayVectorEnv("highway-fast-v0", num_envs=8, config=cfg, autoreset_mode="SameStep")
    oa, _ = a.reset(seed=3)
    ob, _ = b.reset(seed=3)
    np.testing.assert_array_equal(oa, ob)
    acts = np.ones(8, np.int64)
    for t in range(3):   # the first episodes are the same in both modes, terminal observation included
        oa, ra, ta, tra, _ = a.step(acts)
        ob, rb, tb, trb, ib = b.step(acts)
        np.testing.assert_array_equal(ta, tb)
        np.testing.assert_array_equal(tra, trb)
        np.testing.assert_array_equal(ra, rb)
        done = tb | trb
        np.testing.assert_array_equal(oa[~done], ob[~done])
        for e in np.nonzero(done)[0]:
            np.testing.assert_array_equal(ib["final_obs"][e], oa[e])
        if done.any():
            break
    a.close()
    b.close()


@pytest.mark.gpu
def test_torch_outputs_alias_the_device_buffers_and_match_numpy():
    import torch
    cfg = {"vehicles_count": 20, "lanes_count": 3, "duration": 4}
    n = vector.HighwayVectorEnv("highway-fast-v0", num_envs=16, config=cfg)
    t = vector.HighwayVectorEnv("highway-fast-v0", num_envs=16, config=cfg, output="torch")   # engine created BEFORE torch touches the GPU
    on, _ = n.reset(seed=9)
    ot, _ = t.reset(seed=9)
    assert ot.is_cuda
    np.testing.assert_array_equal(ot.cpu().numpy(), on)
    rng = np.random.default_rng(0)
    for k in range(6):
        acts = rng.integers(0, 5, size=16)
        on, rn, tn, trn, inf_n = n.step(acts)
        ot, rt, tt, trt, inf_t = t.step(torch.as_tensor(acts, device="cuda"))
        assert ot.is_cuda and rt.is_cuda and tt.dtype == torch.bool
        np.testing.assert_array_equal(ot.cpu().numpy(), on)
        np.testing.assert_array_equal(rt.cpu().numpy(), rn)
        np.testing.assert_array_equal(tt.cpu().numpy(), tn)
        np.testing.assert_array_equal(trt.cpu().numpy(), trn)
        np.testing.assert_array_equal(inf_t["speed"].cpu().numpy(), inf_n["speed"])
    # an environment ON a caller's stream (stream=): no cross-stream ordering while that stream is current -- same results
    lane = torch.cuda.Stream()
    n2 = vector.HighwayVectorEnv("highway-fast-v0", num_envs=16, config=cfg)
    t2 = vector.HighwayVectorEnv("highway-fast-v0", num_envs=16, config=cfg, output="torch", stream=lane)
    assert t2.stream is lane
    n2.reset(seed=9)
    t2.reset(seed=9)
    with torch.cuda.stream(lane):
        for k in range(4):
            acts = rng.integers(0, 5, size=16)
            on, rn, tn, _, _ = n2.step(acts)
            ot, rt, tt, _, _ = t2.step(torch.as_tensor(acts, device="cuda", dtype=torch.int32))
            np.testing.assert_array_equal(ot.cpu().numpy(), on)
            np.testing.assert_array_equal(rt.cpu().numpy(), rn)
            np.testing.assert_array_equal(tt.cpu().numpy(), tn)
    n2.close()
    t2.close()
    # K steps in one launch from the torch front end == K steps
    acts = rng.integers(0, 5, size=(5, 16))
    outs = [n.step(acts[k]) for k in range(5)]
    obs, rew, term, trunc = t.rollout(torch.as_tensor(acts, device="cuda"))
    for k in range(5):
        np.testing.assert_array_equal(obs[k].cpu().numpy(), outs[k][0])
        np.testing.assert_array_equal(rew[k].cpu().numpy(), outs[k][1])
        np.testing.assert_array_equal(term[k].cpu().numpy(), outs[k][2])
    n.close()
    t.close()
