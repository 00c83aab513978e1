"""Randomised configurations of the straight-road scenarios against the oracle (MI355X only: 40 configurations x 8
envs x 8 steps take seconds there and minutes on the CPU emulation).  Lanes, traffic size, agents, frequencies, speed
ladders, observation variants, reward / termination switches are drawn from a seeded generator; the comparison is the
one of tests/test_edge_cases.py (flags exact, obs 1e-6, reward 1e-9, state 1e-7 on live, wreck-free episodes)."""
import os

import numpy as np
import pytest

from highwayenv_amd import _abi
from tests.test_edge_cases import rollout

pytestmark = pytest.mark.gpu


def random_config(rng):
    fast = bool(rng.integers(2))
    cfg = _abi.highway_fast_default_config() if fast else _abi.highway_default_config()
    lanes = int(rng.integers(1, 7))
    agents = int(rng.integers(1, 4))
    sim = int(rng.choice([5, 10, 15]))
    feats = [["presence", "x", "y", "vx", "vy"], ["presence", "x", "y", "vx", "vy", "cos_h", "sin_h"],
             ["x", "y", "heading", "vx"]][int(rng.integers(3))]
    obs = {"type": "Kinematics", "vehicles_count": int(rng.integers(2, 9)), "features": feats,
           "absolute": bool(rng.integers(2)), "see_behind": bool(rng.integers(2)), "normalize": bool(rng.integers(2)),
           "clip": bool(rng.integers(2))}
    cfg.update({"lanes_count": lanes, "vehicles_count": int(rng.integers(0, 70)), "controlled_vehicles": agents,
                "simulation_frequency": sim, "policy_frequency": int(rng.choice([1, 5]) if sim % 5 == 0 else 1),
                "duration": int(rng.integers(4, 12)), "vehicles_density": float(rng.uniform(0.8, 2.0)),
                "ego_spacing": float(rng.uniform(1.0, 2.5)), "normalize_reward": bool(rng.integers(2)),
                "offroad_terminal": bool(rng.integers(2)), "collision_reward": float(rng.uniform(-2, 0)),
                "right_lane_reward": float(rng.uniform(0, 0.5)), "high_speed_reward": float(rng.uniform(0.1, 1)),
                "observation": obs if agents == 1 else {"type": "MultiAgentObservation", "observation_config": obs}})
    if agents > 1:
        cfg["action"] = {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}}
    if rng.integers(3) == 0:
        tgt = {"type": "DiscreteMetaAction", "target_speeds": sorted(rng.uniform(10, 35, size=int(rng.integers(2, 6))).tolist())}
        cfg["action"] = tgt if agents == 1 else {"type": "MultiAgentAction", "action_config": tgt}
    return cfg, fast


@pytest.mark.parametrize("chunk", range(int(os.environ.get("HWY_FUZZ_CHUNKS", "4"))))  # 10 configurations each
def test_random_configurations_vs_oracle(chunk):
    rng = np.random.default_rng(9000 + chunk)
    for k in range(10):
        cfg, fast = random_config(rng)
        try:
            rollout("hip", cfg, fast, E=8, steps=8, seed=chunk * 100 + k)
        except AssertionError as ex:  # name the configuration in the failure
            raise AssertionError(f"chunk {chunk} config {k}: {cfg}\n{ex}") from ex


def random_merge_config(rng):
    from highwayenv_amd import merge
    cfg = merge.merge_generic_default_config()
    lanes = int(rng.integers(1, 5))
    n = int(rng.integers(0, 50))
    agents = int(rng.integers(1, min(4, n + 1) + 1))
    cfg.update({"lanes_count": lanes, "vehicles_count": n, "controlled_vehicles": agents,
                "before_merge_length": int(rng.integers(50, 200)), "converge_merge_length": int(rng.integers(40, 120)),
                "parallel_merge_length": int(rng.integers(40, 120)), "after_merge_length": int(rng.integers(90, 200)),
                "simulation_frequency": int(rng.choice([5, 15])), "collision_reward": float(rng.uniform(-2, -0.1)),
                "merging_speed_reward": float(rng.uniform(-1, -0.1)), "lane_change_reward": float(rng.uniform(-0.2, 0))})
    if agents > 1:
        cfg["action"] = {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}}
        cfg["observation"] = {"type": "MultiAgentObservation", "observation_config": {"type": "Kinematics"}}
    return cfg


@pytest.mark.parametrize("chunk", range(int(os.environ.get("HWY_FUZZ_CHUNKS", "4"))))
def test_random_merge_configurations_vs_oracle(chunk):
    from tests.test_net_parity import _rollout_vs_oracle
    rng = np.random.default_rng(7000 + chunk)
    for k in range(6):
        cfg = random_merge_config(rng)
        try:
            _rollout_vs_oracle("hip", cfg, "merge-generic", E=6, steps=10, seed=chunk * 100 + k + 1)
        except AssertionError as ex:
            raise AssertionError(f"chunk {chunk} config {k}: {cfg}\n{ex}") from ex
