import sys, numpy as np
sys.path.insert(0, '/root/repo')
from highwayenv_amd import _abi, merge
from highwayenv_amd.engine import Engine
E = 4096
cfg_d = merge.merge_generic_default_config()
cfg_d.update({"lanes_count": 4, "vehicles_count": 40, "controlled_vehicles": 4,
              "action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
              "observation": {"type": "MultiAgentObservation", "observation_config": {"type": "Kinematics"}}})
cfg = _abi.make_config(cfg_d, E, scenario="merge-generic")
eng = Engine(cfg); eng.reset(base_seed=5); eng.set_autoreset(True, base_seed=99)
rng = np.random.default_rng(0)
R, S = [], []
for t in range(60):
    obs = eng.step(rng.integers(0, 5, size=(E, cfg.num_agents)))[0].reshape(E, -1)
    if t >= 20:
        isr = obs[:, 15] == -1.0
        R.append(obs[isr, 0].astype(np.float64)); S.append(obs[~isr, :13].astype(np.float64).sum(1))
R, S = np.concatenate(R), np.concatenate(S)
S = S[S > 1000]
print("reset waves per launch %.0f of %d (%.1f %%)" % (len(R) / 40, E, 100 * len(R) / 40 / E))
print("reset wave lifetime ticks: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (R.mean(), *np.percentile(R, [50, 90, 99]), R.max()))
print("step  wave lifetime ticks: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (S.mean(), *np.percentile(S, [50, 90, 99]), S.max()))
