#!/usr/bin/env python3
"""Developer tool (GPU): per-section s_memtime cycle breakdown of hwy_net_step_kernel on the merge workloads, using
the `nticks` build from tools/ablate/make_variants.py (HWY_ENGINE_LIB must point to it)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd import _abi, merge  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

E = 4096
which = sys.argv[1] if len(sys.argv) > 1 else "merge_ma4"
if which == "merge":
    scenario, cfg_d = "merge", merge.merge_default_config()
else:
    scenario, cfg_d = "merge-generic", merge.merge_generic_default_config()
    cfg_d.update({"lanes_count": 4, "vehicles_count": 40, "controlled_vehicles": 4,
                  "action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
                  "observation": {"type": "MultiAgentObservation", "observation_config": {"type": "Kinematics"}}})
cfg = _abi.make_config(cfg_d, E, scenario=scenario)
eng = Engine(cfg)
eng.reset(base_seed=5)
eng.set_autoreset(True, base_seed=99)
rng = np.random.default_rng(0)
tot = np.zeros(16)
n = 0
per_wave = []
for t in range(40):
    obs = eng.step(rng.integers(0, 5, size=(E, cfg.num_agents)))[0]
    if t >= 20:
        rows = obs.reshape(E, -1)[:, :16].astype(np.float64)
        tot += rows.mean(0)
        per_wave.append(rows)
        n += 1
names = ["load", "A meta-action", "B rank", "B membership+snapshot", "C follow_road+neighbours", "C gaps + MOBIL",
         "C abort chain", "D control", "E integrate", "E closest lane", "F collisions: walk", "G observe",
         "F collisions: list passes"]
tot /= n
print(which)
for k, nm in enumerate(names):
    print(f"{nm:26s} {tot[k]:10.0f} cycles/step/wave  {100 * tot[k] / tot[:13].sum():5.1f}%")
print(f"{'total':26s} {tot[:13].sum():10.0f}")
print(f"walk steps per step {tot[13]:.1f}, list passes with pairs per step {tot[14]:.1f}, waves that ran a SAT {tot[15]:.3f}")
X = np.concatenate(per_wave)
X = X[(X[:, :13] >= 0).all(1) & (X[:, :13].sum(1) > 1000)]  # (re-spawning waves hold observation floats there)
T = X[:, :13].sum(1)
print("per-wave total: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (T.mean(), *np.percentile(T, [50, 90, 99]), T.max()))
for k, nm in enumerate(names):
    c = X[:, k]
    print(f"{nm:26s} p50 {np.percentile(c, 50):8.0f} p90 {np.percentile(c, 90):8.0f} p99 {np.percentile(c, 99):8.0f} max {c.max():8.0f}")
