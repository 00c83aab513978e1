#!/usr/bin/env python3
"""Developer tool (GPU): per-section s_memtime cycle breakdown of the step kernel, using the
instrumented `ticks` build from tools/ablate/make_variants.py (HWY_ENGINE_LIB must point to it)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

E = 4096
cfg_d = _abi.highway_fast_default_config()
cfg_d.update({"vehicles_count": 50, "lanes_count": 4})
cfg = _abi.make_config(cfg_d, E, fast=True)
eng = Engine(cfg)
eng.reset(base_seed=5, ego_spacing=1.5, vehicles_density=1.0)
eng.set_autoreset(True, base_seed=99, ego_spacing=1.5, vehicles_density=1.0)  # the bench's steady-state workload
rng = np.random.default_rng(0)
tot = np.zeros(13)
n = 0
for t in range(60):
    obs = eng.step(rng.integers(0, 5, size=(E, 1)))[0]
    if t >= 40:
        tot += obs.reshape(E, -1)[:, :13].astype(np.float64).mean(0)
        n += 1
names = ["load+chk", "A meta-action", "C rank check/recount", "C membership+snapshot", "D neighbour ranks+gather",
         "D free road + gaps", "D mobil", "D abort chain", "E control", "F integrate", "G collisions", "H observe"]
tot /= n
for k, nm in enumerate(names):
    print(f"{nm:22s} {tot[k]:10.0f} cycles/step/wave  {100 * tot[k] / tot[:12].sum():5.1f}%")
print(f"{'total':22s} {tot[:12].sum():10.0f}")
print("frames whose rank order changed, per step (of 5 frames):", tot[12])
