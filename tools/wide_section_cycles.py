#!/usr/bin/env python3
"""Developer tool (GPU): per-section s_memtime cycle breakdown of the two-vehicles-per-thread step kernel (hwy_wave2.h), using the
instrumented `w2ticks` build from tools/ablate/make_variants.py (HWY_ENGINE_LIB must point to it).
    python tools/wide_section_cycles.py [envs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg_d = _abi.highway_default_config()
cfg_d.update({"vehicles_count": 100})
cfg = _abi.make_config(cfg_d, E, fast=False)
eng = Engine(cfg)
eng.reset(base_seed=5, ego_spacing=2.0, vehicles_density=1.0)
eng.set_autoreset(True, base_seed=99, ego_spacing=2.0, vehicles_density=1.0)  # the bench's steady-state workload
rng = np.random.default_rng(0)
tot = np.zeros(16)
n = 0
for t in range(60):
    obs = eng.step(rng.integers(0, 5, size=(E, 1)))[0]
    if t >= 40:
        tot += obs.reshape(E, -1)[:, :16].astype(np.float64).mean(0)
        n += 1
names = ["load", "A meta-action", "C rank check", "C snapshot + masks", "D neighbour ranks", "D gaps + mobil incentive",
         "D follower safety", "D abort chain", "E+F control + integrate", "G collisions", "H observe"]
tot /= n
for k, nm in enumerate(names):
    print(f"{nm:26s} {tot[k]:10.0f} ticks/step/wave  {100 * tot[k] / tot[:11].sum():5.1f}%")
print(f"{'total':26s} {tot[:11].sum():10.0f}")
print(f"(inside G: publish + walk {tot[11]:.0f}, SAT passes {tot[12]:.0f} ticks; per step: {tot[13]:.2f} walk trips, "
      f"{tot[14]:.2f} SAT passes with pairs, {tot[15]:.2f} pairs)")
