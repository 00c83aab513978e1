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
tot = np.zeros(24)
n = 0
rows = []
for t in range(60):
    obs = eng.step(rng.integers(0, 5, size=(E, 1)))[0]
    if t >= 40:
        r = obs.reshape(E, -1)[:, :24].astype(np.float64)
        r = r[r[:, 2] > 0]  # (auto-resetting wavefronts write a real observation there: x of the first row is 1.0, y 0)
        rows.append(r)
        tot += r.mean(0)
        n += 1
names = ["load", "A meta-action", "C rank check", "C snapshot + masks", "D neighbour ranks", "D free road + own gap   ",
         "D MOBIL tasks    ", "D abort chain", "E+F control + integrate", "G collisions", "H observe"]
tot /= n
for k, nm in enumerate(names):
    print(f"{nm:26s} {tot[k]:10.0f} ticks/step/wave  {100 * tot[k] / tot[:11].sum():5.1f}%")
print(f"{'total':26s} {tot[:11].sum():10.0f}")
print(f"(inside G: publish {tot[11]:.0f}, walk {tot[12]:.0f}, list passes {tot[13]:.0f} ticks; per step: {tot[14]:.2f} walk trips, "
      f"{tot[15]:.2f} list passes with pairs, {tot[18]:.2f} pairs)")

# the launch lasts as long as its slowest wavefront: where do the slow ones spend their time?
allr = np.concatenate(rows)
life = allr[:, :14].sum(1)  # sections 0..10 + the three inner collision clocks
order = np.argsort(life)
top = order[-max(1, len(order) // 100):]
print(f"wavefront lifetime (ticks): mean {life.mean():.0f}, p50 {np.percentile(life, 50):.0f}, p90 {np.percentile(life, 90):.0f}, "
      f"p99 {np.percentile(life, 99):.0f}, max {life.max():.0f}")
print("slowest 1 % of the wavefronts, section means (ticks) against the overall means:")
for k, nm in enumerate(names + ["G publish", "G walk", "G list passes"]):
    print(f"  {nm:26s} {allr[top, k].mean():9.0f}   {allr[:, k].mean():9.0f}")
print(f"  abort chain per step: changers {allr[top, 16].mean():.1f} / {allr[:, 16].mean():.1f}, frames with a chain {allr[top, 17].mean():.1f} / "
      f"{allr[:, 17].mean():.1f}, walk trips {allr[top, 20].mean():.1f} / {allr[:, 20].mean():.1f}, fixed-point rounds "
      f"{allr[top, 21].mean():.1f} / {allr[:, 21].mean():.1f}")
print(f"  collision walk trips {allr[top, 14].mean():.1f} / {allr[:, 14].mean():.1f}, list passes with pairs {allr[top, 15].mean():.1f} / "
      f"{allr[:, 15].mean():.1f}, pairs {allr[top, 18].mean():.1f} / {allr[:, 18].mean():.1f}")
