#!/usr/bin/env python3
"""Developer tool (GPU): per-section s_memtime ticks of the intersection step kernel (the `ixticks` build of
tools/ablate/make_variants.py; HWY_ENGINE_LIB must point to it) on the bench workload, per wave ROLE.

    HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_ixticks.so python tools/ix_section_dist.py [envs] [key=value tuning ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tuning = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}
cfg_d, fast, scenario = bench.workload_config("intersection")
cfg = _abi.make_config(cfg_d, E, fast=fast, scenario=scenario, tuning=tuning)
eng = Engine(cfg)
eng.reset(base_seed=5)
eng.set_autoreset(True, base_seed=99)
rng = np.random.default_rng(0)
rows = []
for t in range(70):
    obs, _, term, trunc, _ = eng.step(rng.integers(0, 3, size=(E, 1)))
    if t >= 40:
        rows.append(obs.reshape(E, -1)[:, :16].astype(np.float64))
X = np.concatenate(rows)
names = ["setup (load, first table walk)", "A+B action, masks, snapshot", "C Road.act", "D regulation: partner loop", "E integrate",
         "table walk: final exchange", "F collisions", "observe", "clear + spawn", "spawn finalise", "store",
         "table walk: prologue + straight", "table walk: exchange + arcs", "D regulation: samples + circles"]
for role, rn in ((0, "STEP"), (1, "RESPAWN")):
    S = X[X[:, 14] == role]
    if not len(S):
        continue
    tot = S[:, :14].sum(1)
    print(f"--- role {rn}: {len(S)} waves ({100 * len(S) / len(X):.1f} %), frames run mean {S[:, 15].mean():.1f} max {S[:, 15].max():.0f}; "
          f"ticks per launch per wave (s_memtime)")
    print(f"{'section':32s} {'mean':>9s} {'p50':>9s} {'p90':>9s} {'p99':>9s} {'max':>9s} {'share':>6s}")
    for k, nm in enumerate(names):
        c = S[:, k]
        print(f"{nm:32s} {c.mean():9.0f} {np.percentile(c, 50):9.0f} {np.percentile(c, 90):9.0f} {np.percentile(c, 99):9.0f} "
              f"{c.max():9.0f} {100 * c.mean() / tot.mean():5.1f}%")
    print(f"{'total':32s} {tot.mean():9.0f} {np.percentile(tot, 50):9.0f} {np.percentile(tot, 90):9.0f} {np.percentile(tot, 99):9.0f} {tot.max():9.0f}")
    # the launch lasts as long as its slowest wavefront: where do the slowest ones spend their time?
    top = np.argsort(tot)[-max(1, len(tot) // 100):]
    print("slowest 1 % of these wavefronts, section means against the overall means:")
    for k, nm in enumerate(names):
        print(f"  {nm:32s} {S[top, k].mean():9.0f} {S[:, k].mean():9.0f}")
