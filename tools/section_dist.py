#!/usr/bin/env python3
"""Developer tool (GPU): DISTRIBUTION over environments of the per-section s_memtime cycles of the one-wavefront step
kernel (the `wticks` build of tools/ablate/make_variants.py; HWY_ENGINE_LIB must point to it), on the bench workload.

A launch at 4096 envs lasts as long as its most loaded SIMD (4 resident wavefronts each), so what matters is not the
mean cost of a section but how unevenly it is spread: prints mean / p50 / p90 / p99 / max per section, each section's
share of the variance of the wave total, and the expected maximum over 1024 SIMDs of the sum of 4 random waves.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg_d = _abi.highway_fast_default_config()
cfg_d.update({"vehicles_count": 50, "lanes_count": 4})
cfg = _abi.make_config(cfg_d, E, fast=True)
eng = Engine(cfg)
eng.reset(base_seed=5, ego_spacing=1.5, vehicles_density=1.0)
eng.set_autoreset(True, base_seed=99, ego_spacing=1.5, vehicles_density=1.0)  # the bench's steady-state workload
rng = np.random.default_rng(0)
rows = []
for t in range(60):
    obs, _, term, trunc, _ = eng.step(rng.integers(0, 5, size=(E, 1)))
    if t >= 40:
        rows.append(obs.reshape(E, -1)[:, :13].astype(np.float64))
names = ["load+chk", "A meta-action", "C rank check/recount", "C membership+snapshot", "D neighbour ranks+gather",
         "D free road + gaps", "D mobil", "D abort chain", "E control", "F integrate", "G collisions", "H observe"]
X = np.concatenate(rows)          # [(steps*E), 13]
# a resetting environment takes the spawn path: its tick slots hold observation floats, not cycles -- drop those rows
sane = (X[:, :12] >= 0).all(1) & (X[:, :12].sum(1) > 1000)
X = X[sane]
S = X[:, :12]
tot = S.sum(1)
print(f"{len(X)} wave samples ({(~sane).sum()} reset waves dropped); cycles per step per wave (s_memtime, 100 MHz x ... as reported)")
print(f"{'section':26s} {'mean':>8s} {'p50':>8s} {'p90':>8s} {'p99':>8s} {'max':>8s} {'share':>6s} {'var share':>9s}")
cov = np.cov(np.column_stack([S, tot]).T)
for k, nm in enumerate(names):
    c = S[:, k]
    print(f"{nm:26s} {c.mean():8.0f} {np.percentile(c, 50):8.0f} {np.percentile(c, 90):8.0f} {np.percentile(c, 99):8.0f} "
          f"{c.max():8.0f} {100 * c.mean() / tot.mean():5.1f}% {100 * cov[k, 12] / cov[12, 12]:8.1f}%")
print(f"{'total':26s} {tot.mean():8.0f} {np.percentile(tot, 50):8.0f} {np.percentile(tot, 90):8.0f} {np.percentile(tot, 99):8.0f} {tot.max():8.0f}")
g = np.random.default_rng(1)
sums = np.array([tot[g.integers(0, len(tot), size=(1024, 4))].sum(1).max() for _ in range(50)])
print(f"sum of 4 random waves: mean {4 * tot.mean():.0f}; expected max over 1024 SIMDs {sums.mean():.0f} "
      f"(x{sums.mean() / (4 * tot.mean()):.2f} of the mean)")
print("frames whose rank order changed, per step (of 5 frames):", X[:, 12].mean())
print(json.dumps({"mean": {n: float(S[:, k].mean()) for k, n in enumerate(names)}, "total_mean": float(tot.mean()),
                  "total_p99": float(np.percentile(tot, 99)), "simd_max_over_mean": float(sums.mean() / (4 * tot.mean()))}))
