#!/usr/bin/env python3
"""HBM-traffic probe for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (run each counter in its own pass).

Launch sequence (all with the step kernel, E=4096 x N=51, steady-state bench workload):
  * 20 x hwy_step_frames(n_frames=0): pure load + store of the state SoA -- a KNOWN byte count in the
    kernel's own access pattern (8 B/lane f64 rows, 448-byte pitch), used to calibrate the counters
    (MI355X_MICROARCH.md: gfx950 FETCH_SIZE under-reports wide coalesced reads, other widths uncalibrated);
  * 20 x full policy steps (the bench's kernel).
Prints the known byte counts so that tools/traffic_report.py can turn counter values into bytes.
"""
import json
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
N = cfg.num_vehicles
pitch = (N + 7) & ~7
eng = Engine(cfg)
eng.reset(base_seed=5, ego_spacing=1.5, vehicles_density=1.0)
eng.set_autoreset(True, base_seed=99, ego_spacing=1.5, vehicles_density=1.0)
rng = np.random.default_rng(0)
for t in range(40):  # reach the steady-state mix of episode ages
    eng.step(rng.integers(0, 5, size=(E, 1)))
for _ in range(20):
    eng.step_frames(None, 0)
for t in range(20):
    eng.step(rng.integers(0, 5, size=(E, 1)))
# bytes a perfect implementation moves for a frames=0 launch, at 64-byte line granularity
row_f64 = ((N * 8 + 63) // 64) * 64
row_i32 = ((N * 4 + 63) // 64) * 64
# a frames=0 launch reads x,y,heading,speed,timer,target_speed,delta (+packed word; the impact pair only
# where flagged) and writes back x,y,heading,speed,timer (+packed word): see load_vehicle / store_vehicle
known = {"E": E, "N": N, "pitch": pitch,
         "frames0_read_bytes": E * (7 * row_f64 + row_i32),
         "frames0_write_bytes": E * (5 * row_f64 + row_i32),
         "frames0_read_bytes_exact": E * N * (7 * 8 + 4),
         "algorithmic_bytes_per_step": (72 * N + 110) * E}
print(json.dumps(known))
