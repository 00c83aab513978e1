#!/usr/bin/env python3
"""GPU micro-benchmark: kernel time of hwy_step_frames(n) for several n (per-frame cost vs the fixed
load/observe/store cost) and of the full policy step, via the engine's HIP-event profile API."""
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
out = {}
for n in (0, 1, 2, 5, 10, 20):
    eng.reset(base_seed=5, ego_spacing=1.5, vehicles_density=1.0)
    eng.step_frames(None, n)  # warm-up
    eng.reset(base_seed=5, ego_spacing=1.5, vehicles_density=1.0)
    eng.profile_enable(1)
    reps = 20
    for _ in range(reps):
        eng.step_frames(None, n)
    ms, k = eng.profile_read()
    eng.profile_enable(0)
    out[f"frames_{n}_us"] = ms / k * 1e3
acts = np.ones((E, 1), np.int32)
eng.reset(base_seed=5, ego_spacing=1.5, vehicles_density=1.0)
eng.step(acts)
eng.profile_enable(1)
for _ in range(20):
    eng.step(acts)
ms, k = eng.profile_read()
out["full_step_us"] = ms / k * 1e3
print(json.dumps(out))
