#!/usr/bin/env python3
"""Developer tool (GPU box): when and where do the wavefronts of ONE launch of hwy_step_wave_kernel run?

    python tools/ablate/make_variants.py wtimeline            # build container
    HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so python tools/wave_timeline.py [envs]

The `wtimeline` variant makes every wavefront write its start / end s_memrealtime (100 MHz) and its HW_ID / XCC_ID
over the first observation words of its environment.  Prints the launch span, the spread of the start times (dispatch
ramp), the wavefront lifetimes, how many wavefronts every SIMD held at once, and the blockIdx -> (xcc, se, cu, simd)
mapping of the first workgroups.
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
acts = np.ones((E, 1), np.int32)
for _ in range(5):
    obs = eng.step(acts)[0]
w = np.ascontiguousarray(obs).view(np.uint32).reshape(E, -1)
t0, t1, hw, xcc = (w[:, k].astype(np.int64) for k in range(4))
base = t0.min()
t0, t1 = (t0 - base) & 0xffffffff, (t1 - base) & 0xffffffff
us = lambda ticks: np.asarray(ticks, float) / 100.0  # noqa: E731  (100 MHz)
simd = ((xcc & 0xf) << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 11) | (((hw >> 8) & 0xf) << 4) | ((hw >> 4) & 3)
order = np.argsort(t0, kind="stable")
conc = {}
for s_id in np.unique(simd):
    sel = simd == s_id
    ev = sorted([(a, 1) for a in t0[sel]] + [(b, -1) for b in t1[sel]])
    cur = best = 0
    for _, d in ev:
        cur += d
        best = max(best, cur)
    conc[int(s_id)] = (int(sel.sum()), best)
out = {
    "envs": E,
    "launch_span_us": float(us(t1.max())),
    "start_offset_us": {q: float(us(np.percentile(t0, q))) for q in (0, 10, 50, 90, 99, 100)},
    "end_us": {q: float(us(np.percentile(t1, q))) for q in (0, 10, 50, 90, 100)},
    "lifetime_us": {"min": float(us((t1 - t0).min())), "mean": float(us((t1 - t0).mean())), "max": float(us((t1 - t0).max()))},
    "simds_used": len(conc),
    "waves_per_simd": {"min": min(v[0] for v in conc.values()), "max": max(v[0] for v in conc.values())},
    "max_concurrent_per_simd": {"min": min(v[1] for v in conc.values()), "max": max(v[1] for v in conc.values())},
    "first_blocks": [{"block": int(b), "xcc": int(xcc[b] & 0xf), "se": int((hw[b] >> 13) & 7), "cu": int((hw[b] >> 8) & 0xf),
                      "simd": int((hw[b] >> 4) & 3), "wave": int(hw[b] & 0xf), "start_us": float(us(t0[b]))} for b in range(0, 40)],
    "start_us_by_block_decile": [float(us(t0[int(k * (E - 1) / 10)])) for k in range(11)],
}
print(json.dumps(out))
