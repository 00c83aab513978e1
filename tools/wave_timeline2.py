#!/usr/bin/env python3
"""Developer tool (GPU box): per-wavefront timeline of ONE steady-state launch of hwy_step_wave_kernel (bench workload:
random actions, auto-reset), grouped by hardware unit -- which SIMDs / CUs / XCDs finish late, and why?

    python tools/ablate/make_variants.py wtimeline            # build container
    HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so python tools/wave_timeline2.py [envs] [steps]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
TUNING = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[3:]}
cfg_d = _abi.highway_fast_default_config()
cfg_d.update({"vehicles_count": 50, "lanes_count": 4})
cfg = _abi.make_config(cfg_d, E, fast=True, tuning=TUNING)
eng = Engine(cfg)
eng.reset(base_seed=5, ego_spacing=1.5, vehicles_density=1.0)
eng.set_autoreset(True, base_seed=99, ego_spacing=1.5, vehicles_density=1.0)
rng = np.random.default_rng(0)
done = np.zeros(E, bool)
spans = []
prev_sid = last_sid = None
for t in range(STEPS):
    obs, _, term, trunc, _ = eng.step(rng.integers(0, 5, size=(E, 1)))
    done = term | trunc
    w = np.ascontiguousarray(obs).view(np.uint32).reshape(E, -1)
    was_reset = w[:, 4] == 1  # (the reset path stamps itself too)
    t0, t1 = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64)
    base = t0.min()
    spans.append(float(((t1 - base) & 0xffffffff).max() / 100.0))
    hw_, xcc_ = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64)
    prev_sid, last_sid = last_sid, ((((xcc_ & 0xf) * 8 + ((hw_ >> 13) & 7)) * 16 + ((hw_ >> 8) & 0xf)) * 4 + ((hw_ >> 4) & 3))
hw, xcc = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64)
t0, t1 = ((t0 - base) & 0xffffffff) / 100.0, ((t1 - base) & 0xffffffff) / 100.0
xcd, se, cu, simd = xcc & 0xf, (hw >> 13) & 7, (hw >> 8) & 0xf, (hw >> 4) & 3
life = t1 - t0
print(f"launch spans of the last 10 steps (us): {[round(s, 1) for s in spans[-10:]]}")
print(f"last step: {int(was_reset.sum())} reset waves; lifetime reset mean {life[was_reset].mean():.1f} max {life[was_reset].max():.1f} us; "
      f"stepping waves mean {life[~was_reset].mean():.1f} p90 {np.percentile(life[~was_reset], 90):.1f} max {life[~was_reset].max():.1f} us")
print("end time percentiles (us):", {q: round(float(np.percentile(t1, q)), 1) for q in (0, 10, 50, 90, 99, 100)})
sid = ((xcd * 8 + se) * 16 + cu) * 4 + simd
cuid = (xcd * 8 + se) * 16 + cu
for name, key in (("xcd", xcd), ("se", xcd * 8 + se), ("cu", cuid), ("simd", sid)):
    ends = np.array([t1[key == k].max() for k in np.unique(key)])
    cnt = np.array([(key == k).sum() for k in np.unique(key)])
    print(f"per {name:4s}: {len(ends):4d} units, waves/unit {cnt.min()}..{cnt.max()}, unit end time min {ends.min():.1f} "
          f"mean {ends.mean():.1f} max {ends.max():.1f} us")
# does a SIMD's end time follow the work of its waves?  (work proxy: nothing measured alone here, so use resets)
rs = np.array([was_reset[sid == k].sum() for k in np.unique(sid)])
es = np.array([t1[sid == k].max() for k in np.unique(sid)])
for r in np.unique(rs):
    print(f"SIMDs holding {r} reset waves: {int((rs == r).sum()):4d}, end time mean {es[rs == r].mean():.1f} max {es[rs == r].max():.1f} us")
# within a SIMD: spread of the 4 end times
spread = np.array([np.ptp(t1[sid == k]) for k in np.unique(sid)])
print(f"spread of end times within a SIMD: mean {spread.mean():.1f} p90 {np.percentile(spread, 90):.1f} max {spread.max():.1f} us")
per_cu_spread = np.array([np.ptp([t1[(cuid == c) & (simd == s)].max() for s in range(4) if ((cuid == c) & (simd == s)).any()]) for c in np.unique(cuid)])
print(f"spread of SIMD end times within a CU: mean {per_cu_spread.mean():.1f} max {per_cu_spread.max():.1f} us")
cu_end = {int(c): float(t1[cuid == c].max()) for c in np.unique(cuid)}
slow = sorted(cu_end, key=cu_end.get)[-4:]
print("slowest CUs (xcd, se, cu): end us:", [((c >> 7), (c >> 4) & 7, c & 15, round(cu_end[c], 1)) for c in slow],
      "| median CU end", round(float(np.median(list(cu_end.values()))), 1))
slot = hw & 0xf
print("by hardware wave slot: count, mean start, mean end, mean lifetime (us):",
      {int(k): (int((slot == k).sum()), round(float(t0[slot == k].mean()), 2), round(float(t1[(slot == k) & ~was_reset].mean()), 1),
                round(float(life[(slot == k) & ~was_reset].mean()), 1)) for k in np.unique(slot)})
# is the block -> SIMD placement the same in every launch?  (previous step's placement kept in `prev_sid`)
if prev_sid is not None:
    print(f"blocks on the same SIMD as in the previous launch: {100.0 * float((prev_sid == sid).mean()):.1f} %")
# event counters of the stepping waves (sat trips, near pairs, chain links, chain links with a rival, follower-test trips)
ev = w[:, 5:10].astype(np.int64)
ev[was_reset] = 0
names_ev = ["sat", "near", "chain", "chain_rival", "follower_trips"]
usid = np.unique(sid)
ev_simd = np.array([ev[sid == k].sum(0) for k in usid])
print("events per wave (mean):", {n: round(float(ev[~was_reset, k].mean()), 2) for k, n in enumerate(names_ev)})
top = es >= np.percentile(es, 99)
print("events per SIMD, all SIMDs (mean):   ", {n: round(float(ev_simd[:, k].mean()), 2) for k, n in enumerate(names_ev)})
print("events per SIMD, slowest 1 % (mean): ", {n: round(float(ev_simd[top, k].mean()), 2) for k, n in enumerate(names_ev)})
A_ = np.column_stack([ev_simd, rs, np.ones(len(usid))])
coef, *_ = np.linalg.lstsq(A_, es, rcond=None)
pred = A_ @ coef
print("least squares: SIMD end time ~", {n: round(float(c), 2) for n, c in zip(names_ev + ["reset_waves", "const"], coef)},
      f"R^2 = {1 - ((es - pred) ** 2).sum() / ((es - es.mean()) ** 2).sum():.2f}")
late = np.argsort(-es)[:8]
u = np.unique(sid)
for k in late:
    sel = sid == u[k]
    print(f"late SIMD xcd{int(xcd[sel][0])} se{int(se[sel][0])} cu{int(cu[sel][0])} simd{int(simd[sel][0])}: ends {np.round(np.sort(t1[sel]), 1).tolist()} "
          f"starts {np.round(np.sort(t0[sel]), 2).tolist()} resets {int(was_reset[sel].sum())}")
print(json.dumps({"envs": E, "span_us_last10": spans[-10:], "end_pct": {str(q): float(np.percentile(t1, q)) for q in (0, 10, 50, 90, 99, 100)}}))
