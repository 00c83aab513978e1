#!/usr/bin/env python3
"""Developer tool (GPU box): what does pairing heavy with light environments on the SIMDs buy the intersection launch?

BASELINE config 4 runs 2048 step wavefronts on 1024 SIMDs: two per SIMD, all resident at once, so the launch lasts as long as its
slowest SIMD -- and an environment's cost is mostly its vehicle count, which is known BEFORE the launch.  Engines with the same seeds and
actions walk through the same trajectory (`ixticks` build: every wavefront reports its section clocks, its SIMD and its workgroup):
  * `hardware` -- workgroup b steps environment b;
  * `paired`   -- before every step: environments in descending vehicle count, longest-processing-time-first on the workgroup -> SIMD
                  map of the previous launch (hwy_set_block_order);
  * `shuffled` -- a random permutation (control).
Reported: the step kernel's duration (HIP events) per engine, the correlation of a wavefront's clock total with the vehicle count, and
how stable the workgroup -> SIMD map is.

    python tools/ablate/make_variants.py ixticks
    HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_ixticks.so python tools/ix_placement_probe.py [envs] [steps]
"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 80
cfg_d, fast, scenario = bench.workload_config("intersection")
cfg = _abi.make_config(cfg_d, E, fast=fast, scenario=scenario)
names = ["hardware", "paired", "oracle", "shuffled", "sorted", "mid"]
engs = {k: Engine(cfg) for k in names}
for eng in engs.values():
    eng.reset(base_seed=5)
    eng.set_autoreset(True, base_seed=99)
    eng.profile_enable(1)
rng = np.random.default_rng(0)


def lpt(cost, simd_of_block):
    units = {}
    for b, s in enumerate(simd_of_block):
        units.setdefault(int(s), []).append(b)
    heap = [(0.0, s) for s in units]
    heapq.heapify(heap)
    env_of_block = np.full(E, -1, np.int64)
    for e in np.argsort(-cost, kind="stable"):
        load, s = heapq.heappop(heap)
        env_of_block[units[s].pop()] = e
        if units[s]:
            heapq.heappush(heap, (load + cost[e], s))
    return env_of_block


simd_of_block = {k: None for k in names}
dur = {k: [] for k in names}
corr, stable, pair_hist = [], [], []
last = {k: (0.0, 0) for k in names}
for t in range(STEPS):
    acts = rng.integers(0, 3, size=(E, 1))
    st = engs["paired"].get_state()
    count = ((st["flags"] & _abi.F_ABSENT) == 0).sum(-1).astype(np.float64)
    for k in names:
        eng = engs[k]
        if t >= 20:
            if k in ("paired", "oracle"):
                # the dispatcher's pairs are (b, b + E / 2) (90 %; WHICH SIMD a pair lands on changes from launch to launch): the
                # k-th heaviest environment to workgroup k, the k-th lightest to workgroup k + E / 2
                c = count if k == "paired" else true_cost
                o = np.argsort(-c, kind="stable")
                eng.set_block_order(np.concatenate([o[:E // 2], o[E // 2:][::-1]]))
            elif k == "mid":  # heavy first, and every pair's sum about equal: k-th heaviest with the k-th heaviest of the lower half
                o = np.argsort(-count, kind="stable")
                eng.set_block_order(o)
            elif k == "shuffled":
                eng.set_block_order(rng.permutation(E))
            elif k == "sorted":  # descending count in workgroup order (no knowledge of the SIMD map)
                eng.set_block_order(np.argsort(-count, kind="stable"))
        obs, _, term, trunc, _ = eng.step(acts)
        w = obs.reshape(E, -1)[:, :18].astype(np.float64)  # by ENVIRONMENT
        blk, sid = w[:, 17].astype(np.int64), w[:, 16].astype(np.int64)
        m = np.empty(E, np.int64)
        m[blk] = sid
        if simd_of_block[k] is not None and k == "hardware":
            stable.append((m == simd_of_block[k]).mean())
        simd_of_block[k] = m
        ms, n = eng.profile_read()  # (cumulative)
        d_ms, d_n = ms - last[k][0], n - last[k][1]
        last[k] = (ms, n)
        if k == 'hardware':
            true_cost = w[:, :14].sum(1)
        if t >= 30:
            dur[k].append(d_ms * 1e3 / max(d_n, 1))
            if k == "hardware":
                tot = w[:, :14].sum(1)
                true_cost = tot
                step_role = w[:, 14] == 0
                corr.append(np.corrcoef(tot[step_role], count[step_role])[0, 1])
                pair_hist.append(np.bincount(np.bincount(sid, minlength=8192), minlength=6)[:6])
for k in names:
    d = np.array(dur[k])
    print(f"{k:10s} step kernel {d.mean():8.2f} us (median {np.median(d):.2f}, {len(d)} launches)")
print(f"corr(wavefront clock total, vehicle count) = {np.mean(corr):.3f}; workgroup -> SIMD map equal to the previous launch's: "
      f"{100 * np.mean(stable):.1f} %")
print("SIMDs by number of step wavefronts [0, 1, 2, 3, 4, 5]:", np.mean(pair_hist, 0).round(1))
h = simd_of_block["hardware"]
print("SIMD of workgroups 0..15:", h[:16], " 1024..1031:", h[1024:1032])
same = [(h[b] == h[b + d]).mean() for d in (1, 8, 512, 1024) for b in [np.arange(E - d)]]
print("fraction of workgroup pairs (b, b + d) on one SIMD, d = 1, 8, 512, 1024:", np.round(same, 3))
