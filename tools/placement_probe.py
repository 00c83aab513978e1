#!/usr/bin/env python3
"""Developer tool (GPU box): what does a cost-balanced placement of the environments on the SIMDs buy the headline launch?

Engines with the same seeds and actions walk through the same trajectory; engine A keeps the hardware's order (workgroup b steps
environment b) and, being the `wtimeline` build, reports for every wavefront its SIMD, its clock stamps and its event counts.
Before each step the other engines get a workgroup order (hwy_set_block_order) computed from
  * `oracle`    -- the priced event counts A reports for THIS very step (perfect foresight: the ceiling of any predictor),
  * `predicted` -- a state predictor (vehicles near the ego, lane changers, the reset flags: tools/placement_study.py),
  * `shuffled`  -- a random permutation (control: placement noise alone),
dealt to the SIMDs by longest-processing-time-first on the workgroup -> SIMD map of A's previous launch.  Reported: the launch
span (first wavefront start to last wavefront end, s_memrealtime) of each engine over the same steps.

    python tools/ablate/make_variants.py wtimeline
    HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so python tools/placement_probe.py [envs] [steps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W_US = np.array([0.53, 0.08, 0.07, 0.02, 0.13])
RESET_US = -3.8
cfg_d = _abi.highway_fast_default_config()
cfg_d.update({"vehicles_count": 50, "lanes_count": 4})
cfg = _abi.make_config(cfg_d, E, fast=True)
names = ["hardware", "oracle", "predicted", "shuffled"]
engs = {k: Engine(cfg) for k in names}
for eng in engs.values():
    eng.reset(base_seed=5, ego_spacing=1.5, vehicles_density=1.0)
    eng.set_autoreset(True, base_seed=99, ego_spacing=1.5, vehicles_density=1.0)
rng = np.random.default_rng(0)


def stamps(obs):
    w = np.ascontiguousarray(obs).view(np.uint32).reshape(E, -1)
    t0, t1 = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64)
    base = t0.min()
    hw, xcc = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64)
    sid = ((((xcc & 0xf) * 8 + ((hw >> 13) & 7)) * 16 + ((hw >> 8) & 0xf)) * 4 + ((hw >> 4) & 3))
    reset = w[:, 4] == 1
    ev = w[:, 5:10].astype(np.float64)
    ev[reset] = 0
    return ((t0 - base) & 0xffffffff) / 100.0, ((t1 - base) & 0xffffffff) / 100.0, sid, reset, ev


def lpt(cost, simd_of_block):
    """env_of_block: environments in descending cost, each to the least loaded SIMD that still has a free workgroup."""
    import heapq
    units = {}
    for b, s in enumerate(simd_of_block):
        units.setdefault(int(s), []).append(b)
    heap = [(0.0, s) for s in units]
    heapq.heapify(heap)
    env_of_block = np.full(E, -1, np.int64)
    lo = cost.min()
    for e in np.argsort(-cost, kind="stable"):
        load, s = heapq.heappop(heap)
        env_of_block[units[s].pop()] = e
        if units[s]:
            heapq.heappush(heap, (load + (cost[e] - lo), s))
    return env_of_block


def predictor(st, done):
    ctrl = (st["flags"] & _abi.F_CONTROLLED) != 0
    ego = ctrl.argmax(-1)
    r = np.arange(E)
    dx, dy = st["x"] - st["x"][r, ego][:, None], st["y"] - st["y"][r, ego][:, None]
    other = ~ctrl
    near10 = (other & (np.abs(dx) < 10) & (np.abs(dy) < 3)).sum(-1)
    near6 = (other & (np.abs(dx) < 6.5) & (np.abs(dy) < 5)).sum(-1)
    changing = (other & (st["lane"] != st["target_lane"])).sum(-1)
    return 0.96 * near10 + 0.46 * near6 + 0.15 * changing + RESET_US * done


spans = {k: [] for k in names}
simd_end = {k: [] for k in names}
same = []
sid_prev = None
done = np.zeros(E, bool)
for t in range(STEPS):
    acts = rng.integers(0, 5, size=(E, 1))
    st = engs["hardware"].get_state() if t >= 20 else None
    obs, _, term, trunc, _ = engs["hardware"].step(acts)
    t0, t1, sid, reset, ev = stamps(obs)
    cost = ev @ W_US + RESET_US * reset
    if t >= 20:
        spans["hardware"].append(t1.max())
        simd_end["hardware"].append(np.mean([t1[sid == s].max() for s in np.unique(sid)]))
        same.append(float((sid == sid_prev).mean()))
        orders = {"oracle": lpt(cost, sid_prev), "predicted": lpt(predictor(st, done), sid_prev), "shuffled": rng.permutation(E)}
    for k in names[1:]:
        if t >= 20:
            engs[k].set_block_order(orders[k])
        o2, _, te2, tr2, _ = engs[k].step(acts)
        assert (te2 == term).all() and (tr2 == trunc).all()
        if t >= 20:
            a0, a1, s2, _, _ = stamps(o2)
            spans[k].append(a1.max())
            simd_end[k].append(np.mean([a1[s2 == s].max() for s in np.unique(s2)]))
    sid_prev = sid
    done = term | trunc
print(f"{E} envs, {STEPS - 20} measured steps; workgroups on the same SIMD as in the previous launch: {100 * np.mean(same):.1f} %")
for k in names:
    v = np.array(spans[k])
    print(f"{k:10s} launch span mean {v.mean():.2f} us  median {np.median(v):.2f}  p10 {np.percentile(v, 10):.2f}  p90 {np.percentile(v, 90):.2f};"
          f"  mean SIMD end {np.mean(simd_end[k]):.2f} us")
