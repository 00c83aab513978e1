#!/usr/bin/env python3
"""Developer tool (GPU box): would a cost-sorted env -> workgroup assignment shorten the headline launch?  Records the event
counts of every wavefront (the `wtimeline` build) over consecutive steps, prices them with the per-SIMD regression of
profiles/r02_wave_timeline_default.txt, and compares the most loaded SIMD (4 wavefronts each) under (a) the hardware's fixed
assignment, (b) an assignment balanced with the PREVIOUS step's costs (+ the reset flags, which are known before a launch),
(c) the unattainable one balanced with the step's own costs.

    HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so python tools/balance_sim.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd import _abi  # noqa: E402
from highwayenv_amd.engine import Engine  # noqa: E402

E, STEPS = 4096, 80
cfg_d = _abi.highway_fast_default_config()
cfg_d.update({"vehicles_count": 50, "lanes_count": 4})
eng = Engine(_abi.make_config(cfg_d, E, fast=True))
eng.reset(base_seed=5, ego_spacing=1.5, vehicles_density=1.0)
eng.set_autoreset(True, base_seed=99, ego_spacing=1.5, vehicles_density=1.0)
rng = np.random.default_rng(0)
W = np.array([0.51, 0.08, 0.08, 0.09, 0.14])  # sat, near, chain, chain_rival, follower_trips (us per event on the SIMD)
costs, resets = [], []
for t in range(STEPS):
    obs, _, term, trunc, _ = eng.step(rng.integers(0, 5, size=(E, 1)))
    w = np.ascontiguousarray(obs).view(np.uint32).reshape(E, -1)
    was_reset = w[:, 4] == 1
    ev = w[:, 5:10].astype(np.float64)
    ev[was_reset] = 0
    if t >= 20:
        costs.append(ev @ W - 3.75 * was_reset)
        resets.append(was_reset.copy())
C, R = np.array(costs), np.array(resets)
print("per-wave variable cost (us on its SIMD): mean %.2f sd %.2f; lag-1 autocorrelation %.2f, lag-2 %.2f, lag-4 %.2f" % (
    C.mean(), C.std(), np.corrcoef(C[:-1].ravel(), C[1:].ravel())[0, 1], np.corrcoef(C[:-2].ravel(), C[2:].ravel())[0, 1],
    np.corrcoef(C[:-4].ravel(), C[4:].ravel())[0, 1]))


def worst(cost, order):  # envs dealt to 1024 SIMDs, 4 each, in the given order (order[k] -> SIMD k % 1024 snake-wise)
    simd = np.empty(E, int)
    ranks = np.arange(E)
    lap, pos = ranks // 1024, ranks % 1024
    simd[order] = np.where(lap % 2 == 0, pos, 1023 - pos)
    return np.bincount(simd, weights=cost, minlength=1024).max()


fixed = np.arange(E)
res = {"fixed": [], "prev": [], "oracle": [], "resets_only": []}
for t in range(1, len(C)):
    known = -3.75 * R[t]  # (the done flags of step t - 1 say who resets in step t)
    res["fixed"].append(worst(C[t], fixed))
    res["resets_only"].append(worst(C[t], np.argsort(-known, kind="stable")))
    res["prev"].append(worst(C[t], np.argsort(-(np.where(R[t - 1], C[t - 1].mean(), C[t - 1]) * 0.5 + known), kind="stable")))
    res["oracle"].append(worst(C[t], np.argsort(-C[t], kind="stable")))
base = 35.26
for k, v in res.items():
    print(f"{k:12s}: most loaded SIMD {base + np.mean(v):.1f} us (variable part {np.mean(v):+.2f})")
print(f"mean SIMD: {base + 4 * C.mean():.1f} us")
