#!/usr/bin/env python3
"""Developer tool (build container, no GPU): can the cost of an environment's NEXT policy step be predicted from what is known
before the launch, well enough for a cost-balanced environment -> SIMD placement of the headline launch to pay?

The variable part of a wavefront's cost is made of EVENTS (SAT trips, near pairs, abort-chain links, follower-safety trips:
the per-SIMD regression of profiles/r03_wave_timeline_default.txt prices them) and events are a function of the simulation
state and the actions, not of the GPU -- so they are counted here on the CPU emulator of the kernel source (tests/emu) with
the `wtimeline` variant's counters, next to the full state at the start of every step.  Then: predictors built from the state
(what the previous step's wavefront could write out for free), their R^2 against the priced cost, and the most loaded SIMD
(4 wavefronts each) under the hardware's fixed placement, a placement balanced with each predictor, and perfect foresight.

    python tools/placement_study.py collect [procs] [envs_per_proc] [steps]     # -> /tmp/hwy_placement/*.npz  (minutes)
    python tools/placement_study.py analyse
"""
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = "/tmp/hwy_placement"
W_US = np.array([0.53, 0.08, 0.07, 0.02, 0.13])  # us on the SIMD per: SAT trip, near pair, chain link, ... with a rival (+), follower trip
RESET_US = -3.8


def build_variant():
    src = os.path.join(OUT, "src")
    shutil.rmtree(src, ignore_errors=True)
    for d in ("tests/emu", "highwayenv_amd/csrc", "include"):
        os.makedirs(os.path.join(src, d))
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".h", ".cpp")):
                shutil.copy(os.path.join(ROOT, d, f), os.path.join(src, d, f))
    p = os.path.join(src, "highwayenv_amd/csrc/hwy_wave.h")
    t = open(p).read()

    def sub(old, new):
        nonlocal t
        assert old in t, old
        t = t.replace(old, new)
    sub("struct WaveShared {", "struct WaveShared {\n  int cnt[8];")
    sub("  WaveTurn turn;\n  wave_turn_init(turn, p.prio_shift);", "  if (i < 8) sh.cnt[i] = 0;\n  WaveTurn turn;\n  wave_turn_init(turn, p.prio_shift);")
    sub("              r = pair_collide(A, Bb, p.dt, &tx, &ty);", "              atomicAdd(&sh.cnt[0], 1); r = pair_collide(A, Bb, p.dt, &tx, &ty);")
    sub("            if (!surely_apart(A, Bb, p.dt)) {", "            atomicAdd(&sh.cnt[1], 1);\n            if (!surely_apart(A, Bb, p.dt)) {")
    sub("        if (__ballot(rival) == 0) continue;", "        if (i == 0) atomicAdd(&sh.cnt[2], 1);\n        if (__ballot(rival) == 0) continue;\n        if (i == 0) atomicAdd(&sh.cnt[3], 1);")
    sub("        const bool pend = pend_l || pend_r;", "        if (i == 0) atomicAdd(&sh.cnt[4], 1);\n        const bool pend = pend_l || pend_r;")
    sub("  store_vehicle<1>(q, e, me, false);\n}",
        "  store_vehicle<1>(q, e, me, false);\n  HWY_WAVE_LDS_FENCE();\n"
        "  if (q.obs && i == 0) { unsigned *o = (unsigned *)(q.obs + (size_t)e * q.A * q.V * q.F); o[4] = 0u; for (int k = 0; k < 5; ++k) o[5 + k] = (unsigned)sh.cnt[k]; }\n}")
    sub("      p.terminated[eo] = 0;\n      p.truncated[eo] = 0;\n    }\n    return;",
        "      p.terminated[eo] = 0;\n      p.truncated[eo] = 0;\n    }\n"
        "    if (p.obs && i == 0) { unsigned *o = (unsigned *)(p.obs + (size_t)e * p.A * p.V * p.F); o[4] = 1u; }\n    return;")
    open(p, "w").write(t)
    lib = os.path.join(OUT, "libhwy_emu_events.so")
    subprocess.run(["g++", "-std=c++20", "-O2", "-pthread", "-fPIC", "-shared", "-ffp-contract=off", "-o", lib,
                    os.path.join(src, "tests/emu/emu_engine.cpp")], check=True)
    return lib


def worker(k, E, steps):
    from highwayenv_amd import _abi
    from tests.emu import emu
    emu._LIB = os.path.join(OUT, "libhwy_emu_events.so")
    emu.build = lambda force=False: emu._LIB
    cfg_d = _abi.highway_fast_default_config()
    cfg_d.update({"vehicles_count": 50, "lanes_count": 4})
    eng = emu.EmuEngine(_abi.make_config(cfg_d, E, fast=True))
    eng.reset(base_seed=5 + 1000 * k, ego_spacing=1.5, vehicles_density=1.0)
    eng.set_autoreset(True, base_seed=99 + 1000 * k, ego_spacing=1.5, vehicles_density=1.0)
    rng = np.random.default_rng(k)
    keep = ("x", "y", "speed", "lane", "target_lane", "timer", "flags", "heading")
    rec = {f: [] for f in keep}
    rec.update({"events": [], "reset": [], "action": []})
    for t in range(steps):
        acts = rng.integers(0, 5, size=(E, 1))
        st = eng.get_state()
        obs, _, term, trunc, _ = eng.step(acts)
        w = np.ascontiguousarray(obs).view(np.uint32).reshape(E, -1)
        if t >= 15:  # (past the spawn transient: the bench's warm-up)
            for f in keep:
                rec[f].append(st[f].astype(np.float32 if st[f].dtype == np.float64 else st[f].dtype))
            rec["events"].append(w[:, 5:10].astype(np.int32).copy())
            rec["reset"].append(w[:, 4] == 1)
            rec["action"].append(acts[:, 0].astype(np.int8))
    np.savez_compressed(os.path.join(OUT, f"part{k}.npz"), **{f: np.array(v) for f, v in rec.items()})


def load():
    parts = [np.load(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT)) if f.startswith("part")]
    return {f: np.concatenate([p[f] for p in parts], axis=1) for f in parts[0].files}  # [T, E, ...]


def features(d):
    """What a wavefront could write out at the end of step t - 1 (== the state at the start of step t) + the reset flag + the
    action of step t (known before the launch)."""
    from highwayenv_amd import _abi
    x, y, lane, tgt, flags = d["x"], d["y"], d["lane"], d["target_lane"], d["flags"]
    ctrl = (flags & _abi.F_CONTROLLED) != 0
    ego = ctrl.argmax(-1)
    T, E, N = x.shape
    tt, ee = np.meshgrid(np.arange(T), np.arange(E), indexing="ij")
    ex, ey = x[tt, ee, ego], y[tt, ee, ego]
    dx, dy = x - ex[..., None], y - ey[..., None]
    other = ~ctrl
    changing = other & (lane != tgt)
    F = {
        "reset": d["reset"].astype(float),
        "n_changing": changing.sum(-1),
        "n_changing_pairs": np.maximum(changing.sum(-1) - 1, 0),
        "near10": (other & (np.abs(dx) < 10) & (np.abs(dy) < 3)).sum(-1),
        "near7": (other & (np.abs(dx) < 7) & (np.abs(dy) < 2.5)).sum(-1),
        "near6_lat5": (other & (np.abs(dx) < 6.5) & (np.abs(dy) < 5)).sum(-1),
        "ego_lane_change": np.isin(d["action"], (0, 2)).astype(float),
    }
    return F


def worst(cost, order, n_simd, per):  # envs dealt to the SIMDs snake-wise in the given order
    E = len(cost)
    simd = np.empty(E, int)
    r = np.arange(E)
    lap, pos = r // n_simd, r % n_simd
    simd[order] = np.where(lap % 2 == 0, pos, n_simd - 1 - pos)
    return np.bincount(simd, weights=cost, minlength=n_simd).max()


def analyse():
    d = load()
    ev = d["events"].astype(float)
    R = d["reset"]
    ev[R] = 0
    C = ev @ W_US + RESET_US * R
    T, E = C.shape
    print(f"{T} steps x {E} envs; events per env-step (SAT, near, links, links+rival, follower trips): {ev[~R].mean(0).round(3)}; resets {R.mean():.3f}")
    print(f"variable cost per wavefront: mean {C.mean():.2f} us, sd {C.std():.2f}; lag-1 autocorrelation {np.corrcoef(C[:-1].ravel(), C[1:].ravel())[0, 1]:.2f}")
    F = features(d)
    names = list(F)
    X = np.stack([F[k].astype(float) for k in names], -1)
    live = ~R
    print("correlation of each feature with the priced cost (live env-steps) / with each event count:")
    for j, k in enumerate(names):
        cc = [np.corrcoef(X[live][:, j], ev[live][:, q])[0, 1] if X[live][:, j].std() > 0 else 0 for q in range(5)]
        print(f"  {k:18s} cost {np.corrcoef(X[live][:, j], C[live])[0, 1]:+.2f}   events " + " ".join(f"{c:+.2f}" for c in cc))
    # least squares on the first half, evaluated on the second
    half = T // 2
    A = np.concatenate([X, np.ones((T, E, 1))], -1)
    coef, *_ = np.linalg.lstsq(A[:half].reshape(-1, A.shape[-1]), C[:half].ravel(), rcond=None)
    pred = A @ coef
    r2 = 1 - ((C[half:] - pred[half:]) ** 2).sum() / ((C[half:] - C[half:].mean()) ** 2).sum()
    print("least-squares predictor:", dict(zip(names + ["1"], coef.round(3))), f"R^2 (held-out half) {r2:.3f}")
    n_simd, per = E // 4, 4
    res = {k: [] for k in ("fixed", "resets_only", "prev_cost", "state_predictor", "state_predictor_q16", "oracle")}
    for t in range(max(half, 1), T):
        c = C[t]
        res["fixed"].append(worst(c, np.arange(E), n_simd, per))
        res["resets_only"].append(worst(c, np.argsort(-(RESET_US * R[t]), kind="stable"), n_simd, per))
        prev = np.where(R[t - 1], C[t - 1][~R[t - 1]].mean(), C[t - 1]) * 0.5 + RESET_US * R[t]
        res["prev_cost"].append(worst(c, np.argsort(-prev, kind="stable"), n_simd, per))
        res["state_predictor"].append(worst(c, np.argsort(-pred[t], kind="stable"), n_simd, per))
        q = np.clip(np.floor((pred[t] - RESET_US) / 0.5), 0, 15)  # 16 cost classes of 0.5 us (what a counting sort in the kernel would use)
        res["state_predictor_q16"].append(worst(c, np.argsort(-q, kind="stable"), n_simd, per))
        res["oracle"].append(worst(c, np.argsort(-c, kind="stable"), n_simd, per))
    print(f"most loaded SIMD, variable part (us; mean SIMD = {4 * C.mean():+.2f}):")
    for k, v in res.items():
        print(f"  {k:22s} {np.mean(v):+.2f}")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1] == "collect":
        procs, E, steps = (int(a) for a in (sys.argv[2:5] + ["8", "512", "55"][len(sys.argv) - 2:]))
        build_variant()
        import multiprocessing as mp
        with mp.Pool(procs) as pool:
            pool.starmap(worker, [(k, E, steps) for k in range(procs)])
    else:
        analyse()
