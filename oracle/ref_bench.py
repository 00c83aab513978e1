"""TEST / MEASUREMENT INFRASTRUCTURE: time the UNMODIFIED reference (Farama-Foundation/HighwayEnv, pure Python) on the
host cores, for ``bench.py``'s ``cpu_baseline`` leg (``"kind": "reference"``).  Never imported by the product.

The reference is imported from ``HWY_REFERENCE_ROOT`` (default ``/root/reference``) through ``oracle/ref_stub.py``
(gymnasium / pygame are not installed in this image).  It exists in the build container only: on the GPU box
``available()`` is False and ``bench.py`` falls back to the C port, quoting the committed build-container figure
(``profiles/reference_cpu_baseline.json``) next to it, labelled "not the same box".

Method = the reference's own benchmark loop (``scripts/regression_test/bench_render_fps.py:55-83``: ``env.reset(seed)``,
then ``env.step(action_space.sample())`` with ``env.reset`` on terminated / truncated, wall clock around the loop)
around ``AbstractEnv.step`` (``highway_env/envs/common/abstract.py:259-285``), with the config dict ``bench.py`` runs on
the GPU, (i) in this process on one core and (ii) in ``multiprocessing`` workers, one env per worker, on every host
core.  Resets are timed separately so that both rates (with / without resets) can be quoted.
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import ref_stub


def available() -> bool:
    return ref_stub.reference_available()


def _make_env(workload: str):
    """The reference environment of a bench.py workload (same config overrides as bench.py)."""
    ref_stub.install()
    if workload in ("fast", "v0", "v0_n100"):
        from highway_env.envs.highway_env import HighwayEnv, HighwayEnvFast
        if workload == "fast":  # BASELINE config 2
            return HighwayEnvFast({"vehicles_count": 50, "lanes_count": 4}), 5
        return HighwayEnv({"vehicles_count": 100} if workload == "v0_n100" else {}), 5
    if workload in ("merge", "merge_ma4"):
        from highway_env.envs.merge_env import MergeEnv, MergeGenericEnv
        if workload == "merge":
            return MergeEnv(), 5
        return MergeGenericEnv({"lanes_count": 4, "vehicles_count": 40, "controlled_vehicles": 4,
                                "action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
                                "observation": {"type": "MultiAgentObservation",
                                                "observation_config": {"type": "Kinematics"}}}), 5
    if workload in ("intersection", "intersection_kin"):
        from highway_env.envs.intersection_env import IntersectionEnv
        return IntersectionEnv({"observation": {"type": "OccupancyGrid"}} if workload == "intersection" else {}), 3
    raise ValueError(workload)


def run(workload: str, budget_s: float, seed: int = 0) -> dict:
    """One env, one core, ``budget_s`` seconds of stepping (resets included in the budget, timed separately)."""
    env, n_actions = _make_env(workload)
    agents = int(env.config.get("controlled_vehicles", 1))
    multi = env.config["action"]["type"] == "MultiAgentAction"
    rng = np.random.default_rng(1234 + seed)
    t_reset = t_step = 0.0
    steps = resets = 0
    next_seed = seed
    t0 = time.perf_counter()
    env.reset(seed=next_seed)
    t_reset += time.perf_counter() - t0
    resets += 1
    n_vehicles = len(env.road.vehicles)
    while t_reset + t_step < budget_s:
        a = tuple(int(x) for x in rng.integers(0, n_actions, size=agents)) if multi else int(rng.integers(0, n_actions))
        t0 = time.perf_counter()
        _obs, _reward, terminated, truncated, _info = env.step(a)
        t_step += time.perf_counter() - t0
        steps += 1
        if terminated or truncated:
            next_seed += 1000
            t0 = time.perf_counter()
            env.reset(seed=next_seed)
            t_reset += time.perf_counter() - t0
            resets += 1
    return {"steps": steps, "resets": resets, "t_step": t_step, "t_reset": t_reset, "vehicles": n_vehicles}


def _worker(args):
    workload, budget_s, seed = args
    return run(workload, budget_s, seed)


def measure(workload: str, budget_s: float = 12.0, all_cores_budget_s: float = 10.0, max_procs: int | None = None) -> dict:
    """The ``cpu_baseline`` object of bench.py: single core first, then one env per core on all host cores."""
    one = run(workload, budget_s, seed=7)
    rate = one["steps"] / (one["t_step"] + one["t_reset"])
    out = {
        "value": rate, "unit": "env-steps/s", "cores": 1, "kind": "reference",
        "sample": (f"unmodified reference ({ref_stub.REFERENCE_ROOT}, pure Python / numpy), workload '{workload}', 1 env on 1 host "
                   f"core: {one['steps']} policy steps + {one['resets']} resets in {one['t_step'] + one['t_reset']:.1f} s "
                   f"(host has {os.cpu_count()} cores)"),
        "value_excluding_resets": one["steps"] / one["t_step"],
        "vehicle_steps_per_s": rate * one["vehicles"], "vehicles_per_env": one["vehicles"],
        "host": os.uname().nodename, "measured_unix_time": int(time.time()),
    }
    n = os.cpu_count() or 1
    if max_procs:
        n = min(n, max_procs)
    if n > 1 and all_cores_budget_s > 0:
        import multiprocessing as mp
        ctx = mp.get_context("spawn")  # the parent may hold a HIP context: never fork it
        t0 = time.perf_counter()
        with ctx.Pool(n) as pool:
            res = pool.map(_worker, [(workload, all_cores_budget_s, 100 + k) for k in range(n)])
        wall = time.perf_counter() - t0
        steps = sum(r["steps"] for r in res)
        busy = max(r["t_step"] + r["t_reset"] for r in res)
        out["all_cores"] = {"value": steps / busy, "unit": "env-steps/s", "cores": n,
                            "sample": f"{n} processes (multiprocessing, spawn) x 1 env, {busy:.1f} s of stepping each "
                                      f"({wall:.1f} s wall incl. interpreter start-up and imports)",
                            "value_excluding_resets": sum(r["steps"] for r in res) / max(r["t_step"] for r in res)}
    return out
