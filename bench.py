#!/usr/bin/env python3
"""Benchmark of the hot path: batched AbstractEnv.step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: started WITHOUT a torch.distributed environment (no WORLD_SIZE) and with --gpus N > 1, bench.py
re-launches itself under torch.distributed.run with N ranks on 127.0.0.1 (a free port) and still prints ONE JSON line.

    python bench.py --gpus 8 --scaling strong                               # 4096 envs in TOTAL, 512 per GPU
    python bench.py --gpus 8 --workload v0_n100 --envs-per-gpu 1024         # BASELINE config 3: 8192 x 101 over 8 GPUs

Workload (BASELINE.json configs[1]): highway-fast-v0 semantics, 4096 batched envs x 50 IDM
vehicles (+1 ego = 51) per GPU, 4 lanes, DiscreteMetaAction (uniform random actions, pre-generated
and resident in HBM), Kinematics 5x5 observation, device-side spawn + auto-reset on
terminated/truncated.  One "step" == one batched policy step of every env == ONE launch of
hwy_step_kernel (5 simulation frames + observe + reward + done).  Weak scaling (default): every rank owns
4096 envs (--scaling strong: --envs-per-gpu is the TOTAL, block-partitioned over the ranks); with N>1 ranks the (obs, reward, terminated, truncated) blocks of every rank are gathered
to rank 0 over RCCL, --gather-every (16) steps per collective, overlapped with the following steps.

Timing protocol: W untimed warm-up steps, then --repeats (5) timed regions of EXACTLY K = --steps
(1000) steps each, every region bracketed by barrier + synchronize on both sides and reduced with MAX over ranks;
`ms_per_step` / `value` are the MEDIAN region (all regions are listed in `ms_per_step_repeats`).

CPU leg: when the reference package is present (build container: /root/reference or $HWY_REFERENCE_ROOT) the
UNMODIFIED reference is timed on the host cores (`cpu_baseline.kind == "reference"`, oracle/ref_bench.py); on a box
without it (the GPU box) the C port of the oracle is timed instead and the committed build-container measurement of the
reference (profiles/reference_cpu_baseline.json) is quoted next to it, labelled as coming from another box.
`python bench.py --cpu-baseline-only [--save-cpu-baseline PATH]` runs just that leg (no GPU needed).

Prints ONE JSON line (rank 0).  `value` = env-steps/s over all GPUs, inputs resident in HBM.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
VEHICLES_COUNT = 50
LANES = 4
EVENT_EVERY = 1  # every launch of the kernel-timing region (rounds 1-5: every 8th launch of the timed regions)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, peak engine clock (MI355X_MICROARCH.md)


def algorithmic_bytes_per_env_step(n_vehicles: int, agents: int, obs_floats: int = 25) -> int:
    """B_env = 72*N + 110*A bytes per env-step (110 = 4*V*F obs + reward, flags, action for
    the 5 x 5 observation; other observation shapes replace the 100 obs bytes)."""
    return 72 * n_vehicles + (4 * obs_floats + 10) * agents


def _kernel_build_id():
    """Identity of the kernel build being timed: sha256 over every kernel source + the hipcc flags
    (highwayenv_amd.build.kernel_source_hash), valid only if the shared library is not older than its sources."""
    from highwayenv_amd import build
    return None if build.is_stale() else build.kernel_source_hash()


def _load_counters(name: str, workload: str):
    """A committed rocprofv3 PMC summary (profiles/<name>) -- returned ONLY if it was recorded for the kernel build
    being timed (its `kernel_source_sha16` equals this build's) and for this workload: counters of another build
    say nothing about this one, and PMC counters cannot be read from inside this process."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):  # the newest round's file first
        path = os.path.join(ROOT, "profiles", name.replace("RND", rnd))
        try:
            d = json.load(open(path))
        except Exception:
            continue
        d = d.get(workload, d) if isinstance(d.get(workload), dict) else d
        if d.get("kernel_source_sha16") is not None and d.get("kernel_source_sha16") == _kernel_build_id():
            d["_file"] = os.path.relpath(path, ROOT)
            return d
    return None


def valu_view(envs_per_gpu: int, avg_kernel_s: float, workload: str = "fast"):
    """The compute-side view of the step kernel (it is VALU-issue bound, not HBM bound: DESIGN.md section 5): f64
    flops per launch from the committed SQ instruction counts (tools/pmc_sq.sh: add / mul = 1 flop, fma = 2, x 64 lanes x
    one wave per environment) over the launch duration measured in THIS run, against the 78.6 TFLOP/s f64 vector peak of
    MI355X, and the measured VALU issue utilisation.  None unless the counters belong to this kernel build."""
    d = _load_counters("RND_pmc_sq.json", workload)
    if d is None or d.get("envs") != envs_per_gpu:
        return None
    c = d["per_wave_per_step"]
    flop = (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2 * c["SQ_INSTS_VALU_FMA_F64"]) * 64 * envs_per_gpu
    achieved = flop / avg_kernel_s / 1e12
    # the ceiling that binds this kernel: what its VALU instructions occupy the SIMDs for.  MEASURED per class on this part
    # (tools/microbench/issue_bench.hip -> profiles/r05_issue_costs.json): f64 arithmetic, VOP3 compares / selects with an SGPR
    # operand, 64-bit shifts, conversions and cross-lane reads take 4 cycles; plain 32-bit VOP1/VOP2 take 2 in a run of their own
    # but 4 when interleaved 1:1 with f64 instructions, which is how these kernels use them (4.0 cycles per instruction of the
    # fma_f64 x and_b32 mix at every occupancy; SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.07 in the counters); f64 transcendental
    # seeds (v_rcp_f64 / v_rsq_f64) take 16.  So a launch cannot end before
    #   (4 x SQ_INSTS_VALU + 12 x SQ_INSTS_VALU_TRANS_F64) x environments / (1024 SIMDs x 2.4 GHz);
    # valu_issue = that floor / the measured launch duration.  `valu_issue_if_32bit_at_2_cycles` prices the instructions that
    # are neither f64 nor cross-lane at 2 cycles -- a bound the microbenchmark says these kernels cannot reach, kept for scale.
    # (the committed counters are sums over a launch / environments: per env-step, whatever the wavefronts per environment)
    trans = c.get("SQ_INSTS_VALU_TRANS_F64")  # (None -- not 0 -- where a summary lacks the counter: the floor is then the 4-cycle one)
    cycles_v1 = 4.0 * c["SQ_INSTS_VALU"]     # rounds 2-4: every VALU instruction 4 cycles (`valu_issue_4cyc`)
    cycles = cycles_v1 + 12.0 * (trans or 0.0)
    # ANY-instruction issue (round 6): a wavefront issues at most one instruction of ANY class per 4 cycles (dependent or not:
    # profiles/r06_issue_costs.json), so with <= 2 wavefronts per SIMD -- configs 3 and 4 -- what bounds a launch is the slowest
    # wavefront's own instruction stream, not the SIMD's VALU: issue_any = 4 x SQ_ACTIVE_INST_ANY cycles per wavefront / 2.4 GHz over
    # the launch duration (and over the mean wavefront's life, SQ_WAVE_CYCLES).  `binding_issue_limit` names the larger of the two.
    any_inst = c.get("SQ_ACTIVE_INST_ANY")
    wps = d.get("waves_per_simd")
    issue_any = (4.0 * any_inst / CLOCK_HZ / avg_kernel_s) if any_inst else None
    issue_floor_s = cycles * envs_per_gpu / (SIMDS * CLOCK_HZ)
    f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + (trans or 0.0)
    narrow = max(0.0, c["SQ_INSTS_VALU"] - f64 - c.get("SQ_INSTS_VALU_INT64", 0.0) - c.get("SQ_INSTS_VALU_CVT", 0.0))
    optimistic_s = (cycles - 2.0 * narrow) * envs_per_gpu / (SIMDS * CLOCK_HZ)
    wait = c.get("SQ_WAIT_ANY")
    return {"f64_flop_per_launch": flop, "achieved": achieved, "peak": 78.6, "unit": "TFLOP/s", "frac": achieved / 78.6,
            "valu_issue": issue_floor_s / avg_kernel_s, "valu_issue_floor_us": issue_floor_s * 1e6,
            "valu_issue_method_version": 2,  # 1 (rounds 2-4) = `valu_issue_4cyc` below; 2 (round 5 on) adds 12 cycles per f64 transcendental
            "valu_issue_4cyc": cycles_v1 * envs_per_gpu / (SIMDS * CLOCK_HZ) / avg_kernel_s,
            "issue_any": issue_any,
            "issue_any_of_mean_wavefront_life": (any_inst / c["SQ_WAVE_CYCLES"]) if any_inst else None,
            "issue_any_method": "4 x SQ_ACTIVE_INST_ANY cycles per wavefront / 2.4 GHz / avg_kernel_us (one wavefront's own issue port)",
            "binding_issue_limit": (None if issue_any is None else
                                    "issue_any (one wavefront's own stream)" if issue_any > issue_floor_s / avg_kernel_s else
                                    "valu_issue (the SIMD's VALU port, shared by its wavefronts)"),
            "waves_per_simd": wps,
            "valu_issue_method": (f"(4 x SQ_INSTS_VALU + 12 x SQ_INSTS_VALU_TRANS_F64) cycles per env-step x {envs_per_gpu} envs / "
                                  f"({SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz) / avg_kernel_us; per-class costs measured: "
                                  "profiles/r05_issue_costs.json"),
            "valu_issue_if_32bit_at_2_cycles": optimistic_s / avg_kernel_s,
            "wait_fraction_of_a_wavefront": (wait / c["SQ_WAVE_CYCLES"]) if wait else None,
            "valu_instructions_per_env_step": c["SQ_INSTS_VALU"], "salu_instructions_per_env_step": c["SQ_INSTS_SALU"],
            # (one wavefront per environment and the whole grid resident: the headline workload only)
            "valu_issue_utilisation_while_resident": (c["SQ_ACTIVE_INST_VALU"] * d.get("waves_per_simd", 4) / c["SQ_WAVE_CYCLES"]
                                                      if workload == "fast" else None),
            "valu_active_fraction_of_a_wavefront": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
            "source": d["_file"] + " (SQ counters of this kernel build and config)"}


def wide_kernel_runs(n_vehicles: int, tune=None) -> bool:
    """The engine's dispatch rule for the straight-road scenarios (csrc/hwy_kernels.hip: wide_kernel_applies; every bench workload
    observes Kinematics there): 64 < N <= 128 runs one wavefront per environment with two vehicles per thread (hwy_wave2.h) unless
    `--tune block_kernel=1` asks for the workgroup kernel; 128 < N <= 256 runs the workgroup kernel unless `--tune block_kernel=2` asks
    for three / four vehicles per thread."""
    tune = TUNE_IN_EFFECT if tune is None else tune
    bk = int(tune.get("block_kernel", 0))
    return (64 < n_vehicles <= 128 and bk != 1) or (128 < n_vehicles <= 256 and bk == 2)


TUNE_IN_EFFECT = {}  # (--tune KEY=VALUE of this run, set by main)


def kernel_resources_view(scenario: str, fast: bool, n_vehicles: int):
    """Registers / spills / LDS the compiler allocated to the step-kernel family this workload launches, read from the code object
    inside the library being timed (highwayenv_amd.build.kernel_resources; the WPE variants of a family that allocate the same are
    listed once).  None if the metadata cannot be read (it is a report, never a reason to fail the bench)."""
    try:
        from highwayenv_amd import build
        fam = ("hwy::hwy_ix_step_kernel<" if scenario == "intersection" else "hwy::hwy_net_step_kernel<" if scenario != "highway" else
               "hwy::hwy_step_wave_kernel<" if n_vehicles <= 64 else
               "hwy::hwy_step_wide_kernel<" if wide_kernel_runs(n_vehicles) else f"hwy::hwy_step_kernel<{(n_vehicles + 63) // 64}, ")
        tail = (", false>" if fast else ", true>") if fam.startswith("hwy::hwy_step_wave") else ""
        lib = os.environ.get("HWY_ENGINE_LIB") or build.LIB_PATH
        out, seen = {}, set()
        for name, r in build.kernel_resources(lib).items():
            if name.startswith(fam) and name.endswith(tail) and tuple(r.values()) not in seen:
                seen.add(tuple(r.values()))
                out[name.replace("hwy::", "")] = dict(r, waves_per_simd_by_vgprs=min(8, 512 // (((r["vgpr"] + 7) // 8) * 8)))
        return out or None
    except Exception:
        return None


def measured_traffic(workload: str, envs_per_gpu: int):
    """HBM bytes per step-kernel launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
    passes, calibrated on a known byte count in this kernel's access pattern: tools/traffic_probe.py,
    tools/traffic_report.py -> profiles/traffic_r02.json).  None if absent, recorded for another kernel build, or for
    another batch size."""
    d = _load_counters("traffic_RND.json", workload)
    if d is None or d.get("envs") != envs_per_gpu:
        return None
    return d["traffic_bytes_per_launch_calibrated"]


def cpu_baseline(workload: str, cfg_dict, fast: bool, scenario: str, have_gpu: bool = True):
    """The CPU leg, >= 30 s of host work.  Top level = what was timed ON THIS BOX IN THIS RUN: the unmodified
    reference where the package exists (kind "reference"), else the C port of the oracle (kind "port").  Next to it, always:
    ``"reference"`` -- the unmodified reference's figure as a first-class object with ``same_box: true`` (this run) or
    ``false`` (the committed build-container measurement, profiles/reference_cpu_baseline.json) -- and ``"port"``."""
    from oracle import ref_bench
    port = None
    if have_gpu or scenario != "intersection":  # (the intersection port takes its start states from the engine)
        port = cpu_baseline_port(cfg_dict, fast, scenario=scenario)
    if ref_bench.available():
        out = ref_bench.measure(workload, budget_s=30.0, all_cores_budget_s=30.0)
        out["reference"] = dict({k: v for k, v in out.items() if k != "all_cores"}, same_box=True, all_cores=out.get("all_cores"))
        out["same_box"] = True
        if port is not None:
            out["port"] = port
        return out
    out = dict(port)
    out["port"] = port
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "reference_cpu_baseline.json")))[workload]
        rec = {k: v for k, v in rec.items() if k not in ("port", "reference")}
        rec["same_box"] = False
        rec["note"] = ("NOT THE SAME BOX: the unmodified reference timed in the build container (the reference package does "
                       "not exist on this box); committed by `python bench.py --cpu-baseline-only --save-cpu-baseline`")
        out["reference"] = rec
    except Exception:
        out["reference"] = None
    return out


def cpu_baseline_port(cfg_dict, fast: bool = True, budget_s: float = 30.0, scenario: str = "highway"):
    """The CPU oracle (C port of the reference hot path, 1 thread) on a bounded sample of the
    same workload: same config, same spawn rule, random actions."""
    from highwayenv_amd import _abi, merge, spawn
    from oracle import oracle
    if scenario == "intersection":
        return cpu_baseline_intersection(cfg_dict, budget_s)
    E = 64 if fast else 8
    cfg = _abi.make_config(cfg_dict, E, fast=fast, scenario=scenario)
    if scenario == "highway":
        st0 = spawn.spawn_reference_stream(cfg, np.arange(E) + 7, cfg_dict["ego_spacing"], cfg_dict["vehicles_density"])
        episode_len = int(cfg_dict["duration"])
    else:
        st0 = merge.spawn_reference_stream(cfg, cfg_dict, scenario == "merge-generic", np.arange(E) + 7)
        episode_len = 8  # the ego reaches the end of the merge section (or crashes) after ~8-10 policy steps
    rng = np.random.default_rng(3)
    steps_done, t_used = 0, 0.0
    st = _abi.copy_state(st0)
    while t_used < budget_s:
        if steps_done % episode_len == 0:
            st = _abi.copy_state(st0)  # bounded stand-in for per-env resets: restart the batch
        acts = rng.integers(0, 5, size=(E, cfg.num_agents)).astype(np.int32)
        t0 = time.perf_counter()
        oracle.step(cfg, st, acts)
        t_used += time.perf_counter() - t0
        steps_done += 1
    rate = steps_done * E / t_used
    out = {"value": rate, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{E} envs x {steps_done} policy steps of the same workload on 1 host core "
                     f"(oracle/hwy_oracle{'' if scenario == 'highway' else '_net'}.c, {t_used:.1f} s; host has {os.cpu_count()} cores)",
           "vehicle_steps_per_s": rate * cfg.num_vehicles}
    # the same port on many host cores at once ("one env batch per core"): independent batches, one Python
    # thread each (ctypes releases the GIL inside the C call)
    import threading
    n_thr = min(64, os.cpu_count() or 1)
    if n_thr > 1:
        counts = [0] * n_thr
        stop_at = time.perf_counter() + 30.0

        def worker(k):
            st_k = _abi.copy_state(st0)
            rng_k = np.random.default_rng(100 + k)
            n = 0
            while time.perf_counter() < stop_at:
                if n % episode_len == 0:
                    st_k = _abi.copy_state(st0)
                oracle.step(cfg, st_k, rng_k.integers(0, 5, size=(E, cfg.num_agents)).astype(np.int32))
                n += 1
            counts[k] = n

        t0 = time.perf_counter()
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_thr)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        dt_all = time.perf_counter() - t0
        out["all_threads"] = {"value": sum(counts) * E / dt_all, "unit": "env-steps/s", "cores": n_thr,
                              "sample": f"{n_thr} threads x {E} envs, {dt_all:.1f} s"}
    return out


def cpu_baseline_intersection(cfg_dict, budget_s: float = 30.0):
    """oracle/hwy_oracle_ix.c (1 thread) on the same workload: start states from the engine's device reset, random
    actions, clearing and spawning with numpy draws in the reference's format."""
    from highwayenv_amd import _abi
    from highwayenv_amd import intersection as hix
    from highwayenv_amd.engine import Engine
    from oracle import oracle_ix
    E = 16
    cfg = _abi.make_config(cfg_dict, E, scenario="intersection")
    eng = Engine(cfg)
    eng.reset(base_seed=11)
    st_h = eng.get_state()
    eng.close()
    oc = oracle_ix.config_from_engine(cfg_dict, cfg, E)
    st0 = oracle_ix.state_from_engine(st_h, cfg)
    rng = np.random.default_rng(3)
    steps_done, t_used = 0, 0.0
    st = {k: v.copy() for k, v in st0.items()}
    while t_used < budget_s:
        if steps_done % int(cfg_dict["duration"]) == 0:
            st = {k: v.copy() for k, v in st0.items()}  # bounded stand-in for per-env resets: restart the batch
        acts = rng.integers(0, 3, size=E).astype(np.int32)
        route = np.stack([rng.permutation(4)[:2] for _ in range(E)]).astype(np.float64)
        draws = np.column_stack([rng.uniform(size=E), route, rng.normal(size=(E, 2)), rng.uniform(3.5, 4.5, size=E),
                                 np.zeros((E, 2))])
        t0 = time.perf_counter()
        oracle_ix.step(oc, st, acts)
        oracle_ix.clear_spawn(oc, st, draws, np.full(E, 6, np.int32))
        t_used += time.perf_counter() - t0
        steps_done += 1
    rate = steps_done * E / t_used
    return {"value": rate, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{E} envs x {steps_done} policy steps of the same workload on 1 host core "
                      f"(oracle/hwy_oracle_ix.c, {t_used:.1f} s; host has {os.cpu_count()} cores)",
            "vehicle_slots": cfg.num_vehicles}


FRONTEND_IDS = {"fast": "highway-fast-v0", "v0": "highway-v0", "v0_n100": "highway-v0", "v0_n200": "highway-v0", "merge": "merge-v0",
                "merge_ma4": "merge-generic-v0", "intersection": "intersection-v0", "intersection_kin": "intersection-v0"}


def frontend_view(workload: str, cfg_dict: dict, E: int, device: int, steps: int = 300, warm: int = 200) -> dict:
    """The drop-in boundary as a USER sees it ("wall-clock around the Python step()"; the reference's seam is
    AbstractEnv.step, envs/common/abstract.py:259-285): the same workload stepped through the package's Python classes, wall clock
    per step() call over `steps` calls after `warm` (the engine picks its issue-priority turn there), synchronised at both ends.
      batched_numpy  -- Batched*Env.step(numpy actions [E]) -> numpy outputs: H2D actions, the launch, ONE D2H copy of the packed
                        outputs, per step (device spawn + auto-reset, like the headline)
      vector_torch   -- HighwayVectorEnv(output="torch").step(int32 device tensor) -> torch views of the engine's output planes:
                        the device-pointer path behind the gymnasium.vector interface, run on the engine's own stream
    Never part of `value`."""
    import torch
    from highwayenv_amd import envs as _envs
    from highwayenv_amd.vector import HighwayVectorEnv
    out = {"workload": workload, "envs": E, "steps": steps, "unit": "us per step() call (wall clock)"}
    env_id = FRONTEND_IDS[workload]
    n_act = 3 if "intersection" in workload else 5
    try:
        cls = _envs.batched_class(env_id)
        env = cls(dict(cfg_dict), num_envs=E, device=device, spawn_mode="device", autoreset=True)
        env.reset(seed=0)
        A = env._hcfg.num_agents
        acts = np.random.default_rng(7).integers(0, n_act, size=(steps + warm, E, A)).astype(np.int32)
        for t in range(warm):
            env.step(acts[t])
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for t in range(warm, warm + steps):
            env.step(acts[t])
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        out["batched_numpy"] = {"class": cls.__name__, "us_per_step": dt / steps * 1e6, "env_steps_per_s": steps * E / dt}
        env.close()
    except Exception as ex:  # a front end that cannot run must not cost the headline its line
        out["batched_numpy"] = {"error": repr(ex)}
    try:
        lane = torch.cuda.Stream(device=device)
        venv = HighwayVectorEnv(env_id, num_envs=E, config=dict(cfg_dict), output="torch", device=device, stream=lane)
        venv.reset(seed=0)
        A = venv.env._hcfg.num_agents
        acts_d = torch.randint(0, n_act, (steps + warm, E, A) if A > 1 else (steps + warm, E), device=f"cuda:{device}", dtype=torch.int32)
        torch.cuda.synchronize(device)
        with torch.cuda.stream(lane):
            for t in range(warm):
                venv.step(acts_d[t])
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for t in range(warm, warm + steps):
                venv.step(acts_d[t])
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
        out["vector_torch"] = {"class": "HighwayVectorEnv(output='torch')", "us_per_step": dt / steps * 1e6,
                               "env_steps_per_s": steps * E / dt}
        venv.close()
    except Exception as ex:
        out["vector_torch"] = {"error": repr(ex)}
    return out


def workload_config(workload: str):
    """(cfg_dict, fast, scenario) of a --workload, in the reference's config vocabulary."""
    from highwayenv_amd import _abi
    fast = workload == "fast"
    scenario = "highway"
    if fast:
        cfg_dict = _abi.highway_fast_default_config()
        cfg_dict.update({"vehicles_count": VEHICLES_COUNT, "lanes_count": LANES})
    elif workload in ("merge", "merge_ma4"):
        from highwayenv_amd import merge
        if workload == "merge":
            scenario, cfg_dict = "merge", merge.merge_default_config()
        else:
            scenario, cfg_dict = "merge-generic", merge.merge_generic_default_config()
            cfg_dict.update({"lanes_count": 4, "vehicles_count": 40, "controlled_vehicles": 4,
                             "action": {"type": "MultiAgentAction", "action_config": {"type": "DiscreteMetaAction"}},
                             "observation": {"type": "MultiAgentObservation",
                                             "observation_config": {"type": "Kinematics"}}})
    elif workload in ("intersection", "intersection_kin"):
        from highwayenv_amd import intersection as hix
        scenario, cfg_dict = "intersection", hix.intersection_default_config()
        cfg_dict.update({"max_vehicles": 30})
        if workload == "intersection":  # BASELINE config 4: IntersectionEnv({"observation": {"type": "OccupancyGrid"}})
            cfg_dict["observation"] = {"type": "OccupancyGrid"}
    else:
        cfg_dict = _abi.highway_default_config()
        if workload == "v0_n100":
            cfg_dict.update({"vehicles_count": 100})
        elif workload == "v0_n200":  # beyond BASELINE's configurations: the N > 128 path (four vehicles per thread, round 5)
            cfg_dict.update({"vehicles_count": 200})
    return cfg_dict, fast, scenario


def launcher_command(argv, n_gpus: int, port: int):
    """`python bench.py --gpus N ...` started without a torch.distributed environment: the command it re-launches itself with
    (one rank per GPU of this node, rendezvous on 127.0.0.1 -- the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(argv, n_gpus: int) -> int:
    """Re-exec under torch.distributed.run and pass the ranks' stdout (rank 0's ONE JSON line) and stderr through."""
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n_gpus:
        raise SystemExit(f"--gpus {n_gpus}: only {have} GPU(s) visible on this node")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    cmd = launcher_command(argv, n_gpus, _free_port())
    print("bench.py: no torch.distributed environment, launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


SECONDARY = [("config 3 shard: highway-v0, 1024 envs x 101 vehicles", ["--workload", "v0_n100", "--envs-per-gpu", "1024"]),
             ("config 4: intersection-v0, 2048 envs x 30 slots, OccupancyGrid", ["--workload", "intersection", "--envs-per-gpu", "2048"]),
             ("config 5: merge-generic multi-agent, 4096 envs x 43 slots, 4 agents", ["--workload", "merge_ma4"]),
             ("highway-v0 defaults, 4096 envs x 51 vehicles", ["--workload", "v0"])]


def secondary_workloads(steps: int = 200, repeats: int = 3, budget_s: float = 100.0) -> dict:
    """BASELINE's other single-GPU configurations, one short run of this same script each (its own process, after the headline's
    timed regions; same box, same build), so that the driver's record carries their numbers too.  Never part of `value`.
    `budget_s` bounds ALL legs together (a leg takes ~8 s; one that hangs gets what is left of the budget, the legs after it are
    reported as skipped): the headline's one JSON line is never held back by more than that."""
    import subprocess
    out = {}
    t_end = time.perf_counter() + budget_s
    for name, argv in SECONDARY:
        left = t_end - time.perf_counter()
        if left < 5.0:
            out[argv[1]] = {"workload": name, "error": f"skipped: the {budget_s:.0f} s budget of the secondary legs was spent"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), *argv, "--steps", str(steps), "--repeats", str(repeats), "--warmup", "20",
               "--settle-ms", "100", "--no-cpu-baseline", "--no-secondary", "--no-frontend", "--rollout-k", "0"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rf = d["roofline"]
            v = rf.get("valu") or {}
            out[argv[1]] = {"workload": name, "ms_per_step": d["ms_per_step"], "ms_per_step_device": d["ms_per_step_device"],
                            "value": d["value"], "unit": d["unit"], "vehicle_steps_per_s": d["vehicle_steps_per_s"],
                            "steps": steps, "repeats": repeats, "avg_kernel_us": rf["avg_kernel_us"], "kernel": rf["kernel"],
                            "roofline_frac_hbm": rf["frac"], "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                            # the ceiling that binds these kernels (None unless profiles/ holds SQ counters of THIS build)
                            "valu_issue": v.get("valu_issue"), "valu_issue_floor_us": v.get("valu_issue_floor_us"),
                            "issue_any": v.get("issue_any"), "issue_any_of_mean_wavefront_life": v.get("issue_any_of_mean_wavefront_life"),
                            "binding_issue_limit": v.get("binding_issue_limit"), "waves_per_simd": v.get("waves_per_simd"),
                            "wait_fraction_of_a_wavefront": v.get("wait_fraction_of_a_wavefront"),
                            "valu_active_fraction_of_a_wavefront": v.get("valu_active_fraction_of_a_wavefront")}
        except Exception as ex:  # a report next to the headline, never a reason to lose the headline
            out[argv[1]] = {"workload": name, "error": repr(ex)[:300]}
    return out


class GpuPlatform:
    """What the rank body needs from the machine: a device, a stream with an engine on it, events, the process-group backend.
    tests/test_bench_dryrun.py swaps in a CPU stand-in (gloo, a fake engine) to run the WHOLE control flow of an N-rank bench --
    shard, two alternating output buffers, --gather-every, the per-step-gather leg, the one JSON line -- without a GPU."""
    backend = "nccl"

    def check(self):
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")

    def device(self, local_rank):
        import torch
        torch.cuda.set_device(local_rank)
        return torch.device("cuda", local_rank)

    def init_process_group(self, dist, rank, world, dev):
        dist.init_process_group(self.backend, rank=rank, world_size=world, device_id=dev)

    def make_engine(self, cfg, local_rank, dev):
        # ONE stream for everything: the engine's launches, torch's tensor ops and the event torch.distributed records before it
        # hands a buffer to RCCL.  It must be a real stream object: the handle of torch's default stream is NULL, which hwy_create
        # reads as "create your own" -- and an engine-owned stream is not ordered with torch's.
        import torch
        from highwayenv_amd.engine import Engine
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        return Engine(cfg, device=local_rank, stream=stream.cuda_stream), stream

    def synchronize(self, dev):
        import torch
        torch.cuda.synchronize(dev)

    def event(self):
        import torch
        return torch.cuda.Event(enable_timing=True)


def main(argv=None, platform=None, emit=None):
    platform = platform or GpuPlatform()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps PER REGION (SURVEY 8d: >= 1000)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--settle-ms", type=float, default=200.0,
                    help="untimed: after the --warmup steps keep stepping until the GPU has been busy this long (engine clock ramp-up "
                         "after idle); 0 = off.  Reported as `settle_steps`")
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU,
                    help="environments per GPU (weak scaling, the default); with --scaling strong: the TOTAL over all GPUs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every GPU steps --envs-per-gpu environments; strong: that many in TOTAL, block-partitioned "
                         "over the ranks like highwayenv_amd.dist.shard_range")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="run only the CPU leg (needs no GPU): the unmodified reference where it is installed, else the C port")
    ap.add_argument("--save-cpu-baseline", metavar="PATH", default=None,
                    help="with --cpu-baseline-only: merge the result into this JSON file under the workload's name "
                         "(profiles/reference_cpu_baseline.json is what a box without the reference quotes)")
    ap.add_argument("--gather-every", type=int, default=16,
                    help="N > 1: the (obs, reward, done) blocks of this many consecutive steps travel to rank 0 in one RCCL "
                         "gather (every step's outputs still reach rank 0 inside the timed region); 1 = one gather per step")
    ap.add_argument("--workload", choices=["fast", "v0", "v0_n100", "v0_n200", "merge_ma4", "merge", "intersection", "intersection_kin"], default="fast",
                    help="fast = BASELINE config 2 (the headline metric); v0_n100 = the per-GPU shard of config 3 "
                         "(highway-v0, 101 vehicles, 15 frames/step, full pairwise collisions; use --envs-per-gpu 1024); "
                         "merge_ma4 = BASELINE config 5 (merge-generic, 4 lanes, 40 traffic vehicles, 4 controlled agents "
                         "per env); merge = merge-v0 defaults; intersection = BASELINE config 4's world model "
                         "(intersection-v0, 30 vehicle slots, OccupancyGrid 4 x 11 x 11 obs; use --envs-per-gpu 2048); intersection_kin = "
                         "the same with intersection-v0's default Kinematics 15 x 7 observation")
    ap.add_argument("--rollout-k", type=int, default=16,
                    help="also time hwy_rollout_device with this many policy steps per launch (reported as rollout_k<K> next to the "
                         "headline; 0 / 1 = skip)")
    ap.add_argument("--split-batch", type=int, default=0,
                    help="N=1 only: also time the batch as S independent sub-batches (one engine and one HIP stream each, stepped "
                         "round-robin: the tail of one sub-batch's launch overlaps the body of the next one's), reported as "
                         "`split_batch_sS` next to the headline.  Off by default: its launches carry the headline kernel's name, "
                         "so they would mix into a rocprofv3 --stats average of the default command")
    ap.add_argument("--shape", default=None, metavar="VEHICLES,LANES",
                    help="developer knob: another traffic size / lane count for --workload fast (e.g. 20,3 = BASELINE config 1's shape "
                         "batched); the line's config.workload says so.  Never the headline.")
    ap.add_argument("--no-frontend", action="store_true",
                    help="skip the `frontend` legs (the Python classes a user of the reference switches to, timed on the same workload)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N=1, --workload fast only: skip the short legs on BASELINE's other single-GPU configurations (config 3's "
                         "per-GPU shard, config 4, config 5) that are reported NEXT TO the headline as `secondary_workloads`")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="hwy_config.tune_* knob (block_kernel, waves_per_eu, ix_no_helpers, ix_no_prewarm, extra_lds, prio_shift, "
                         "ix_prewarm_frames: highwayenv_amd._abi.TUNING_KEYS); selects a kernel variant, never changes a result; repeatable")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.cpu_baseline_only:
        raise SystemExit(self_launch(sys.argv[1:] if argv is None else list(argv), args.gpus))
    tuning = {kv.split("=", 1)[0]: int(kv.split("=", 1)[1]) for kv in args.tune}
    if args.shape:
        global VEHICLES_COUNT, LANES
        VEHICLES_COUNT, LANES = (int(x) for x in args.shape.split(","))
    TUNE_IN_EFFECT.clear()
    TUNE_IN_EFFECT.update(tuning)
    cfg_dict, fast, scenario = workload_config(args.workload)
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints its version banner through C
    # stdio when a communicator is created), so everything else is sent to stderr: fd 1 is pointed at fd 2 for the whole
    # run and the JSON line goes to the saved descriptor at the end.
    if emit is None:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)

        def emit(obj) -> None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(obj) + "\n").encode())

    if args.cpu_baseline_only:
        out = cpu_baseline(args.workload, cfg_dict, fast, scenario, have_gpu=False)
        emit({"workload": args.workload, "cpu_baseline": out})
        if args.save_cpu_baseline and out.get("kind") == "reference":
            try:
                allw = json.load(open(args.save_cpu_baseline))
            except Exception:
                allw = {}
            allw[args.workload] = out
            json.dump(allw, open(args.save_cpu_baseline, "w"), indent=1)
        return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the torch.distributed environment has WORLD_SIZE={world}")
    platform.check()
    dev = platform.device(local_rank)
    # HWY_BENCH_FORCE_DIST=1 (developer knob): run the RCCL gather path even with a single rank
    use_dist = world > 1 or os.environ.get("HWY_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        platform.init_process_group(dist, rank, world, dev)

    from highwayenv_amd import _abi
    from highwayenv_amd.dist import PackedStepOutputs

    E = args.envs_per_gpu
    if args.scaling == "strong":  # --envs-per-gpu environments in TOTAL: this rank's block (remainders to the lowest ranks)
        from highwayenv_amd.dist import shard_range
        if args.envs_per_gpu % world:
            raise SystemExit(f"--scaling strong: {args.envs_per_gpu} environments do not split evenly over {world} ranks "
                             "(the packed gather needs equal blocks)")
        E = len(shard_range(args.envs_per_gpu, world, rank))
    cfg = _abi.make_config(cfg_dict, E, fast=fast, scenario=scenario, tuning=tuning)
    N, A = cfg.num_vehicles, cfg.num_agents
    spawn_kw = ({"ego_spacing": cfg_dict["ego_spacing"], "vehicles_density": cfg_dict["vehicles_density"]}
                if scenario == "highway" else {})

    eng, stream = platform.make_engine(cfg, local_rank, dev)
    eng.reset(base_seed=1_000_003 * (rank + 1), **spawn_kw)
    eng.set_autoreset(True, base_seed=77_000_001 * (rank + 1), **spawn_kw)

    R = max(1, args.repeats)
    total = args.warmup + R * args.steps
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    n_actions = 3 if scenario == "intersection" else 5
    actions = torch.randint(0, n_actions, (total, E, A), generator=g, device=dev, dtype=torch.int32)
    # two alternating output buffers of K step blocks each: the RCCL gather of one buffer (async, on RCCL's stream)
    # overlaps the step kernels that fill the other one.  K = --gather-every (1 without a collective).
    K = max(1, args.gather_every) if use_dist else 1

    def make_stepper(K):
        outs = [PackedStepOutputs(cfg, dev, world, rank, force_collective=use_dist, depth=K) for _ in range(2)]
        works = [None, None]
        pending = [False, False]

        def one_step(t: int) -> None:
            k, slot = (t // K) & 1, t % K
            if slot == 0 and works[k] is not None:
                works[k].wait()  # stream-level: buffer k was gathered, the engine may overwrite it
                works[k] = None
            eng.step_device(actions[t].data_ptr(), *outs[k].pointers(slot))
            pending[k] = True
            if use_dist and slot == K - 1:
                works[k] = outs[k].gather_async()
                pending[k] = False

        def drain() -> None:
            for k in (0, 1):
                if use_dist and pending[k]:  # a partly filled buffer at the end of a region still travels
                    works[k] = outs[k].gather_async()
                    pending[k] = False
                if works[k] is not None:
                    works[k].wait()
                    works[k] = None

        def fence() -> None:
            drain()
            if use_dist:
                dist.barrier()
            platform.synchronize(dev)

        return one_step, fence, outs

    one_step, fence, outs = make_stepper(K)

    for t in range(args.warmup):
        one_step(t)
    # Clock settling (untimed, like the warm-up; --settle-ms 0 switches it off): a short command line (the driver's --warmup 5
    # --steps 20 is 5 ms of GPU work in all) would otherwise time the first milliseconds after an idle period, while the engine
    # clock is still ramping up -- measured 44.4 us per step there against 41.7 us in steady state on the same box
    # (profiles/r04_history.md).  More of the same untimed warm-up steps until the GPU has been busy for --settle-ms.
    # The settling time counts from the END of the warm-up launches (a box's first launch loads the code object: hundreds of
    # milliseconds of host time on a cold or busy box, during which the GPU does nothing -- counted from the first launch, as
    # rounds 4-5 did, such a box skipped the settling altogether and timed the regions while the engine was still sampling its
    # issue-priority turns: 48.5 us per step against 40.6 on the next box), and goes on -- bounded -- until the engine has chosen
    # its turn (hwy_get_prio_turn state 1 = still sampling; 0 = no selection, 2 = chosen).
    platform.synchronize(dev)
    t_settle0 = time.perf_counter()
    settle_steps = 0
    while args.settle_ms > 0:
        busy_ms = (time.perf_counter() - t_settle0) * 1e3
        want = busy_ms < args.settle_ms or (eng.prio_turn()[1] == 1 and busy_ms < 20 * args.settle_ms)
        more = torch.tensor([1.0 if want else 0.0], device=dev)
        if use_dist:  # every rank must run the SAME number of rounds (each round holds collectives): go on while any rank wants to
            dist.all_reduce(more, op=dist.ReduceOp.MAX)
        if more.item() == 0.0:
            break
        for t in range(max(args.warmup, 1)):
            one_step(t)
        settle_steps += max(args.warmup, 1)
        platform.synchronize(dev)
    fence()
    # HIP events on every 8th launch of the timed regions: the engine hands the pair to hipExtLaunchKernelGGL, which records the
    # DISPATCH's own begin / end timestamps into them (the clock readings rocprofv3 --kernel-trace reports) on the launch stream
    # (the timed regions carry NO events since round 6; the kernel's own duration is measured after them, see below)
    prio_turn, prio_state = eng.prio_turn()   # chosen during the warm-up launches (hwy_get_prio_turn)
    region_s, region_dev_ms = [], []
    for r in range(R):
        t_first = args.warmup + r * args.steps
        # the same region on the DEVICE timeline too: an event pair on the stream the engine launches on, around the K launches
        # (host wall = this + the latency of fence()'s drain / barrier / synchronize, which a short region does not amortise)
        ev0, ev1 = platform.event(), platform.event()
        t0 = time.perf_counter()
        ev0.record(stream)
        for t in range(t_first, t_first + args.steps):
            one_step(t)
        ev1.record(stream)
        fence()
        dt_r = time.perf_counter() - t0
        region_dev_ms.append(ev0.elapsed_time(ev1))
        el = torch.tensor([dt_r], device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        region_s.append(el.item())
    elapsed = float(np.median(region_s))
    # The dominant kernel's own duration: one more (untimed) region of the same loop in which EVERY launch carries the HIP event pair
    # hipExtLaunchKernelGGL fills with the dispatch's begin / end timestamps -- what rocprofv3 --kernel-trace reports, under the
    # same condition (every dispatch signalled).  Rounds 1-5 sampled every 8th launch of the timed regions: the sampled launches
    # alone were serialised against their neighbours, which made their average exceed the step they are part of (41.50 us against a
    # 40.96 us device step).
    kernel_ms, launches = 0.0, 0
    if os.environ.get("HWY_BENCH_NO_EVENTS") != "1":
        # (at least 300 launches whatever --steps says, over the action rows of the timed regions, behind 10 launches that are not
        #  counted: the driver's --steps 20 averaged the first 20 launches after the switch to signalled dispatches, 41.3 us where
        #  the device step of the same run was 39.6)
        n_k = min(max(args.steps, 300), 1000)
        eng.profile_enable(EVENT_EVERY)
        for j in range(10):
            one_step(args.warmup + j % (R * args.steps))
        fence()
        ms0, n0 = eng.profile_read()  # (totals since profile_enable)
        for j in range(n_k):
            one_step(args.warmup + j % (R * args.steps))
        fence()
        ms1, n1 = eng.profile_read()
        kernel_ms, launches = ms1 - ms0, n1 - n0
        eng.profile_enable(0)
    # N > 1: the same loop with ONE gather per step (what a policy that needs every step's outputs on rank 0 before it can
    # act would see), reported next to the batched number
    per_step_gather = None
    if use_dist and K != 1:
        one_step1, fence1, _ = make_stepper(1)
        n1 = min(args.steps, 300)
        for j in range(20):   # (action rows of the timed regions, re-used: a short smoke run stages fewer than 20 of them)
            one_step1(args.warmup + j % (R * args.steps))
        fence1()
        t0 = time.perf_counter()
        for t in range(args.warmup, args.warmup + n1):
            one_step1(t)
        fence1()
        el1 = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(el1, op=dist.ReduceOp.MAX)
        per_step_gather = {"steps": n1, "ms_per_step": el1.item() / n1 * 1e3, "value": n1 * E * world / el1.item(),
                           "unit": "env-steps/s", "gather": "one RCCL gather per step"}

    # K policy steps per launch with pre-staged actions (hwy_rollout_device: open-loop rollouts / action repeat): reported NEXT TO
    # the headline, never instead of it -- `value` stays one launch per step
    rollout = None
    if world == 1 and args.rollout_k > 1 and R * args.steps > args.rollout_k:  # (needs more staged action blocks than one call reads)
        Kr = args.rollout_k
        n_calls = max(1, args.steps // Kr)
        r_obs = torch.empty((Kr, E, A, *_abi.obs_shape(cfg)), dtype=torch.float32, device=dev)
        r_rew = torch.empty((Kr, E, A), dtype=torch.float64, device=dev)
        r_term = torch.empty((Kr, E), dtype=torch.uint8, device=dev)
        r_trunc = torch.empty((Kr, E), dtype=torch.uint8, device=dev)
        r_speed = torch.empty((Kr, E, A), dtype=torch.float64, device=dev)
        r_crashed = torch.empty((Kr, E, A), dtype=torch.uint8, device=dev)

        def roll(c):
            t0_ = args.warmup + (c * Kr) % (R * args.steps - Kr)
            assert 0 <= t0_ and t0_ + Kr <= actions.shape[0]
            eng.rollout_device(Kr, actions[t0_].data_ptr(), r_obs.data_ptr(), r_rew.data_ptr(), r_term.data_ptr(),
                               r_trunc.data_ptr(), r_speed.data_ptr(), r_crashed.data_ptr())
        for c in range(2):
            roll(c)
        torch.cuda.synchronize(dev)
        reg = []
        for r in range(R):
            t0 = time.perf_counter()
            for c in range(n_calls):
                roll(r * n_calls + c)
            torch.cuda.synchronize(dev)
            reg.append(time.perf_counter() - t0)
        el = float(np.median(reg))
        rollout = {"k_steps_per_launch": Kr, "launches_per_region": n_calls, "ms_per_step": el / (n_calls * Kr) * 1e3,
                   "value": n_calls * Kr * E / el, "unit": "env-steps/s",
                   "ms_per_step_repeats": [x / (n_calls * Kr) * 1e3 for x in reg],
                   "what": "hwy_rollout_device: K policy steps per launch with pre-staged actions, every step's outputs written; "
                           "bit-identical to K one-step launches (tests/test_rollout.py)"}

    # The same batch as S independent sub-batches (S engines of E / S environments, one stream each, stepped round-robin with the
    # same pre-staged device actions): what an actor that alternates between sub-batches sees.  Environments are independent,
    # so sub-batch s of step k + 1 only waits for sub-batch s of step k -- the latency-bound tail of one launch (the slowest
    # wavefronts of a launch run alone on their SIMDs) overlaps the next sub-batch.  Reported NEXT TO the headline.
    split = None
    S = args.split_batch
    if world == 1 and S > 1 and E % S == 0:
        Es = E // S
        cfg_s = _abi.make_config(cfg_dict, Es, fast=fast, scenario=scenario, tuning=tuning)
        from highwayenv_amd.engine import Engine
        sub_streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        subs = []
        for k in range(S):
            e_ = Engine(cfg_s, device=local_rank, stream=sub_streams[k].cuda_stream)
            e_.reset(base_seed=1_000_003 * (k + 11), **spawn_kw)
            e_.set_autoreset(True, base_seed=77_000_001 * (k + 11), **spawn_kw)
            subs.append((e_, PackedStepOutputs(cfg_s, dev, 1, 0, force_collective=False, depth=1)))
        n_s = min(args.steps, 400)

        a_base, a_step, a_sub = actions.data_ptr(), E * A * 4, Es * A * 4   # (raw pointers: the loop is launch-bound on the host)
        sub_ptrs = [o_.pointers(0) for _, o_ in subs]

        def sub_steps(t_lo, t_hi):
            for t in range(t_lo, t_hi):
                for k, (e_, _) in enumerate(subs):
                    e_.step_device(a_base + t * a_step + k * a_sub, *sub_ptrs[k])
        sub_steps(0, args.warmup)
        torch.cuda.synchronize(dev)
        reg = []
        for r in range(R):
            t0 = time.perf_counter()
            sub_steps(args.warmup, args.warmup + n_s)
            torch.cuda.synchronize(dev)
            reg.append(time.perf_counter() - t0)
        el = float(np.median(reg))
        split = {"sub_batches": S, "envs_per_sub_batch": Es, "steps": n_s, "ms_per_step": el / n_s * 1e3, "value": n_s * E / el,
                 "unit": "env-steps/s", "ms_per_step_repeats": [x / n_s * 1e3 for x in reg],
                 "what": "S engines on S streams stepped round-robin (one launch per sub-batch and step, actions pre-staged on the "
                         "device): every environment advances one policy step per step, sub-batches are not synchronised with "
                         "each other"}
        for e_, _ in subs:
            e_.close()

    # PCIe-inclusive rate of the host-pointer entry point (hwy_step: H2D actions, kernel, D2H results, sync);
    # reported for DESIGN.md, never as `value`
    host_rate = None
    if world == 1:
        acts_h = np.random.default_rng(5).integers(0, n_actions, size=(E, A)).astype(np.int32)
        for _ in range(5):
            eng.step(acts_h)
        th = time.perf_counter()
        for _ in range(40):
            eng.step(acts_h)
        host_rate = 40 * E / (time.perf_counter() - th)

    # statistics of the run (sanity: the workload really stepped and reset)
    term = outs[((total - 1) // K) & 1].terminated((total - 1) % K).sum().item()

    if rank == 0:
        env_steps = args.steps * E * world
        value = env_steps / elapsed
        b_env = algorithmic_bytes_per_env_step(N, A, int(np.prod(_abi.obs_shape(cfg))))
        # The dominant kernel's launch duration: the dispatch timestamps of every 8th launch (HIP events filled in by
        # hipExtLaunchKernelGGL), reported AS MEASURED.  A kernel cannot take longer than the step that contains it: if the
        # events say so anyway the run is flagged (`event_exceeds_wall_step`) instead of clamped.  Without events
        # (HWY_BENCH_NO_EVENTS=1, a developer knob): the wall step.
        wall_step_s = elapsed / args.steps
        event_kernel_s = (kernel_ms / 1e3 / launches) if launches else None
        avg_kernel_s = event_kernel_s if event_kernel_s and event_kernel_s > 0 else wall_step_s
        achieved = b_env * E / avg_kernel_s / 1e9
        line = {
            "metric": "env-steps/s",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "settle_ms": args.settle_ms,
            "repeats": R,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_repeats": [x / args.steps * 1e3 for x in region_s],
            "ms_per_step_device": float(np.median(region_dev_ms)) / args.steps,
            "ms_per_step_device_repeats": [x / args.steps for x in region_dev_ms],
            "timing": (f"median of {R} regions of {args.steps} steps (barrier + synchronize on both sides, max over ranks); "
                       "ms_per_step / value = host wall clock around each region incl. the closing fence; ms_per_step_device = the "
                       "same regions between two events on the engine's stream (rank 0)"),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"highway-fast-v0, {E} envs/GPU x {VEHICLES_COUNT} IDM vehicles (+1 ego, N={N}), "
                                    f"{LANES} lanes, 5 frames/step, DiscreteMetaAction random actions, Kinematics 5x5 obs, "
                                    "device spawn + auto-reset") if fast else
                                   (f"intersection-v0, {E} envs/GPU x {N} vehicle slots (4-way junction of 20 straight / circular lanes, "
                                    f"planned routes, RegulatedRoad priorities, vehicles cleared and spawned every policy step on the "
                                    f"device), {cfg.frames_per_step} frames/step, full pairwise collisions, random actions (3), "
                                    + (f"OccupancyGrid {'x'.join(str(k) for k in _abi.obs_shape(cfg))} obs (presence, vx, vy, on_road)"
                                       if cfg.obs_type == _abi.OBS_OCCUPANCY_GRID else
                                       f"Kinematics {cfg.obs_vehicles} x {cfg.obs_features} absolute obs") + ", device reset + auto-reset") if scenario == "intersection" else
                                   (f"{'merge-v0' if scenario == 'merge' else 'merge-generic-v0'}, {E} envs/GPU x {N} slots "
                                    f"({A} controlled MDP vehicles, {N - A - 2} IDM traffic slots of which the rejection-sampled "
                                    f"spawn fills most, 1 merging IDM vehicle, 1 obstacle), {cfg.lanes_count} highway lanes + ramp "
                                    f"({cfg.net_lanes} network lanes), {cfg.frames_per_step} frames/step, full pairwise collisions, "
                                    f"random actions, per-agent Kinematics 5x5 obs, device spawn + auto-reset") if scenario != "highway" else
                                   (f"highway-v0, {E} envs/GPU x {N - A} IDM vehicles (+1 ego, N={N}), 4 lanes, 15 frames/step, full "
                                    "pairwise collisions, random actions, Kinematics 5x5 obs, device spawn + auto-reset"),
                       "envs_per_gpu": E, "vehicles_per_env": N, "parallelism": f"env-sharded x{world}",
                       "gather": (f"one RCCL gather of every rank's (obs, reward, done) blocks to rank 0 per {K} steps"
                                  if use_dist else "none (single rank)"),
                       "world_size_reported_by_the_process_group": dist.get_world_size() if use_dist else 1,
                       "issue_priority_turn": {"turn": prio_turn, "unit": ("off" if prio_turn <= 0 else "2^k clock ticks" if prio_turn <= 30
                                                                            else "k x 64 clock ticks"),
                                               "chosen_by": ("--tune prio_shift" if "prio_shift" in tuning else
                                                             {0: "scenario default (no selection: turns off or not applicable)",
                                                              1: "engine, still sampling", 2: "engine (timed its first launches)"}[prio_state])}},
            "gather_every_1": per_step_gather,
            f"rollout_k{args.rollout_k}": rollout,
            **({f"split_batch_s{args.split_batch}": split} if split else {}),
            "vehicle_steps_per_s": value * N,
            "vehicle_steps_per_s_excl_ego": value * (N - A),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(args.workload, E),
                         "kernel_source_sha16": _kernel_build_id(),
                         "kernel": ("hwy_ix_step_kernel  (one 64-wide wavefront per env)" if scenario == "intersection" else
                                    "hwy_net_step_kernel  (one 64-wide wavefront per env)" if scenario != "highway" else
                                    f"hwy_step_wave_kernel<WPE,{str(not fast).lower()}>  (one 64-wide wavefront per env; every WPE variant is the same "
                                    "102 / 128-VGPR code)" if N <= 64 else
                                    f"hwy_step_wide_kernel<{(N + 63) // 64},{2 if N <= 128 else 1}>  (one 64-wide wavefront per env, {(N + 63) // 64} vehicles per thread)" if wide_kernel_runs(N) else
                                    f"hwy_step_kernel<{(N + 63) // 64},WPE>  ({(N + 63) // 64} wavefronts per env)"), "avg_kernel_us": avg_kernel_s * 1e6, "launches": launches, "timed_every": EVENT_EVERY,
                         "avg_kernel_us_method": ("mean over ALL launches of a separate region after the timed ones, each carrying the HIP "
                                                  "start/stop events hipExtLaunchKernelGGL fills with the dispatch's own timestamps "
                                                  "(unclamped)" if event_kernel_s else "wall ms_per_step (no events)"),
                         "event_kernel_us": (kernel_ms / launches * 1e3) if launches else None,
                         "event_exceeds_wall_step": bool(event_kernel_s and event_kernel_s > wall_step_s * 1.02),
                         "algorithmic_bytes_per_launch": b_env * E,
                         "valu": valu_view(E, avg_kernel_s, args.workload),
                         "kernel_resources": kernel_resources_view(scenario, fast, N)},
            "terminated_in_last_step": int(term),
            "ix_spawn_counters": (lambda c: dict(c, drop_rate=c["ix_spawns_dropped"] / max(1, c["ix_spawns"] + c["ix_spawns_dropped"])))(eng.counters()) if scenario == "intersection" else None,
            "host_path_env_steps_per_s": host_rate,
        }
        if world == 1 and not args.no_frontend:
            line["frontend"] = frontend_view(args.workload, cfg_dict, E, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, cfg_dict, fast, scenario)
        if world == 1 and args.workload == "fast" and not args.no_secondary and not tuning:
            line["secondary_workloads"] = secondary_workloads()
        emit(line)
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    return outs


if __name__ == "__main__":
    main()
