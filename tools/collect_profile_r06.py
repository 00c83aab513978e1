#!/usr/bin/env python3
"""Build container: turn the raw output of tools/run_profile_r06.sh (gpurun_out/r06prof/) into the committed summaries
under profiles/ (r06_*): bench lines, rocprofv3 kernel stats, calibrated HBM traffic per launch, SQ counters, timelines."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06prof")
DST = os.path.join(ROOT, "profiles")
sys.path.insert(0, ROOT)
from highwayenv_amd import build  # noqa: E402

sha = os.environ.get("HWY_COLLECT_SHA") or build.kernel_source_hash()


def jl(path):
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError(path)


KERNEL_OF = {"fast": "hwy_step_wave_kernel", "merge_ma4": "hwy_net_step_kernel", "intersection": "hwy_ix_step_kernel",
             "v0": "hwy_step_wave_kernel", "cfg3": "hwy_step_wide_kernel"}
out_notes = {}
# bench lines
for name in ("fast", "merge_ma4", "intersection", "v0", "cfg3", "intersection_kin", "merge", "fast_1024", "fast_2048", "fast_8192",
             "fast_16384", "fast_forcedist", "fast_split2", "intersection_split2", "cfg3_2048", "cfg3_block_kernel",
             "cfg3_2048_block_kernel", "v0_n200", "v0_n200_wide4"):
    p = os.path.join(SRC, f"bench_{name}.json")
    if os.path.exists(p):
        d = jl(p)
        assert d["roofline"]["kernel_source_sha16"] == sha, f"{name}: bench ran another kernel build ({d['roofline']['kernel_source_sha16']} != {sha})"
        json.dump(d, open(os.path.join(DST, f"r06_bench_{name}.json" if name != "fast" else "r06_bench.json"), "w"), indent=1)
        out_notes[name] = (d["ms_per_step"], d["value"])
# kernel stats (rocprofv3 --kernel-trace --stats)
for name in KERNEL_OF:
    f = glob.glob(os.path.join(SRC, f"stats_{name}", "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(DST, f"r06_kernel_stats_{name}.csv" if name != "fast" else "r06_kernel_stats.csv"))
# headline traffic (calibrated on the kernel's own access pattern) and SQ counters
traffic = {}
tf = os.path.join(SRC, "traffic_fast.json")
if os.path.exists(tf) and os.path.getsize(tf):
    d = json.load(open(tf))
    assert d["kernel_source_sha16"] == sha
    traffic["fast"] = d
    cal_f, cal_w = d["FETCH_SIZE"]["calibration"], d["WRITE_SIZE"]["calibration"]
else:
    cal_f, cal_w = 1.726, 1.066  # round-1 calibration (profiles/traffic_r01.json)


def pmc_mean(name, counter, kernel):
    f = glob.glob(os.path.join(SRC, f"pmc_{name}_{counter}", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals[-40:]) / len(vals[-40:]) if vals else None


for name, envs, wl in (("merge_ma4", 4096, "merge_ma4"), ("intersection", 2048, "intersection"), ("v0", 4096, "v0"), ("cfg3", 1024, "v0_n100")):
    fk, wk = pmc_mean(name, "FETCH_SIZE", KERNEL_OF[name]), pmc_mean(name, "WRITE_SIZE", KERNEL_OF[name])
    if fk is None or wk is None:
        continue
    b = jl(os.path.join(SRC, f"bench_{name}.json"))
    traffic[wl] = {"kernel": b["roofline"]["kernel"], "envs": envs, "workload": wl, "kernel_source_sha16": sha,
                   "FETCH_SIZE_reported_KB": fk, "WRITE_SIZE_reported_KB": wk,
                   "traffic_bytes_per_launch_calibrated": fk * 1024 * cal_f + wk * 1024 * cal_w,
                   "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
                   "calibration": {"FETCH": cal_f, "WRITE": cal_w, "from": "the headline kernel's pure load/store launches (traffic_probe.py)"}}
    traffic[wl]["ratio_to_algorithmic"] = traffic[wl]["traffic_bytes_per_launch_calibrated"] / traffic[wl]["algorithmic_bytes_per_launch"]
if "fast" in traffic:
    traffic["fast"]["ratio_to_algorithmic"] = traffic["fast"]["traffic_bytes_per_launch_calibrated"] / traffic["fast"]["algorithmic_bytes_per_launch"]
traffic["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (unit KB); headline: calibrated on hwy_step_frames(0) launches "
                   "(known byte count, same access pattern); the other workloads: mean over the last 40 step-kernel launches of bench.py "
                   "--steps 40 --warmup 40, same calibration factors.  bench.py quotes an entry only while kernel_source_sha16 matches the build it runs.")
json.dump(traffic, open(os.path.join(DST, "traffic_r06.json"), "w"), indent=1)
sq_all = {}
for name, wl in (("fast", "fast"), ("merge_ma4", "merge_ma4"), ("intersection", "intersection"), ("v0", "v0"), ("cfg3", "v0_n100")):
    sq = os.path.join(SRC, f"pmc_sq_{name}.json")
    if os.path.exists(sq):
        d = json.load(open(sq))
        assert d["kernel_source_sha16"] == sha, f"SQ counters of {name} belong to another kernel build"
        sq_all[wl] = d
if sq_all:  # one file, keyed by bench.py's --workload (bench.py: valu_view)
    json.dump(sq_all, open(os.path.join(DST, "r06_pmc_sq.json"), "w"), indent=1)
for src_name, dst_name in (("sections_fast.txt", "r06_section_clocks.txt"), ("sections_merge_ma4.txt", "r06_section_clocks_merge_ma4.txt"),
                           ("sections_intersection.txt", "r06_section_clocks_intersection.txt"),
                           ("sections_cfg3.txt", "r06_section_clocks_cfg3.txt")):
    sec = os.path.join(SRC, src_name)
    if os.path.exists(sec):
        open(os.path.join(DST, dst_name), "w").writelines(line for line in open(sec) if "amdgpu.ids" not in line)
for t in ("timeline_default", "timeline_noturns"):
    p = os.path.join(SRC, t + ".txt")
    if os.path.exists(p):
        keep = [line for line in open(p) if not line.startswith("late SIMD")]
        open(os.path.join(DST, f"r06_wave_{t}.txt"), "w").writelines(keep)
log = os.path.join(SRC, "pytest.log")
if os.path.exists(log):
    tail = [line for line in open(log, errors="replace")][-3:]
    open(os.path.join(DST, "r06_gpu_pytest_tail.txt"), "w").writelines(tail)
print("kernel_source_sha16", sha)
for k, (ms, v) in out_notes.items():
    print(f"{k:18s} {ms * 1e3:9.1f} us/step  {v / 1e6:8.2f} M env-steps/s")
for k, v in traffic.items():
    if isinstance(v, dict):
        print(f"traffic {k:14s} {v['traffic_bytes_per_launch_calibrated'] / 1e6:7.2f} MB/launch = {v['ratio_to_algorithmic']:.2f} x algorithmic")
