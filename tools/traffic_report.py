#!/usr/bin/env python3
"""Turn two rocprofv3 counter_collection.csv files (FETCH_SIZE pass, WRITE_SIZE pass) of
tools/traffic_probe.py into calibrated HBM bytes per step-kernel launch."""
import csv
import glob
import json
import sys


def per_launch(dirname, counter):
    f = sorted(glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True))[-1]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if "hwy_step" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return vals


fetch_dir, write_dir, known_json = sys.argv[1:4]
known = json.loads(open(known_json).read().strip().splitlines()[-1])
out = {}
for name, d, key in (("FETCH_SIZE", fetch_dir, "frames0_read_bytes"), ("WRITE_SIZE", write_dir, "frames0_write_bytes")):
    v = per_launch(d, name)
    cal, step = v[-40:-20], v[-20:]
    kb_cal, kb_step = sum(cal) / len(cal), sum(step) / len(step)
    factor = known[key] / (kb_cal * 1024)  # bytes really moved per reported byte, in this access pattern
    out[name] = {"frames0_reported_KB": kb_cal, "known_bytes": known[key], "calibration": factor,
                 "step_reported_KB": kb_step, "step_bytes_raw": kb_step * 1024, "step_bytes_calibrated": kb_step * 1024 * factor}
out["traffic_bytes_per_launch_calibrated"] = out["FETCH_SIZE"]["step_bytes_calibrated"] + out["WRITE_SIZE"]["step_bytes_calibrated"]
out["traffic_bytes_per_launch_raw"] = out["FETCH_SIZE"]["step_bytes_raw"] + out["WRITE_SIZE"]["step_bytes_raw"]
out["algorithmic_bytes_per_launch"] = known["algorithmic_bytes_per_step"]
out["envs"], out["workload"] = known["E"], "fast"
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from highwayenv_amd import build  # noqa: E402
out["kernel_source_sha16"] = build.kernel_source_hash()  # bench.py quotes these counters only for this kernel build
print(json.dumps(out, indent=1))
