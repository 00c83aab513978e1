#!/bin/bash
# Round 4, GPU call 4: GPU suite on the build, intersection A/B (reciprocal radius against IEEE divisions), the full default bench line.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r04_call4; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $out/gpu_suite.txt 2>&1
one() {
  local L=$1; shift
  if [ "$L" != - ]; then export HWY_ENGINE_LIB=$L; else unset HWY_ENGINE_LIB; fi
  timeout 300 python bench.py --steps 300 --repeats 3 --no-cpu-baseline --no-secondary --rollout-k 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-36s %-50s %8.2f us' % ('$L', ' '.join(sys.argv[1:]), d['ms_per_step']*1e3))" "$@"
}
{ for rep in 1 2 3; do for L in - _ab/libhwy_engine_ixdiv.so; do one "$L" --workload intersection --envs-per-gpu 2048; one "$L" --workload intersection_kin --envs-per-gpu 2048; done; done; } > $out/ab_ix.txt 2>&1
unset HWY_ENGINE_LIB
( time timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time
tail -n 6 $out/gpu_suite.txt; grep "full size" $out/gpu_suite.txt; cat $out/ab_ix.txt; cat $out/bench_default.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_call4/bench_default.json"))
print(round(d["ms_per_step"]*1e3,2), "us;", round(d["value"]/1e6,2), "M env-steps/s; frac", round(d["roofline"]["frac"],4), "cpu", d["cpu_baseline"]["kind"], round(d["cpu_baseline"]["value"]), d["cpu_baseline"].get("all_threads",{}).get("value"))
for k,v in d["secondary_workloads"].items(): print(k, v.get("ms_per_step"), v.get("error"))
PY
