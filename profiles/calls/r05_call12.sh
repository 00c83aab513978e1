#!/bin/bash
# r05 call 12: the bench lines once more, now that profiles/ holds the counters of this very build (bench.py quotes them)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05prof; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_fast.json 2> $O/bench_fast.err; echo "bench rc=$?"
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fast_driver_shape.json 2>> $O/bench_misc.err
for spec in "merge_ma4 merge_ma4 4096" "intersection intersection 2048" "v0 v0 4096" "cfg3 v0_n100 1024"; do
  set -- $spec
  timeout 300 python bench.py --workload $2 --envs-per-gpu $3 --steps 300 --repeats 5 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
done
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05prof")
for n in ("fast", "fast_driver_shape", "merge_ma4", "intersection", "v0", "cfg3"):
    d = json.loads([l for l in open(f"{O}/bench_{n}.json") if l.startswith("{")][-1])
    v = d["roofline"].get("valu") or {}
    print(n, round(d["ms_per_step"] * 1e3, 2), round(d["roofline"]["avg_kernel_us"], 2), "hbm", round(d["roofline"]["frac"], 4), "valu_issue", v.get("valu_issue"), "wait", v.get("wait_fraction_of_a_wavefront"), "traffic", d["roofline"].get("traffic"))
    if n == "fast":
        print({k: (x.get("ms_per_step"), x.get("valu_issue"), x.get("wait_fraction_of_a_wavefront")) for k, x in d.get("secondary_workloads", {}).items()})
PY
# how the host waits at the end of a region (the driver's 20-step shape carries the latency of one synchronize per region)
for v in unset 100 1000; do
  if [ $v = unset ]; then unset ROC_ACTIVE_WAIT_TIMEOUT; else export ROC_ACTIVE_WAIT_TIMEOUT=$v; fi
  for rep in 1 2; do
    timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/ds_wait${v}_$rep.json 2>> $O/bench_misc.err
  done
done
unset ROC_ACTIVE_WAIT_TIMEOUT
python - <<'PY'
import json, os, glob
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05prof")
for f in sorted(glob.glob(O + "/ds_wait*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["ms_per_step"] * 1e3, 2), round(d["ms_per_step_device"] * 1e3, 2), [round(x * 1e3, 2) for x in d["ms_per_step_repeats"]])
PY
# the intersection kernel with turns at the issue port (off by default there since round 3)
for x in -1 512 1024 2048 4096; do
  timeout 150 python bench.py --workload intersection --envs-per-gpu 2048 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/ix_p${x}.json 2>> $O/bench_misc.err
  python -c "
import json,sys
d=json.loads([l for l in open('$O/ix_p${x}.json') if l.startswith('{')][-1]); print('ix prio_shift $x', round(d['ms_per_step']*1e3,2))"
done
