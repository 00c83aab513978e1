#!/bin/bash
# Developer tool (GPU box): one bench line per "workload:envs" argument, printed as us per step.
#   bash tools/quick_bench.sh v0:4096 merge_ma4:4096 intersection:2048 fast:4096 v0_n100:1024
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/quick; mkdir -p $O; cd $R
for spec in "$@"; do
  w=${spec%%:*}; e=${spec##*:}
  timeout 150 python bench.py --workload $w --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 $QUICK_FLAGS > $O/$w_$e.json 2>> $O/err.txt
  python - $O/$w_$e.json $w $e <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get("rollout_k16") or {}
    print(f"{sys.argv[2]:16s} {sys.argv[3]:>6s} envs  {d['ms_per_step'] * 1e3:8.2f} us/step   K=16: {k.get('ms_per_step', 0) * 1e3:8.2f}")
except Exception as ex:
    print(sys.argv[2], "unreadable:", ex)
PY
done
