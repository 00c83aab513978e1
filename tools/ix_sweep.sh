#!/bin/bash
# GPU box: intersection workload, launch time against batch size (optionally with tuning knobs: TUNES="a=1 b=2")
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ixsweep; mkdir -p $O; cd $R
for e in ${ENVS:-512 1024 1536 2048 3072 4096}; do
  for t in "" $TUNES; do
    n=${t:-default}; arg=""; [ -n "$t" ] && arg="--tune $t"
    timeout 120 python bench.py --workload ${WORKLOAD:-intersection} --envs-per-gpu $e --no-cpu-baseline --steps 200 --warmup 60 --repeats 3 $arg > $O/ix_${e}_$n.json 2>> $O/err.log
    python - <<PY
import json
try:
    d = json.load(open("$O/ix_${e}_$n.json")); print("$e", "$n", round(d["ms_per_step"] * 1e3, 1), "us", round(d["value"] / 1e6, 2), "M/s")
except Exception as ex:
    print("$e $n failed", ex)
PY
  done
done
