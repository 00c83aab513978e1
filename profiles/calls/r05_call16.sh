#!/bin/bash
# r05 call 16: the length of a turn at the issue port (tune_prio_shift) on the final build: headline 12..15, highway-v0 / config 5 15..17
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c16; mkdir -p $O
cd $R
for rep in 1 2; do
  for x in 0 12 13 14 15; do
    timeout 150 python bench.py --workload fast --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 300 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/fast_p${x}_$rep.json 2>> $O/err.txt
  done
  for x in 0 15 16 17; do
    timeout 150 python bench.py --workload v0 --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/v0_p${x}_$rep.json 2>> $O/err.txt
    timeout 150 python bench.py --workload merge_ma4 --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/merge_p${x}_$rep.json 2>> $O/err.txt
  done
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c16")
for f in sorted(glob.glob(O + "/*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us")
PY
