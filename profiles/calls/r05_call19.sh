#!/bin/bash
# r05 call 19: who gets which priority in a turn (HWY_TURN_POLICY): 3-2-1-0 (the product) / one on top / two on top / one at the bottom
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c19; mkdir -p $O
cd $R
B=$R/tools/ablate/_build
for rep in 1 2; do
  for v in cur tp1 tp2 tp3; do
    if [ $v = cur ]; then unset HWY_ENGINE_LIB; else export HWY_ENGINE_LIB=$B/libhwy_engine_$v.so; fi
    timeout 150 python bench.py --workload fast --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 300 --repeats 3 --rollout-k 0 > $O/fast_${v}_$rep.json 2>> $O/err.txt
    timeout 150 python bench.py --workload v0 --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --rollout-k 0 > $O/v0_${v}_$rep.json 2>> $O/err.txt
  done
done
unset HWY_ENGINE_LIB
python - <<'PY'
import json, glob, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c19")
acc = collections.defaultdict(list)
for f in sorted(glob.glob(O + "/*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    acc[os.path.basename(f)[:-7]].append(round(d['ms_per_step'] * 1e3, 2))
for k in sorted(acc): print(f"{k:20s}", acc[k])
PY
