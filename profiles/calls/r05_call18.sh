#!/bin/bash
# r05 call 18: turn length around 768 x 64 ticks for the 15-frame workloads
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c18; mkdir -p $O
cd $R
for rep in 1 2; do
  for x in 16 352 384 416 704 736 768 800 832 896 960; do
    timeout 150 python bench.py --workload v0 --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/v0_p${x}_$rep.json 2>> $O/err.txt
  done
  for x in 16 704 768 832 896 960; do
    timeout 150 python bench.py --workload merge_ma4 --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/merge_p${x}_$rep.json 2>> $O/err.txt
    timeout 150 python bench.py --workload merge --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/mergev0_p${x}_$rep.json 2>> $O/err.txt
  done
done
python - <<'PY'
import json, glob, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c18")
acc = collections.defaultdict(list)
for f in sorted(glob.glob(O + "/*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    acc[os.path.basename(f)[:-7]].append(round(d['ms_per_step'] * 1e3, 2))
for k in sorted(acc, key=lambda s: (s.split('_p')[0], int(s.split('_p')[1]))):
    print(f"{k:20s}", acc[k])
PY
