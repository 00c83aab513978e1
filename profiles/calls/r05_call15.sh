#!/bin/bash
# r05 call 15: fewer resident workgroups per CU (tune_extra_lds), the rest dispatched as wavefronts retire: 16 / 15 / 14 / 12 per CU
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c15; mkdir -p $O
cd $R
for rep in 1 2; do
  for x in 0 2800 3600 5600; do
    timeout 150 python bench.py --workload fast --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 300 --repeats 3 --rollout-k 0 --tune extra_lds=$x > $O/fast_x${x}_$rep.json 2>> $O/err.txt
  done
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c15")
for f in sorted(glob.glob(O + "/*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}")
PY
tail -2 $O/err.txt
