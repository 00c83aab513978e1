#!/bin/bash
# r04 call 7: the two-vehicles-per-thread one-wavefront kernel (hwy_wave2.h) on the MI355X: its GPU tests, then config 3 (1024 and
# 2048 x 101) interleaved against the workgroup kernel it replaces there (--tune block_kernel=1).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c7; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_wide_kernel.py -m gpu -q -s -p no:cacheprovider > $O/pytest_wide.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_wide.log
for rep in 1 2; do
  for e in 1024 2048; do
    timeout 100 python bench.py --workload v0_n100 --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/wide_${e}_$rep.json 2>> $O/err.txt
    timeout 100 python bench.py --workload v0_n100 --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 --tune block_kernel=1 > $O/block_${e}_$rep.json 2>> $O/err.txt
  done
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04c7")
for f in sorted(glob.glob(O + "/*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"] * 1e3, "us", d.get("ms_per_step_device"), d.get("rollout_k16", {}).get("ms_per_step") if isinstance(d.get("rollout_k16"), dict) else d.get("rollout_k16"))
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
tail -5 $O/err.txt
