#!/bin/bash
# r05 call 17: turn lengths between the powers of two (tune_prio_shift >= 64: k x 64 ticks; 256 = 2^14, 512 = 2^15, 1024 = 2^16)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c17; mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_engine_parity.py tests/test_abi.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
for rep in 1 2; do
  for x in 14 192 224 256 288 320 352; do
    timeout 150 python bench.py --workload fast --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 300 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/fast_p${x}_$rep.json 2>> $O/err.txt
  done
  for x in 320 384 448 512 576 640 768; do
    timeout 150 python bench.py --workload v0 --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/v0_p${x}_$rep.json 2>> $O/err.txt
    timeout 150 python bench.py --workload merge_ma4 --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --rollout-k 0 --tune prio_shift=$x > $O/merge_p${x}_$rep.json 2>> $O/err.txt
  done
done
python - <<'PY'
import json, glob, os, collections
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c17")
acc = collections.defaultdict(list)
for f in sorted(glob.glob(O + "/*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    acc[os.path.basename(f)[:-7]].append(round(d['ms_per_step'] * 1e3, 2))
for k in sorted(acc, key=lambda s: (s.split('_p')[0], int(s.split('_p')[1]))):
    print(f"{k:20s}", acc[k])
PY
tail -2 $O/err.txt
