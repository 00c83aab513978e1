#!/bin/bash
# r05 call 1: issue-cost microbenchmark; the rank-space abort chain in the one-wavefront kernel + the ds_or exchange in the wide
# kernel: their GPU tests, then headline / highway-v0 / config-3 shard interleaved against the round-4 library.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c1; mkdir -p $O
cd $R
timeout 120 tools/microbench/issue_bench > $O/issue_costs.json 2> $O/issue_err.txt; echo "issue_bench rc=$?"
timeout 500 python -m pytest tests/test_wide_kernel.py tests/test_engine_parity.py tests/test_rollout.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
L4=$R/tools/ablate/_build/libhwy_engine_r04.so
for rep in 1 2; do
  for spec in fast:4096 v0:4096 v0_n100:1024 merge_ma4:4096; do
    w=${spec%%:*}; e=${spec##*:}
    for lib in r04 cur; do
      if [ $lib = r04 ]; then export HWY_ENGINE_LIB=$L4; else unset HWY_ENGINE_LIB; fi
      timeout 150 python bench.py --workload $w --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/${w}_${lib}_$rep.json 2>> $O/err.txt
    done
  done
done
unset HWY_ENGINE_LIB
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c1")
for f in sorted(glob.glob(O + "/*_*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("rollout_k16") or {}
        print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}  K16 {k.get('ms_per_step', 0) * 1e3:8.2f}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
tail -5 $O/err.txt
head -c 1500 $O/issue_costs.json
