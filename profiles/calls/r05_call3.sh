#!/bin/bash
# r05 call 3: road-network kernel with the rank carried in the packed word + two walk steps per trip, against the round-4 library
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c3; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_net_parity.py tests/test_net_reset.py tests/test_rollout.py tests/test_ix_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
HWY_FUZZ_CHUNKS=40 timeout 600 python -m pytest tests/test_fuzz_configs.py -m gpu -q -x -k merge -p no:cacheprovider > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
B=$R/tools/ablate/_build
for rep in 1 2; do
  for spec in merge_ma4:4096 merge:4096 intersection:2048; do
    w=${spec%%:*}; e=${spec##*:}
    for v in r04 cur; do
      if [ $v = cur ]; then unset HWY_ENGINE_LIB; else export HWY_ENGINE_LIB=$B/libhwy_engine_$v.so; fi
      timeout 150 python bench.py --workload $w --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/${w}_${v}_$rep.json 2>> $O/err.txt
    done
  done
done
unset HWY_ENGINE_LIB
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c3")
for f in sorted(glob.glob(O + "/*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("rollout_k16") or {}
        print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}  K16 {k.get('ms_per_step', 0) * 1e3:8.2f}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
tail -3 $O/err.txt
