#!/bin/bash
# r05 call 4: three / four vehicles per thread (N <= 192 / 256) on the MI355X: tests, then N = 201 against the workgroup kernel;
# the full-size parity tests with ALL environments through the oracle; the intersection fuzz with the end-of-step products.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c4; mkdir -p $O
cd $R
python -c "from oracle import oracle; oracle.build(force=True)"
timeout 600 python -m pytest tests/test_wide_kernel.py -m gpu -q -x -p no:cacheprovider > $O/pytest_wide.log 2>&1; echo "wide rc=$?"; tail -2 $O/pytest_wide.log
HWY_FUZZ_CHUNKS=24 timeout 900 python -m pytest tests/test_fuzz_configs.py -m gpu -q -x -s -k "wide or intersection" -p no:cacheprovider > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
( time timeout 1200 python -m pytest tests/test_full_size_properties.py -m gpu -q -x -s -p no:cacheprovider ) > $O/full.log 2>&1; echo "full rc=$?"; grep -a "full size\|at full size\|passed\|failed\|real" $O/full.log
B=$R/tools/ablate/_build
for rep in 1 2; do
  for e in 512 1024 2048; do
    timeout 150 python bench.py --workload v0_n200 --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 200 --repeats 3 > $O/n200_wide_${e}_$rep.json 2>> $O/err.txt
    timeout 150 python bench.py --workload v0_n200 --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --tune block_kernel=1 > $O/n200_block_${e}_$rep.json 2>> $O/err.txt
  done
  for v in cur net_ws1; do
    if [ $v = cur ]; then unset HWY_ENGINE_LIB; else export HWY_ENGINE_LIB=$B/libhwy_engine_$v.so; fi
    timeout 150 python bench.py --workload merge_ma4 --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/merge_${v}_$rep.json 2>> $O/err.txt
  done
  unset HWY_ENGINE_LIB
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c4")
for f in sorted(glob.glob(O + "/*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("rollout_k16") or {}
        print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}  K16 {k.get('ms_per_step', 0) * 1e3:8.2f}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
tail -3 $O/err.txt
