#!/bin/bash
# r05 call 8: the batched prologue (one HBM round trip, argument segment touched up front) in the wide, road-network and intersection
# kernels, against the build before (`pre`); tests of the touched kernels first.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c8; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_wide_kernel.py tests/test_net_parity.py tests/test_net_reset.py tests/test_ix_parity.py tests/test_ix_device_traffic.py tests/test_rollout.py tests/test_engine_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
B=$R/tools/ablate/_build
for rep in 1 2; do
  for spec in fast:4096 v0_n100:1024 merge_ma4:4096 intersection:2048; do
    w=${spec%%:*}; e=${spec##*:}
    for v in pre cur; do
      if [ $v = cur ]; then unset HWY_ENGINE_LIB; else export HWY_ENGINE_LIB=$B/libhwy_engine_$v.so; fi
      timeout 150 python bench.py --workload $w --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/${w}_${v}_$rep.json 2>> $O/err.txt
    done
  done
done
unset HWY_ENGINE_LIB
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c8")
for f in sorted(glob.glob(O + "/*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("rollout_k16") or {}
        print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}  K16 {k.get('ms_per_step', 0) * 1e3:8.2f}  kernel {d['roofline']['avg_kernel_us']:.2f}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
tail -3 $O/err.txt
