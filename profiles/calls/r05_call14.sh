#!/bin/bash
# r05 call 14: the checker's body to every lane through LDS instead of ten v_readlane (HWY_CHECKER_VIA_LDS)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c14; mkdir -p $O
cd $R
B=$R/tools/ablate/_build
for rep in 1 2 3 4; do
  for v in chklds cur; do
    if [ $v = cur ]; then unset HWY_ENGINE_LIB; else export HWY_ENGINE_LIB=$B/libhwy_engine_$v.so; fi
    timeout 150 python bench.py --workload fast --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/fast_${v}_$rep.json 2>> $O/err.txt
  done
done
unset HWY_ENGINE_LIB
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c14")
for f in sorted(glob.glob(O + "/*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}  kernel {d['roofline']['avg_kernel_us']:.2f}")
PY
