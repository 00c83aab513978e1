#!/bin/bash
# r05 call 6: where the kernel arguments live (HIP_FORCE_DEV_KERNARG) -- headline, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c6; mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in unset 0 1; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    timeout 150 python bench.py --workload fast --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/fast_dk${v}_$rep.json 2>> $O/err.txt
  done
done
unset HIP_FORCE_DEV_KERNARG
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c6")
for f in sorted(glob.glob(O + "/*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("rollout_k16") or {}
        print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}  K16 {k.get('ms_per_step', 0) * 1e3:8.2f}  kernel {d['roofline']['avg_kernel_us']:.2f}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
