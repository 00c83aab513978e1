#!/bin/bash
# r05 call 5: one HBM round trip at the start of the one-wavefront kernel (done flag, clock, actions, state requested together) +
# the kernel-argument segment touched line by line up front: headline and highway-v0 against the build before.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c7; mkdir -p $O
cd $R


B=$R/tools/ablate/_build
for rep in 1 2 3; do
  for spec in fast:4096 v0:4096; do
    w=${spec%%:*}; e=${spec##*:}
    for v in pre epi0 cur; do
      if [ $v = cur ]; then unset HWY_ENGINE_LIB; else export HWY_ENGINE_LIB=$B/libhwy_engine_$v.so; fi
      timeout 150 python bench.py --workload $w --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/${w}_${v}_$rep.json 2>> $O/err.txt
    done
  done
done
unset HWY_ENGINE_LIB
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c7")
for f in sorted(glob.glob(O + "/*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("rollout_k16") or {}
        print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}  K16 {k.get('ms_per_step', 0) * 1e3:8.2f}  kernel {d['roofline']['avg_kernel_us']:.2f}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
tail -3 $O/err.txt
