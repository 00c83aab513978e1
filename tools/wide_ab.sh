#!/bin/bash
# Developer tool (GPU box): config 3 (v0_n100) on the two-vehicles-per-thread kernel, optionally against the workgroup kernel,
# plus its section clocks:   bash tools/wide_ab.sh [block]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/wide_ab; mkdir -p $O; cd $R
for e in 1024 2048; do
  timeout 100 python bench.py --workload v0_n100 --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/wide_$e.json 2>> $O/err.txt
  [ "$1" = block ] && timeout 100 python bench.py --workload v0_n100 --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 --tune block_kernel=1 > $O/block_$e.json 2>> $O/err.txt
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "wide_ab")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d["ms_per_step"] * 1e3, 2), "us; K=16:", d.get("rollout_k16"))
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
[ -f tools/ablate/_build/libhwy_engine_w2ticks.so ] && HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_w2ticks.so timeout 120 python tools/wide_section_cycles.py 1024 2>&1 | tail -12
