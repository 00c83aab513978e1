#!/bin/bash
# r05 call 2: intersection kernel -- partner slots asked per trip of the pair loops (HWY_IX_PAIR_TRIPS) x rows per trip of the straight
# walk (HWY_IX_WALK_ROWS): tests on the tree's build, then the variants interleaved on config 4 (2048 x 30 slots, OccupancyGrid).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c2; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_ix_parity.py tests/test_ix_device_traffic.py tests/test_occupancy_grid.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
B=$R/tools/ablate/_build
for rep in 1 2; do
  for v in r04 cur ix_p1w2 ix_p4w2 ix_p4w6 ix_p8w3 ix_p2w3; do
    if [ $v = cur ]; then unset HWY_ENGINE_LIB; else export HWY_ENGINE_LIB=$B/libhwy_engine_$v.so; fi
    timeout 150 python bench.py --workload intersection --envs-per-gpu 2048 --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/ix_${v}_$rep.json 2>> $O/err.txt
  done
done
unset HWY_ENGINE_LIB
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c2")
for f in sorted(glob.glob(O + "/ix_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get("rollout_k16") or {}
        print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}  K16 {k.get('ms_per_step', 0) * 1e3:8.2f}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
tail -5 $O/err.txt
