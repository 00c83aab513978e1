#!/bin/bash
# r05 call 11: the ego's SAT by the whole wavefront (pair_collide_coop) against the build before
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c11; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_engine_parity.py tests/test_collision_steps.py tests/test_pileup.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
HWY_FUZZ_CHUNKS=60 timeout 600 python -m pytest tests/test_fuzz_configs.py -m gpu -q -x -k "test_random_configurations" -p no:cacheprovider > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
B=$R/tools/ablate/_build
for rep in 1 2 3 4; do
  for v in pre2 cur; do
    if [ $v = cur ]; then unset HWY_ENGINE_LIB; else export HWY_ENGINE_LIB=$B/libhwy_engine_$v.so; fi
    timeout 150 python bench.py --workload fast --envs-per-gpu 4096 --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/fast_${v}_$rep.json 2>> $O/err.txt
  done
done
unset HWY_ENGINE_LIB
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c11")
for f in sorted(glob.glob(O + "/*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):32s} {d['ms_per_step'] * 1e3:8.2f} us  dev {d.get('ms_per_step_device', 0) * 1e3:8.2f}  kernel {d['roofline']['avg_kernel_us']:.2f}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex)
PY
