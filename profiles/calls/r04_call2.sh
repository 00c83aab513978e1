#!/bin/bash
# Round 4, GPU call 2: placement probe (hwy_set_block_order with perfect-foresight / predicted / shuffled orders), the GPU suite
# and the fuzz on the -ffp-contract=on + compiler-materialised-constants build, one bench line per workload.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r04_call2; mkdir -p $out
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so timeout 300 python tools/placement_probe.py 4096 80 > $out/placement_probe.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/gpu_suite.txt 2>&1
HWY_FUZZ_CHUNKS=40 timeout 900 python -m pytest tests/test_fuzz_configs.py -m gpu -q -s -p no:cacheprovider > $out/gpu_fuzz.txt 2>&1
for spec in "fast 4096" "v0 4096" "v0_n100 1024" "merge_ma4 4096" "intersection 2048"; do
  set -- $spec
  timeout 300 python bench.py --workload $1 --envs-per-gpu $2 --steps 300 --repeats 3 --no-cpu-baseline --rollout-k 0 > $out/bench_$1.json 2>> $out/bench.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_fast_driver_shape.json 2>> $out/bench.err
cat $out/placement_probe.txt; tail -n 5 $out/gpu_suite.txt; grep -c "intersection fuzz chunk" $out/gpu_fuzz.txt; tail -n 3 $out/gpu_fuzz.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_call2/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"]*1e3,2), "us wall", round(d["ms_per_step_device"]*1e3,2), "us device", round(d["roofline"]["avg_kernel_us"],2), "us kernel")
    except Exception as ex: print(f, "ERR", ex)
PY
