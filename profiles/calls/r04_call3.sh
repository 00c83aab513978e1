#!/bin/bash
# Round 4, GPU call 3: the whole GPU suite on the build (new: ASan host build, block order, NaN guard, rollout refusal), what
# -ffp-contract=on costs against =fast and buys against =off on ONE box, the counted knife-edge cases of the intersection fuzz
# over 200 chunks (calibration of the per-chunk ceilings), the bench line in its default and in the driver's shape.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r04_call3; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/gpu_suite.txt 2>&1
one() {
  local L=$1; shift
  if [ "$L" != - ]; then export HWY_ENGINE_LIB=$L; else unset HWY_ENGINE_LIB; fi
  timeout 300 python bench.py --steps 300 --repeats 3 --no-cpu-baseline --rollout-k 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-36s %-40s %8.2f us' % ('$L', ' '.join(sys.argv[1:]), d['ms_per_step']*1e3))" "$@"
}
{ for rep in 1 2 3; do for L in - _ab/libhwy_engine_f_fast.so _ab/libhwy_engine_f_nofma.so; do one "$L" --workload fast; done; done; } > $out/ab_contract.txt 2>&1
unset HWY_ENGINE_LIB
HWY_FUZZ_CALIBRATE=1 HWY_FUZZ_CHUNKS=200 timeout 1500 python -m pytest tests/test_fuzz_configs.py -m gpu -q -s -p no:cacheprovider > $out/gpu_fuzz.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > $out/bench_fast.json 2>> $out/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_fast_driver_shape.json 2>> $out/bench.err
tail -n 6 $out/gpu_suite.txt; cat $out/ab_contract.txt; tail -n 3 $out/gpu_fuzz.txt
python - <<'PY'
import json,glob,re
for f in sorted(glob.glob("gpurun_out/r04_call3/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"]*1e3,2), "us wall", round(d["ms_per_step_device"]*1e3,2), "us device", round(d["roofline"]["avg_kernel_us"],2), "us kernel", "settle", d["settle_steps"], [round(x*1e3,2) for x in d["ms_per_step_repeats"]])
    except Exception as ex: print(f, "ERR", ex)
cnt={"edge":[], "touch":[], "lane":[], "flip":[], "cut":[], "frames":[]}
for l in open("gpurun_out/r04_call3/gpu_fuzz.txt"):
    m=re.search(r"tolerated and counted: (\d+) env-steps.*?, (\d+) pending-impact.*?, (\d+) frames with.*?, (\d+) lane-index flips and (\d+) queue", l)
    if m:
        for k,v in zip(("edge","touch","lane","flip","cut"), m.groups()): cnt[k].append(int(v))
        cnt["frames"].append(float(re.search(r"compared at 1e-9: \d+ \(([\d.]+) %", l).group(1)))
for k,v in cnt.items():
    if v: print(k, "chunks", len(v), "sum", sum(v), "max", max(v), "min", min(v))
PY
