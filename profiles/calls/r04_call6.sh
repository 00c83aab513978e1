#!/bin/bash
# Round 4, GPU call 6 (after the profile call): issue-priority turn length re-swept on the faster kernels, the GPU suite with the
# MTV-axis tie rule, 500 intersection fuzz chunks, the default bench line with this build's counters in place.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r04_call6; mkdir -p $out
one() {
  timeout 300 python bench.py --steps 300 --repeats 3 --no-cpu-baseline --no-secondary --rollout-k 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-70s %8.2f us' % (' '.join(sys.argv[1:]), d['ms_per_step']*1e3))" "$@"
}
{
for rep in 1 2; do
  one --workload fast; for s in 12 13 15 16; do one --workload fast --tune prio_shift=$s; done
done
one --workload merge_ma4; for s in 15 17; do one --workload merge_ma4 --tune prio_shift=$s; done
one --workload v0; for s in 15 17; do one --workload v0 --tune prio_shift=$s; done
one --workload v0_n100 --envs-per-gpu 1024; one --workload v0_n100 --envs-per-gpu 1024 --tune prio_shift=-1; one --workload v0_n100 --envs-per-gpu 1024 --tune prio_shift=15
one --workload intersection --envs-per-gpu 2048; one --workload intersection --envs-per-gpu 2048 --tune prio_shift=16
} > $out/prio_sweep.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $out/gpu_suite.txt 2>&1
HWY_FUZZ_CHUNKS=500 timeout 1500 python -m pytest tests/test_fuzz_configs.py -m gpu -q -s -p no:cacheprovider -k intersection > $out/gpu_fuzz_ix.txt 2>&1
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
cat $out/prio_sweep.txt; tail -n 3 $out/gpu_suite.txt; tail -n 2 $out/gpu_fuzz_ix.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_call6/bench_default.json"))
print(round(d["ms_per_step"]*1e3,2), "us;", round(d["value"]/1e6,2), "M; frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"]["traffic"], "valu_issue", (d["roofline"]["valu"] or {}).get("valu_issue"))
PY
