#!/bin/bash
O=gpurun_out/r2i; mkdir -p $O
run() { local name=$1 w=$2 e=$3; shift 3
  local t=""; for kv in "$@"; do t="$t --tune $kv"; done
  timeout 200 python bench.py --workload $w --envs-per-gpu $e --no-cpu-baseline --steps 300 --warmup 40 --repeats 3 $t > $O/$name.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/$name.json')); print('$name', '$*', 'us/step', round(d['ms_per_step']*1e3,1), [round(x*1e3,1) for x in d['ms_per_step_repeats']])"
}
run fast_default fast 4096
run v0_default v0 4096
run v0_w4 v0 4096 waves_per_eu=4
run v0_w4_off v0 4096 waves_per_eu=4 prio_shift=-1
run m4_default merge_ma4 4096
run merge_default merge 4096
run merge_off merge 4096 prio_shift=-1
run fast_8192 fast 8192
run fast_8192_14 fast 8192 prio_shift=14
run fast_2048 fast 2048
run fast_2048_off fast 2048 prio_shift=-1
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
