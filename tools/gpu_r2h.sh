#!/bin/bash
O=gpurun_out/r2h; mkdir -p $O
run() { # name workload envs tune...
  local name=$1 w=$2 e=$3; shift 3
  local t=""; for kv in "$@"; do t="$t --tune $kv"; done
  timeout 200 python bench.py --workload $w --envs-per-gpu $e --no-cpu-baseline --steps 200 --warmup 40 --repeats 3 $t > $O/$name.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/$name.json')); print('$name', '$*', 'us/step', round(d['ms_per_step']*1e3,1), [round(x*1e3,1) for x in d['ms_per_step_repeats']])"
}
run m4_off merge_ma4 4096 prio_shift=-1
run m4_14 merge_ma4 4096 prio_shift=14
run m4_16 merge_ma4 4096 prio_shift=16
run m4_17 merge_ma4 4096 prio_shift=17
run m3_off merge_ma4 4096 prio_shift=-1 waves_per_eu=3
run m3_16 merge_ma4 4096 prio_shift=16 waves_per_eu=3
run ix_off intersection 2048 prio_shift=-1
run ix_14 intersection 2048 prio_shift=14
run ix_16 intersection 2048 prio_shift=16
run ix_18 intersection 2048 prio_shift=18
run ixnh_16 intersection 2048 prio_shift=16 ix_no_helpers=1 waves_per_eu=3
run c3_off v0_n100 1024 prio_shift=-1
run v0_off v0 4096 prio_shift=-1
run v0_14 v0 4096 prio_shift=14
run v0_16 v0 4096 prio_shift=16
