#!/bin/bash
O=gpurun_out/r2d; mkdir -p $O
run() { # name lib shift
  HWY_ENGINE_LIB=$2 timeout 200 python bench.py --no-cpu-baseline --repeats 3 --tune prio_shift=$3 > $O/b_$1_$3.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/b_$1_$3.json')); print('$1 prio_shift', $3, 'us/step', round(d['ms_per_step']*1e3,2), [round(x*1e3,2) for x in d['ms_per_step_repeats']], 'event avg', round(d['roofline']['avg_kernel_us'],2))"
}
BASE=highwayenv_amd/csrc/libhwy_engine.so
WR=tools/ablate/_build/libhwy_engine_wreload.so
for sh in -1 13 14 15 16; do run base $BASE $sh; done
for sh in -1 14 15; do run wreload $WR $sh; done
timeout 300 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
