#!/bin/bash
O=gpurun_out/r2c; mkdir -p $O
for sh in -1 8 10 12 14; do
  timeout 200 python bench.py --no-cpu-baseline --repeats 3 --tune prio_shift=$sh > $O/bench_prio_$sh.json 2>> $O/bench.err
  python -c "
import json; d=json.load(open('$O/bench_prio_$sh.json')); print('prio_shift', $sh, 'ms/step', round(d['ms_per_step']*1e3,2), 'us; event avg', round(d['roofline']['avg_kernel_us'],2))"
done
for sh in -1 10; do
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so timeout 120 python tools/wave_timeline2.py 4096 60 prio_shift=$sh > $O/timeline_prio_$sh.txt 2>&1
echo "--- timeline prio_shift=$sh"; cut -c1-260 $O/timeline_prio_$sh.txt | head -24
done
