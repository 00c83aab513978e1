#!/bin/bash
O=gpurun_out/r2e; mkdir -p $O
for sh in -1 62 14; do
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so timeout 120 python tools/wave_timeline2.py 4096 60 prio_shift=$sh > $O/timeline_prio_$sh.txt 2>&1
echo "--- timeline prio_shift=$sh"; cut -c1-300 $O/timeline_prio_$sh.txt | grep -v "^late\|^{" 
done
