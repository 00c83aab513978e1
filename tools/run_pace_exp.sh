# GPU box: headline bench + wave timeline + section clocks of the current build (developer experiment script)
O=gpurun_out/${1:-exp}; mkdir -p $O
for k in 1 2; do timeout 120 python bench.py --no-cpu-baseline --steps 500 --repeats 5 > $O/bench_$k.json 2>$O/bench_$k.err; python -c "
import json
d=json.load(open('$O/bench_$k.json')); print('bench', round(d['ms_per_step']*1e3,2), [round(x*1e3,2) for x in d['ms_per_step_repeats']], 'kernel', round(d['roofline']['avg_kernel_us'],2))"; done
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so timeout 120 python tools/wave_timeline2.py 4096 60 > $O/timeline.txt 2>&1; head -22 $O/timeline.txt | cut -c1-220
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wticks.so timeout 120 python tools/section_cycles.py > $O/sections.txt 2>&1; cat $O/sections.txt
