#!/bin/bash
O=gpurun_out/r2b; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['ms_per_step_repeats'], d['roofline']['avg_kernel_us'], d['value'])"
timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 30 --repeats 1 > $O/bench300.json 2>> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench300.json')); print('300:', d['ms_per_step'], d['roofline']['avg_kernel_us'])"
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so timeout 120 python tools/wave_timeline2.py 4096 60 > $O/timeline4096.txt 2>&1
cat $O/timeline4096.txt
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so timeout 120 python tools/wave_timeline2.py 4096 400 > $O/timeline4096_400.txt 2>&1
head -12 $O/timeline4096_400.txt
