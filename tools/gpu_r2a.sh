#!/bin/bash
# GPU box: tests + bench + per-section cycle distribution (round 2, first call)
O=gpurun_out/r2a; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-1500 $O/bench.json
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wticks.so timeout 120 python tools/section_dist.py 1024 > $O/dist1024.txt 2>&1
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wticks.so timeout 120 python tools/section_dist.py 4096 > $O/dist4096.txt 2>&1
cat $O/dist1024.txt
