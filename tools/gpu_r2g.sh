#!/bin/bash
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log; grep -a "drop rate\|SELFTEST\|agreement\|first-collision" $O/pytest.log | head -20
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
HWY_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --repeats 3 > $O/bench_forcedist.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_forcedist.json')); print('force-dist', d['ms_per_step'], d['gather_every_1'])"
timeout 300 python bench.py --workload intersection --envs-per-gpu 2048 --no-cpu-baseline --steps 300 --repeats 3 > $O/bench_ix.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_ix.json')); print('ix', d['ms_per_step'], d['ix_spawn_counters'])"
timeout 300 python bench.py --workload merge_ma4 --no-cpu-baseline --steps 300 --repeats 3 > $O/bench_merge.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_merge.json')); print('merge_ma4', d['ms_per_step'])"
timeout 300 python bench.py --workload v0_n100 --envs-per-gpu 1024 --no-cpu-baseline --steps 300 --repeats 3 > $O/bench_cfg3.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print('cfg3 shard', d['ms_per_step'])"
tail -5 $O/bench.err
