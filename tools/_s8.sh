mkdir -p gpurun_out/s8
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --workload intersection --envs-per-gpu 2048 > gpurun_out/s8/bench_ix.json 2> gpurun_out/s8/bench_ix.err
tail -3 gpurun_out/s8/bench_ix.err
cat gpurun_out/s8/bench_ix.json
timeout 300 python bench.py --workload intersection --envs-per-gpu 4096 --no-cpu-baseline > gpurun_out/s8/bench_ix_4096.json 2>> gpurun_out/s8/bench_ix.err
python -c "
import json
d=json.loads(open('gpurun_out/s8/bench_ix_4096.json').read().strip().splitlines()[-1]); print('4096:', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s8/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --workload intersection --envs-per-gpu 2048 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/s8/prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/s8/prof -name "*kernel_stats.csv" | head -1 | xargs head -4 | cut -c1-200
