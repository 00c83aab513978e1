mkdir -p gpurun_out/s6
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/s6/pytest.log 2>&1; tail -3 gpurun_out/s6/pytest.log
timeout 300 python bench.py --workload merge_ma4 > gpurun_out/s6/bench_merge_ma4.json 2> gpurun_out/s6/bench_merge_ma4.err
timeout 300 python bench.py --workload merge --no-cpu-baseline > gpurun_out/s6/bench_merge.json 2> gpurun_out/s6/bench_merge.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s6/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --workload merge_ma4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/s6/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/s6/prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/s6/prof -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
cat gpurun_out/s6/bench_merge_ma4.json
