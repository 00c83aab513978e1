mkdir -p gpurun_out/s7
timeout 400 python -m pytest tests/test_full_size_properties.py -m gpu -x -q 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s7/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --workload merge_ma4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/s7/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/s7/prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/s7/prof -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
