# GPU box: the runs behind profiles/r01_bench_intersection*.json and r01_kernel_stats_intersection.csv (+ the gpu test suite)
mkdir -p gpurun_out/h3
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/h3/pytest.log 2>&1; tail -2 gpurun_out/h3/pytest.log
timeout 120 python bench.py --workload intersection --envs-per-gpu 2048 > gpurun_out/h3/bench_intersection.json 2> gpurun_out/h3/bench_intersection.err
timeout 60 python bench.py --workload intersection_kin --envs-per-gpu 2048 --no-cpu-baseline > gpurun_out/h3/bench_intersection_kin.json 2> gpurun_out/h3/bench_intersection_kin.err
cut -c1-200 gpurun_out/h3/bench_intersection.json gpurun_out/h3/bench_intersection_kin.json
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/h3/prof -- python $R/bench.py --workload intersection --envs-per-gpu 2048 --no-cpu-baseline > $R/gpurun_out/h3/prof.log 2>&1
find $R/gpurun_out/h3/prof -name "*kernel_stats.csv" | head -2
