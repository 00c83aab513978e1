mkdir -p gpurun_out/h2
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/h2/pytest.log 2>&1; tail -2 gpurun_out/h2/pytest.log
for e in 2048 4096 8192; do
  timeout 120 python bench.py --workload intersection --envs-per-gpu $e --steps 100 --warmup 40 --no-cpu-baseline > gpurun_out/h2/b_$e.json 2> gpurun_out/h2/b_$e.err
  python -c "
import json
d=json.loads(open('gpurun_out/h2/b_$e.json').read().strip().splitlines()[-1]); print('envs=$e', round(d['ms_per_step']*1000,1),'us', round(d['value']/1e6,3),'M')"
done
timeout 200 python bench.py --workload intersection --envs-per-gpu 2048 > gpurun_out/h2/bench_intersection.json 2> gpurun_out/h2/bench_intersection.err
timeout 200 python bench.py --workload intersection_kin --envs-per-gpu 2048 --no-cpu-baseline > gpurun_out/h2/bench_intersection_kin.json 2> gpurun_out/h2/bench_intersection_kin.err
cut -c1-200 gpurun_out/h2/bench_intersection.json gpurun_out/h2/bench_intersection_kin.json
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/h2/prof -- python $R/bench.py --workload intersection --envs-per-gpu 2048 --no-cpu-baseline > $R/gpurun_out/h2/prof.log 2>&1
find $R/gpurun_out/h2/prof -name "*kernel_stats.csv" | head -2
