mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; tail -3 gpurun_out/final/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
timeout 300 python bench.py --workload intersection --envs-per-gpu 2048 > gpurun_out/final/bench_ix.json 2> gpurun_out/final/bench_ix.err
timeout 300 python bench.py --workload merge_ma4 > gpurun_out/final/bench_merge_ma4.json 2> gpurun_out/final/bench_merge_ma4.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/final/prof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof_ix -o run -- python $GRAFT_REPO_ROOT/bench.py --workload intersection --envs-per-gpu 2048 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/final/prof_ix.err
cd $GRAFT_REPO_ROOT
head -3 gpurun_out/final/prof/run_kernel_stats.csv | cut -c1-160
head -3 gpurun_out/final/prof_ix/run_kernel_stats.csv | cut -c1-160
python - <<'PY'
import json
for f in ["bench","bench_ix","bench_merge_ma4"]:
    d=json.loads(open(f"gpurun_out/final/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]/1e6,3), "M env-steps/s", round(d["ms_per_step"]*1e3,1), "us/step", round(d["roofline"]["avg_kernel_us"],1), (d.get("cpu_baseline") or {}).get("value"))
PY
