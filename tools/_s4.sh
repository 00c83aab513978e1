mkdir -p gpurun_out/s4
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/s4/pytest.log 2>&1; tail -3 gpurun_out/s4/pytest.log
for w in 2 3 4; do
  HWY_STEP_WAVES_PER_EU=$w timeout 200 python bench.py --workload merge_ma4 --no-cpu-baseline > gpurun_out/s4/merge_ma4_w$w.json 2> gpurun_out/s4/merge_ma4_w$w.err
done
timeout 200 python bench.py --workload merge --no-cpu-baseline > gpurun_out/s4/merge.json 2> gpurun_out/s4/merge.err
timeout 300 python bench.py --workload merge_ma4 > gpurun_out/s4/merge_ma4.json 2> gpurun_out/s4/merge_ma4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s4/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['terminated_in_last_step'], d.get('cpu_baseline',{}).get('value'))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -5 gpurun_out/s4/*.err
