mkdir -p gpurun_out/s5
timeout 300 python -m pytest tests/test_net_parity.py tests/test_net_reset.py -m gpu -x -q 2>&1 | tail -3
for w in 2 3 4; do
  HWY_STEP_WAVES_PER_EU=$w timeout 200 python bench.py --workload merge_ma4 --no-cpu-baseline > gpurun_out/s5/merge_ma4_w$w.json 2> gpurun_out/s5/merge_ma4_w$w.err
done
timeout 200 python bench.py --workload merge --no-cpu-baseline > gpurun_out/s5/merge.json 2> gpurun_out/s5/merge.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s5/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['terminated_in_last_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
export HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_nticks.so; timeout 120 python tools/net_section_cycles.py merge_ma4
