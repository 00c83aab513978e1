mkdir -p gpurun_out/h1
for h in 1 0; do for w in 2 3; do for e in 2048 4096; do
  timeout 120 python bench.py --tune ix_no_helpers=$((1-h)) --tune waves_per_eu=$w --workload intersection --envs-per-gpu $e --steps 60 --warmup 40 --no-cpu-baseline > gpurun_out/h1/b_${h}_${w}_${e}.json 2> gpurun_out/h1/b_${h}_${w}_${e}.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/h1/b_${h}_${w}_${e}.json").read().strip().splitlines()[-1]); print("helpers=$h wpe=$w envs=$e", round(d["ms_per_step"]*1000,1),"us", round(d["value"]/1e6,3),"M")
except Exception as ex: print("helpers=$h wpe=$w envs=$e FAIL", ex)
PY
done; done; done
timeout 300 python -m pytest tests/test_ix_parity.py tests/test_ix_device_traffic.py tests/test_full_size_properties.py -m gpu -x -q 2>&1 | tail -3
