timeout 300 python -m pytest tests/test_net_parity.py tests/test_net_reset.py tests/test_full_size_properties.py -m gpu -x -q 2>&1 | tail -3
for w in 2 3 4; do
HWY_STEP_WAVES_PER_EU=$w timeout 200 python bench.py --workload merge_ma4 --no-cpu-baseline --steps 200 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge_ma4 wpe=$w', round(d['value']/1e6,3), 'M env-steps/s', round(d['ms_per_step']*1e3), 'us/step')"
done
HWY_STEP_WAVES_PER_EU=3 timeout 200 python bench.py --workload merge --no-cpu-baseline --steps 200 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge wpe=3', round(d['value']/1e6,3), round(d['ms_per_step']*1e3))"
HWY_STEP_WAVES_PER_EU=4 timeout 200 python bench.py --workload merge --no-cpu-baseline --steps 200 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge wpe=4', round(d['value']/1e6,3), round(d['ms_per_step']*1e3))"
