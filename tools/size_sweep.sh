# Developer tool (GPU box): headline workload at several batch sizes + fixed vs per-frame kernel cost
for e in 1024 2048 3072 4096 6144 8192 16384; do
  python bench.py --envs-per-gpu $e --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('envs=$e', round(d['ms_per_step']*1000,2),'us', round(d['value']/1e6,2),'M', 'kernel', round(d['roofline']['avg_kernel_us'],2))"
done
python tools/phase_timing.py 4096
python tools/phase_timing.py 8192
