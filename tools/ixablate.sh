# time the ix ablation variants (tools/ablate/make_variants.py ix*) on the intersection bench workload
mkdir -p gpurun_out/abl
for v in ixbase ixnoreg ixnocoll ixnoarc ixnostraight ixnomask ixnoobs ixnospawn ixnoact ixnointeg; do
  HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_$v.so timeout 120 python bench.py --workload ${1:-intersection} --envs-per-gpu 2048 --steps 60 --warmup 40 --no-cpu-baseline > gpurun_out/abl/$v.json 2> gpurun_out/abl/$v.err
  python -c "
import json
try:
    d=json.loads(open('gpurun_out/abl/$v.json').read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step']*1000,1),'us')
except Exception as ex: print('$v FAIL', ex)"
done
