# GPU box: A/B of two engine builds on ONE box, interleaved (developer script).  usage: bash tools/ab_bench.sh <libA or ""> <libB or ""> [bench args]
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in "$A" "$B"; do
    if [ -n "$L" ]; then export HWY_ENGINE_LIB=$L; else unset HWY_ENGINE_LIB; fi
    timeout 200 python bench.py --steps 500 --repeats 5 --no-cpu-baseline --rollout-k 0 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${L:-current}', round(d['ms_per_step']*1e3,2))"
  done
done
