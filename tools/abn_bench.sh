# GPU box: A/B/.. of several engine builds on ONE box, interleaved (developer script).
# usage: bash tools/abn_bench.sh "<lib1> <lib2> ..." [bench args]     ("-" = the in-tree build)
LIBS=$1; shift
for rep in 1 2 3; do
  for L in $LIBS; do
    if [ "$L" != "-" ]; then export HWY_ENGINE_LIB=$L; else unset HWY_ENGINE_LIB; fi
    timeout 200 python bench.py --steps 300 --repeats 5 --no-cpu-baseline --rollout-k 0 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L'.split('_engine_')[-1], round(d['ms_per_step']*1e3,2))"
  done
done
