#!/bin/bash
# The first GPU call of round 4 (queued at the end of round 3, profiles/r03_history.md): the fused-multiply-add build and the
# compiler-materialised SGPR constants, A/B on all five workloads on ONE box, and the whole GPU suite on the fused build.
#
#   bash tools/r04_first_call.sh prep      # build container: builds the variant libraries into _ab/ (ships with gpurun; *.so is git-ignored)
#   gpurun --timeout 1500 -- 'bash tools/r04_first_call.sh run'      # ~15 GPU-minutes
#
# Then: adopt what wins in highwayenv_amd/build.py (FP_CONTRACT) / hwy_math.h, apply tools/ablate/r04_*.patch, re-run the suite,
# re-profile with tools/run_profile_r03.sh (new kernel hash), rm -rf _ab.
set -u
cd "$(dirname "$0")/.."
VARIANTS="f_fma f_kcmix f_fma_kcmix"
if [ "${1:-}" = prep ]; then
  python tools/ablate/make_variants.py $VARIANTS || exit 1
  mkdir -p _ab && for v in $VARIANTS; do cp tools/ablate/_build/libhwy_engine_$v.so _ab/ || exit 1; done
  ls -la _ab; exit 0
fi
out=gpurun_out/r04_first; mkdir -p $out
one() {  # one() <lib or ""> <bench args...>: us per step of one short bench run
  local L=$1; shift
  if [ -n "$L" ]; then export HWY_ENGINE_LIB=$L; else unset HWY_ENGINE_LIB; fi
  timeout 300 python bench.py --steps 300 --repeats 3 --no-cpu-baseline --rollout-k 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-36s %-40s %8.2f us' % ('${L:-as built}', ' '.join(sys.argv[1:]), d['ms_per_step']*1e3))" "$@"
}
{
for rep in 1 2 3; do
  for L in "" $(for v in $VARIANTS; do echo _ab/libhwy_engine_$v.so; done); do
    one "$L" --workload fast
  done
done
for rep in 1 2; do
  for L in "" $(for v in $VARIANTS; do echo _ab/libhwy_engine_$v.so; done); do
    one "$L" --workload v0
    one "$L" --workload v0_n100 --envs-per-gpu 1024
    one "$L" --workload merge_ma4
    one "$L" --workload intersection --envs-per-gpu 2048
  done
done
} > $out/ab.txt 2>&1
HWY_ENGINE_LIB=_ab/libhwy_engine_f_fma.so timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/gpu_suite_f_fma.txt 2>&1
HWY_ENGINE_LIB=_ab/libhwy_engine_f_fma.so HWY_FUZZ_CHUNKS=60 timeout 600 python -m pytest tests/test_fuzz_configs.py -m gpu -q -p no:cacheprovider > $out/gpu_fuzz_f_fma.txt 2>&1
tail -3 $out/gpu_suite_f_fma.txt $out/gpu_fuzz_f_fma.txt; cat $out/ab.txt
