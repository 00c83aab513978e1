#!/bin/bash
# Round 4, GPU call 1: the fused build (build.py FP_CONTRACT = "fast" + the two queued source patches) against the round-3
# arithmetic (f_nofma = the same sources with -ffp-contract=off; r03 = the round-3 library as shipped) and against the
# compiler-materialised SGPR constants (f_kcmix) on all five workloads on ONE box, then the whole GPU suite + fuzz on the build.
#   (build container)  python tools/ablate/make_variants.py f_nofma f_kcmix; cp tools/ablate/_build/libhwy_engine_f_{nofma,kcmix}.so _ab/
#   gpurun --timeout 1500 -- 'bash tools/r04_call1.sh'
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r04_call1; mkdir -p $out
LIBS="- _ab/libhwy_engine_r03.so _ab/libhwy_engine_f_nofma.so _ab/libhwy_engine_f_kcmix.so"
one() {
  local L=$1; shift
  if [ "$L" != - ]; then export HWY_ENGINE_LIB=$L; else unset HWY_ENGINE_LIB; fi
  timeout 300 python bench.py --steps 300 --repeats 3 --no-cpu-baseline --rollout-k 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-36s %-40s %8.2f us' % ('$L', ' '.join(sys.argv[1:]), d['ms_per_step']*1e3))" "$@"
}
{
for rep in 1 2 3; do for L in $LIBS; do one "$L" --workload fast; done; done
for rep in 1 2; do
  for L in $LIBS; do
    one "$L" --workload v0
    one "$L" --workload v0_n100 --envs-per-gpu 1024
    one "$L" --workload merge_ma4
    one "$L" --workload intersection --envs-per-gpu 2048
  done
done
} > $out/ab.txt 2>&1
unset HWY_ENGINE_LIB
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/gpu_suite.txt 2>&1
HWY_FUZZ_CHUNKS=60 timeout 600 python -m pytest tests/test_fuzz_configs.py -m gpu -q -p no:cacheprovider > $out/gpu_fuzz.txt 2>&1
tail -4 $out/gpu_suite.txt $out/gpu_fuzz.txt; cat $out/ab.txt
