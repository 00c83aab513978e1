#!/bin/bash
# GPU box: everything behind profiles/r06_* in ONE call (tests, bench lines, rocprofv3 kernel stats, PMC traffic and SQ
# counters, wave timeline, secondary workloads).  Raw output under gpurun_out/r06prof/; tools/collect_profile_r06.py
# (build container) turns it into the committed summaries.  Build the instrumented variants first:
#   python tools/ablate/make_variants.py wtimeline wticks nticks ixticks w2ticks
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/run_profile_r06.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06prof; mkdir -p $O
cd $R
# HWY_PROFILE_SKIP_TESTS=1: measurements only (a source-comment change re-keys the kernel build: the counters are re-recorded, the
# GPU suite and the fuzz of the same object code are not repeated)
[ -z "$HWY_PROFILE_SKIP_TESTS" ] && { timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log; }
timeout 600 python bench.py > $O/bench_fast.json 2> $O/bench_fast.err; echo "bench rc=$?"
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_fast_driver_shape.json 2>> $O/bench_misc.err
prof() { # name workload envs [kernel-substr]
  local name=$1 w=$2 e=$3
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o run -- python $R/bench.py --workload $w --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --warmup 40 --repeats 3 > $O/stats_$name.json 2> $O/stats_$name.err
  cd $R
}
pmc_bench() { # name workload envs counter
  local name=$1 w=$2 e=$3 c=$4
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${name}_$c -o run -- python $R/bench.py --workload $w --envs-per-gpu $e --no-cpu-baseline --no-secondary --settle-ms 0 --steps 40 --warmup 40 --repeats 1 > /dev/null 2> $O/pmc_${name}_$c.err
  cd $R
}
prof fast fast 4096
# headline traffic: calibrated on pure load/store launches of the same kernel (tools/traffic_probe.py)
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/probe_$c -o run -- python $R/tools/traffic_probe.py > $O/probe_$c.known 2> $O/probe_$c.err
  cd $R
done
python tools/traffic_report.py $O/probe_FETCH_SIZE $O/probe_WRITE_SIZE $O/probe_FETCH_SIZE.known > $O/traffic_fast.json 2> $O/traffic_fast.err
bash tools/pmc_sq.sh fast 4096 hwy_step_wave > $O/pmc_sq_fast.log 2>&1; cp gpurun_out/pmc_sq_fast.json $O/ 2>/dev/null
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wticks.so timeout 120 python tools/section_cycles.py > $O/sections_fast.txt 2>&1
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so timeout 120 python tools/wave_timeline2.py 4096 60 > $O/timeline_default.txt 2>&1
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_wtimeline.so timeout 120 python tools/wave_timeline2.py 4096 60 prio_shift=-1 > $O/timeline_noturns.txt 2>&1
# secondary workloads: bench line (with the CPU leg), kernel stats, PMC traffic
for spec in "merge_ma4 merge_ma4 4096" "intersection intersection 2048" "v0 v0 4096" "cfg3 v0_n100 1024"; do
  set -- $spec
  timeout 300 python bench.py --workload $2 --envs-per-gpu $3 --steps 300 --repeats 5 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  prof $1 $2 $3
  pmc_bench $1 $2 $3 FETCH_SIZE
  pmc_bench $1 $2 $3 WRITE_SIZE
  K=hwy_step_wave; [ $1 = merge_ma4 ] && K=hwy_net_step; [ $1 = intersection ] && K=hwy_ix_step; [ $1 = cfg3 ] && K=hwy_step_wide
  bash tools/pmc_sq.sh $2 $3 $K > $O/pmc_sq_$1.log 2>&1; cp gpurun_out/pmc_sq_$2.json $O/pmc_sq_$1.json 2>/dev/null
done
timeout 120 python bench.py --workload intersection_kin --envs-per-gpu 2048 --no-cpu-baseline --steps 300 --repeats 3 > $O/bench_intersection_kin.json 2>> $O/bench_misc.err
timeout 120 python bench.py --workload merge --no-cpu-baseline --steps 300 --repeats 3 > $O/bench_merge.json 2>> $O/bench_misc.err
for e in 1024 2048 8192 16384; do
  timeout 120 python bench.py --envs-per-gpu $e --no-cpu-baseline --no-secondary --steps 300 --repeats 3 > $O/bench_fast_$e.json 2>> $O/bench_misc.err
done
HWY_BENCH_FORCE_DIST=1 timeout 200 python bench.py --no-cpu-baseline --no-secondary --repeats 3 > $O/bench_fast_forcedist.json 2>> $O/bench_misc.err
# the batch as two sub-batches on two streams (bench --split-batch), and the merge kernel's section clocks
timeout 200 python bench.py --no-cpu-baseline --no-secondary --repeats 3 --split-batch 2 > $O/bench_fast_split2.json 2>> $O/bench_misc.err
timeout 200 python bench.py --workload intersection --envs-per-gpu 2048 --no-cpu-baseline --steps 300 --repeats 3 --split-batch 2 > $O/bench_intersection_split2.json 2>> $O/bench_misc.err
timeout 200 python bench.py --workload v0_n100 --envs-per-gpu 2048 --no-cpu-baseline --steps 300 --repeats 3 > $O/bench_cfg3_2048.json 2>> $O/bench_misc.err
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_nticks.so timeout 200 python tools/net_section_cycles.py merge_ma4 > $O/sections_merge_ma4.txt 2>&1
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_ixticks.so timeout 200 python tools/ix_section_dist.py > $O/sections_intersection.txt 2>&1
HWY_ENGINE_LIB=tools/ablate/_build/libhwy_engine_w2ticks.so timeout 120 python tools/wide_section_cycles.py 1024 > $O/sections_cfg3.txt 2>&1
# beyond BASELINE's configurations: N = 201 on the workgroup kernel (the engine's choice there) and with four vehicles per thread
timeout 200 python bench.py --workload v0_n200 --envs-per-gpu 1024 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 > $O/bench_v0_n200.json 2>> $O/bench_misc.err
timeout 200 python bench.py --workload v0_n200 --envs-per-gpu 1024 --no-cpu-baseline --no-secondary --steps 200 --repeats 3 --tune block_kernel=2 > $O/bench_v0_n200_wide4.json 2>> $O/bench_misc.err
timeout 120 tools/microbench/issue_bench > $O/issue_costs.json 2>> $O/bench_misc.err
# config 3 on the workgroup kernel the wide kernel replaced there (same box, same run)
timeout 200 python bench.py --workload v0_n100 --envs-per-gpu 1024 --no-cpu-baseline --steps 300 --repeats 3 --tune block_kernel=1 > $O/bench_cfg3_block_kernel.json 2>> $O/bench_misc.err
timeout 200 python bench.py --workload v0_n100 --envs-per-gpu 2048 --no-cpu-baseline --steps 300 --repeats 3 --tune block_kernel=1 > $O/bench_cfg3_2048_block_kernel.json 2>> $O/bench_misc.err
# the fuzz last (the longest single item): chunks HWY_FUZZ_FIRST .. + HWY_FUZZ_CHUNKS of every family on the final library
[ -z "$HWY_PROFILE_SKIP_TESTS" ] && { HWY_FUZZ_CHUNKS=${HWY_FUZZ_CHUNKS:-300} HWY_FUZZ_FIRST=${HWY_FUZZ_FIRST:-0} timeout 2400 python -m pytest tests/test_fuzz_configs.py -m gpu -q -s -p no:cacheprovider > $O/gpu_fuzz.txt 2>&1; tail -2 $O/gpu_fuzz.txt; }
# keep the merged output small: only the stats / counter CSVs (gpurun merges at most 64 MiB back; tools/pmc_sq.sh leaves its raw passes in gpurun_out/pmc_*)
rm -rf $R/gpurun_out/pmc_* $R/gpurun_out/pmc_sq_*.json
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
