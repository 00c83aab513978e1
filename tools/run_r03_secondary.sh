# GPU box: bench lines + SQ counters of the secondary workloads (developer script; raw output under gpurun_out/$1)
O=gpurun_out/${1:-r03sec}; mkdir -p $O
for spec in "cfg3 v0_n100 1024 hwy_step_kernel" "merge_ma4 merge_ma4 4096 hwy_net_step" "intersection intersection 2048 hwy_ix_step" "v0 v0 4096 hwy_step_wave"; do
  set -- $spec
  timeout 300 python bench.py --workload $2 --envs-per-gpu $3 --steps 300 --repeats 5 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  python -c "
import json
d=json.load(open('$O/bench_$1.json')); print('$1', round(d['ms_per_step']*1e3,1), 'us; rollout', round(d['rollout_k16']['ms_per_step']*1e3,1))"
  bash tools/pmc_sq.sh $2 $3 $4 > $O/pmc_$1.log 2>&1; cp gpurun_out/pmc_sq_$2.json $O/pmc_sq_$1.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/pmc_sq_$1.json'))['per_wave_per_step']; print({k: round(v) for k,v in d.items() if k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_WAVE_CYCLES','SQ_ACTIVE_INST_VALU','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_INSTS_SMEM','SQ_ACTIVE_INST_LDS','SQ_WAIT_INST_LDS')})"
done
