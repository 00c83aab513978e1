#!/bin/bash
# The first 8-GPU lease in ONE command (nothing here has ever run on more than one MI355X: gpurun hands out one GPU per call):
#
#   bash tools/run_scale.sh [out_dir]           # on a node with 2 / 4 / 8 visible GPUs
#
#   1. the two-rank RCCL tests (tests/test_multi_gpu.py: ncclGather through hwy_comm_init / hwy_gather and through
#      torch.distributed, skipped on every box so far)
#   2. the headline, weak scaling: bench.py --gpus {1,2,4,8}, 4096 envs x 51 vehicles per GPU (BASELINE's metric)
#   3. BASELINE config 3, strong scaling: 8192 envs x 101 vehicles over {1,2,4,8} GPUs (--workload v0_n100 --envs-per-gpu 8192
#      --scaling strong; at 8 GPUs that is config 3's 1024 envs per GPU)
# and prints one line per run: n_gpus, the world size the process group itself reported, env-steps/s, ms per step.  bench.py
# re-launches itself under torch.distributed.run (127.0.0.1 rendezvous) for N > 1; no efficiency is computed here -- the
# driver derives it from the per-N values.
O=${1:-gpurun_out/scale}; mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $NG"
timeout 900 python -m pytest tests/test_multi_gpu.py -m gpu -q -rs 2>&1 | tail -8 | tee "$O/test_multi_gpu.txt"
for N in 1 2 4 8; do
  [ "$N" -le "$NG" ] || continue
  timeout 900 python bench.py --gpus $N --steps 500 --repeats 3 --no-cpu-baseline --no-secondary --no-frontend > "$O/fast_weak_$N.json" 2> "$O/fast_weak_$N.err"
  timeout 900 python bench.py --gpus $N --workload v0_n100 --envs-per-gpu 8192 --scaling strong --steps 200 --repeats 3 \
    --no-cpu-baseline --no-secondary --no-frontend > "$O/cfg3_strong_$N.json" 2> "$O/cfg3_strong_$N.err"
done
python - "$O" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*_[0-9].json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):24s} n_gpus {d['n_gpus']}  world_size_reported_by_the_process_group "
              f"{d['config']['world_size_reported_by_the_process_group']}  scaling {d['scaling']:6s}  {d['value'] / 1e6:9.2f} M env-steps/s  "
              f"{d['ms_per_step'] * 1e3:8.2f} us per step  gather: {d['config']['gather']}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable:", ex)
PY
