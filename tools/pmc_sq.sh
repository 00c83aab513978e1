#!/bin/bash
# Developer tool (GPU box): SQ instruction / cycle counters of the step kernel of a bench workload, three rocprofv3 --pmc
# passes -> gpurun_out/pmc_sq_<workload>.json (mean over the last 30 launches, per wave == per env-step).
#   bash tools/pmc_sq.sh [workload] [envs] [kernel-name-substring]
W=${1:-fast}; E=${2:-4096}; K=${3:-hwy_step_wave}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc_$W
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES SQ_INSTS_BRANCH"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE"
k=0
for P in "$P1" "$P2" "$P3"; do k=$((k+1))
timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc_$W/p$k -o run -- python $R/bench.py --workload $W --envs-per-gpu $E --no-cpu-baseline --no-secondary --settle-ms 0 --steps 30 --warmup 40 --repeats 1 > /dev/null 2> $R/gpurun_out/pmc_$W/p$k.err
done
cd $R
python - "$W" "$E" "$K" <<'PY'
import csv, glob, json, sys, collections
W, E, K = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(list)
for d in sorted(glob.glob(f"gpurun_out/pmc_{W}/p*/")):
    for f in glob.glob(d + "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if K in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v[-30:]) / len(v[-30:]) / E for k, v in acc.items()}
sys.path.insert(0, ".")
from highwayenv_amd import build
json.dump({"workload": W, "envs": E, "kernel": K, "kernel_source_sha16": build.kernel_source_hash(),
           "waves_per_simd": min(4.0, E / 1024.0), "per_wave_per_step": out}, open(f"gpurun_out/pmc_sq_{W}.json", "w"), indent=1)
print(json.dumps(out))
PY
