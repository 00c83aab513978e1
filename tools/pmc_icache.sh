#!/bin/bash
# Developer tool (GPU box): instruction-cache counters of the step kernel of a bench workload (one rocprofv3 --pmc pass).
#   bash tools/pmc_icache.sh [workload] [envs] [kernel-name-substring] [extra bench flags...]
W=${1:-fast}; E=${2:-4096}; K=${3:-hwy_step_wave}; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc_ic_$W
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_ic_$W/p -o run -- python $R/bench.py --workload $W --envs-per-gpu $E --no-cpu-baseline --no-secondary --settle-ms 0 --steps 30 --warmup 40 --repeats 1 "$@" > /dev/null 2> $R/gpurun_out/pmc_ic_$W/p.err
cd $R
python - "$W" "$E" "$K" <<'PY'
import csv, glob, sys, collections
W, E, K = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmc_ic_{W}/p/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v[-30:]) / len(v[-30:]) / E, 1) for k, v in acc.items()})
PY
