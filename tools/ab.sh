#!/bin/bash
# GPU box: interleaved A/B/.. of engine builds x bench workloads on ONE box (developer script; replaces the per-call scripts of
# rounds 4-5).  Builds come from tools/build_variant.py / tools/build_rev.py (tools/ablate/_build/libhwy_engine_<name>.so);
# "-" is the in-tree library.
#
#   bash tools/ab.sh <tag> "<variants>" "<workloads>" [reps] [extra bench args ...]
#     variants : names separated by blanks, e.g. "- outline_sat pinned"; a name may carry tuning: "-:prio_shift=14"
#     workloads: fast | v0 | cfg3 | merge | ix | ixkin | n200 | fast<E> (fast with E envs), e.g. "fast cfg3"
#   results: gpurun_out/<tag>/<workload>_<variant>_<rep>.json and a table (median wall / device / kernel microseconds per variant)
TAG=$1; VARS=$2; WLS=$3; REPS=${4:-3}; shift 4 2>/dev/null || shift $#
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
B=$R/tools/ablate/_build
wl_args() {
  case $1 in
    fast) echo "--workload fast --envs-per-gpu 4096";;
    fast[0-9]*) echo "--workload fast --envs-per-gpu ${1#fast}";;
    cfg1) echo "--workload fast --envs-per-gpu 4096 --shape 20,3";;
    v0) echo "--workload v0 --envs-per-gpu 4096";;
    cfg3) echo "--workload v0_n100 --envs-per-gpu 1024";;
    n200) echo "--workload v0_n200 --envs-per-gpu 1024";;
    merge) echo "--workload merge_ma4";;
    mergev0) echo "--workload merge";;
    ix) echo "--workload intersection --envs-per-gpu 2048";;
    ixkin) echo "--workload intersection_kin --envs-per-gpu 2048";;
    *) echo "--workload $1";;
  esac
}
for rep in $(seq 1 $REPS); do
  for W in $WLS; do
    for V in $VARS; do
      name=${V%%:*}; tune=""
      if [ "$V" != "$name" ]; then for t in $(echo ${V#*:} | tr ',' ' '); do tune="$tune --tune $t"; done; fi
      if [ "$name" = "-" ]; then unset HWY_ENGINE_LIB; else export HWY_ENGINE_LIB=$B/libhwy_engine_$name.so; fi
      timeout 300 python bench.py $(wl_args $W) --no-cpu-baseline --no-secondary --no-frontend --rollout-k 0 --steps ${STEPS:-300} --repeats 3 $tune "$@" \
        > "$O/${W}_$(echo $V | tr ':=,' '___')_$rep.json" 2>> $O/err.txt
    done
  done
done
unset HWY_ENGINE_LIB
python - "$O" <<'PY'
import json, glob, os, sys, collections, statistics
O = sys.argv[1]
rows = collections.defaultdict(list)
turns = {}
for f in sorted(glob.glob(O + "/*_[0-9]*.json")):
    key = os.path.basename(f).rsplit("_", 1)[0]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows[key].append((d["ms_per_step"] * 1e3, d.get("ms_per_step_device", 0) * 1e3, d["roofline"]["avg_kernel_us"]))
        turns[key] = d["config"].get("issue_priority_turn", {}).get("turn")
    except Exception as ex:
        rows[key].append(None)
for key, v in rows.items():
    ok = [x for x in v if x]
    if not ok:
        print(f"{key:44s} unreadable")
        continue
    med = [statistics.median(c) for c in zip(*ok)]
    print(f"{key:44s} wall {med[0]:8.2f}  device {med[1]:8.2f}  kernel {med[2]:8.2f} us   turn {turns.get(key)}  runs " + " ".join(f"{x[0]:.2f}" for x in ok))
PY
