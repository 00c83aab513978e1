#!/bin/bash
# r05 call 13: the headline batch as S sub-batches on S streams (bench --split-batch S), S = 2, 4, 8
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05c13; mkdir -p $O
cd $R
for S in 2 4 8; do
  timeout 200 python bench.py --no-cpu-baseline --no-secondary --repeats 3 --steps 500 --split-batch $S > $O/split_$S.json 2>> $O/err.txt
done
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05c13")
for S in (2, 4, 8):
    d = json.loads([l for l in open(f"{O}/split_{S}.json") if l.startswith("{")][-1])
    s = d[f"split_batch_s{S}"]
    print(S, round(d["ms_per_step"] * 1e3, 2), "split", round(s["ms_per_step"] * 1e3, 2), round(s["value"] / 1e6, 1), [round(x * 1e3, 2) for x in s["ms_per_step_repeats"]])
PY
