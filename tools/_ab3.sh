run() { # workload envs tune...
  w=$1; e=$2; shift 2
  timeout 200 python bench.py --workload $w --envs-per-gpu $e --steps 300 --repeats 5 --no-cpu-baseline --rollout-k 0 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w $e $*', round(d['ms_per_step']*1e3,2))"
}
run v0_n100 1024
for ps in 14 15 16 17 18; do run v0_n100 1024 --tune prio_shift=$ps; done
run v0_n100 2048
run v0_n100 2048 --tune prio_shift=16
run fast 4096
