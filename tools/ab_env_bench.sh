#!/bin/bash
# Same-box A/B of the headline step under an environment switch: bash tools/ab_env_bench.sh VAR "0 1" [bench.py args]
V=$1; VALS=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2 3; do
  for x in $VALS; do
    echo "== $V=$x rep $rep"
    env $V=$x python3 $R/bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-forward-eval "$@" 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print(round(d["value"],3),"images/s",round(d["ms_per_step"],2),"ms/step  wino43 frac",round(r["frac"],4),"avg launch ms",round(r["avg_launch_ms"],4))'
  done
done
