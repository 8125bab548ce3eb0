#!/bin/bash
# Same-box A/B of tools/bench_kernels.py under an environment switch: bash tools/ab_env_kernels.sh VAR "0 1" [bench_kernels args]
V=$1; VALS=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  for x in $VALS; do
    echo "== $V=$x rep $rep"; env $V=$x python3 $R/tools/bench_kernels.py "$@" 2>/dev/null
  done
done
