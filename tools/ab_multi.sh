#!/bin/bash
# Same-box comparison of several builds of the library: bash tools/ab_multi.sh "libA.so libB.so ..." [bench_kernels args]
LIBS=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  for L in $LIBS; do echo "== $L rep $rep"; ADH_LIB_PATH=$L python3 $R/tools/bench_kernels.py "$@" 2>/dev/null; done
done
