#!/bin/bash
# Same-box A/B of two builds of the library (clocks differ by several per cent from box to box, so numbers from
# different gpurun calls do not compare): alternates tools/bench_kernels.py between lib A and lib B.
#   bash tools/ab_kernels.sh A.so B.so [bench_kernels args...]
A=$1; B=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  echo "== A ($A) rep $rep"; ADH_LIB_PATH=$A python3 $R/tools/bench_kernels.py "$@" 2>/dev/null
  echo "== B ($B) rep $rep"; ADH_LIB_PATH=$B python3 $R/tools/bench_kernels.py "$@" 2>/dev/null
done
