#!/bin/bash
# BatchNorm pass removal, kernel level: the F(4x4,3x3) data-gradient launch with the producing layer's
# BN-backward sums in its epilogue (adh_conv_wino43_dgrad_bnred) against the plain launch and the bn_bwd_reduce pass it replaces.
# (profiles/r03_ab_bnred_kernel.txt is this comparison from the throw-away -DW4_FUSE_BNRED=1 build that preceded the product one.)
R=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  echo "== rep $rep"
  python3 $R/tools/bench_kernels.py --only conv --pass bnred --iters 10 2>/dev/null
done
