#!/bin/bash
# BatchNorm pass removal, measured: the F(4x4,3x3) data-gradient launch with the producing
# layer's BN-backward sums (sum g m, sum g m xhat) taken in its epilogue (-DW4_FUSE_BNRED=1: one more 16-byte load per output
# pixel quad, mask + two accumulations) against the plain launch plus the bn_bwd_reduce pass it would replace.
#   here:        bash tools/ab_bnred.sh build          (tools/micro/ablate/libw43_bnred.so -- not tracked)
#   on the box:  bash tools/ab_bnred.sh run
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/adam-dehaze_amd/csrc
O=$R/tools/micro/ablate
if [ "$1" = build ]; then
  mkdir -p $O
  make -C $C > /dev/null
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DW4_FUSE_BNRED=1 -c $C/conv_wino43.hip -o /tmp/w43_bnred.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libw43_bnred.so $(ls $C/*.o | grep -v conv_wino43.o) /tmp/w43_bnred.o
else
  for rep in 1 2; do
    echo "== product library (dgrad+bnred here = the ordinary scale/shift/residual/statistics epilogue), rep $rep"
    python3 $R/tools/bench_kernels.py --only conv --pass bnred --iters 10 2>/dev/null
    echo "== -DW4_FUSE_BNRED=1, rep $rep"
    ADH_LIB_PATH=$O/libw43_bnred.so python3 $R/tools/bench_kernels.py --only conv --pass bnred --iters 10 2>/dev/null
  done
fi
