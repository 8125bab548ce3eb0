#!/bin/bash
# Phase ablation of conv_wgrad32v2_kernel: one library per compile-time H2_DBG value
# (1 = no transform, 4 = no contraction, 8 = stage only the first strip; sums of these).
#   here:        bash tools/ablate_wgrad32v2.sh build        (tools/micro/ablate/libwg32_d*.so -- not tracked)
#   on the box:  bash tools/ablate_wgrad32v2.sh run
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/adam-dehaze_amd/csrc
O=$R/tools/micro/ablate
DS="${DS:-1 4 8 9 12 13}"
if [ "$1" = build ]; then
  mkdir -p $O
  make -C $C > /dev/null
  for d in $DS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DH2_DBG=$d $2 -c $C/conv_wgrad32.hip -o /tmp/wg32_d$d.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libwg32_d$d.so $(ls $C/*.o | grep -v conv_wgrad32.o) /tmp/wg32_d$d.o
  done
else
  for case in down192to384 up384to96 up384to192; do
    echo "H2_DBG=0"; python3 $R/tools/bench_kernels.py --only $case --pass wgrad --iters 5 2>/dev/null
    for d in $DS; do
      echo "H2_DBG=$d"; ADH_LIB_PATH=$O/libwg32_d$d.so python3 $R/tools/bench_kernels.py --only $case --pass wgrad --iters 5 2>/dev/null
    done
  done
fi
