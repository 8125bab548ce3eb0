#!/bin/bash
# Phase ablation of conv_wgrad_wino43_kernel: one library per compile-time G4_DBG value
# (1 = no transform, 4 = no contraction, 8 = stage only the first strip; sums of these) and the three headline shapes.
#   here:        bash tools/ablate_wgrad43.sh build ["extra hipcc flags"]   (tools/micro/ablate/libwg43_d*.so -- not tracked)
#   on the box:  bash tools/ablate_wgrad43.sh run
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/adam-dehaze_amd/csrc
O=$R/tools/micro/ablate
DS="${DS:-1 4 5 8 9 12 13}"
if [ "$1" = build ]; then
  mkdir -p $O
  make -C $C > /dev/null
  for d in $DS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DG4_DBG=$d $2 -c $C/conv_wgrad43.hip -o /tmp/wg43_d$d.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libwg43_d$d.so $(ls $C/*.o | grep -v conv_wgrad43.o) /tmp/wg43_d$d.o
  done
else
  export ADH_WINO43_WGRAD=1
  echo "G4_DBG=0"; python3 $R/tools/bench_kernels.py --only conv --pass wgrad --iters 5 2>/dev/null
  for d in $DS; do
    echo "G4_DBG=$d"; ADH_LIB_PATH=$O/libwg43_d$d.so python3 $R/tools/bench_kernels.py --only conv --pass wgrad --iters 5 2>/dev/null
  done
fi
