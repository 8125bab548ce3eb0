#!/bin/bash
# Phase ablation of conv_wino43_kernel: builds the library once per compile-time W4_DBG value
# (1 = no input transform, 2 = no contraction, 4 = no epilogue, 8 = stage only the first chunk; a run-time switch
# would change the register allocation) and times the three headline layer shapes with each.
#   here:        bash tools/ablate_wino43.sh build        (writes tools/micro/ablate/lib43_d*.so -- not tracked, travels with gpurun)
#   on the box:  bash tools/ablate_wino43.sh run
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/adam-dehaze_amd/csrc
O=$R/tools/micro/ablate
if [ "$1" = build ]; then
  mkdir -p $O
  make -C $C > /dev/null
  for d in 1 2 4 6 7 8; do
    hipcc --offload-arch=gfx950 -O3 -fPIC -DW4_DBG=$d -c $C/conv_wino43.hip -o /tmp/w43_d$d.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib43_d$d.so $(ls $C/*.o | grep -v conv_wino43.o) /tmp/w43_d$d.o
  done
else
  echo "W4_DBG=0"; python3 $R/tools/bench_kernels.py --only conv --pass fwd --iters 5 2>/dev/null
  for d in 1 2 4 6 7 8; do
    echo "W4_DBG=$d"; ADH_LIB_PATH=$O/lib43_d$d.so python3 $R/tools/bench_kernels.py --only conv --pass fwd --iters 5 2>/dev/null
  done
fi
