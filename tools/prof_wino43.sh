#!/bin/bash
# Per-workgroup timeline of conv_wino43_kernel / conv_wino32_kernel (s_memtime stamps, -DW4_PROF -DW3_PROF dev build): where
# a workgroup's time goes and how long a CU sits between two workgroups.
#   here:        bash tools/prof_wino43.sh build     (writes tools/micro/ablate/lib43_prof.so -- not tracked, travels with gpurun)
#   on the box:  bash tools/prof_wino43.sh run [--kernel w32 --case down|up] [--kernel wgw [--cin 384 --h 128 --w 256]] [--cin 384 --h 128 --w 256]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/adam-dehaze_amd/csrc
O=$R/tools/micro/ablate
if [ "$1" = build ]; then
  mkdir -p $O
  make -C $C > /dev/null
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DW4_PROF -c $C/conv_wino43.hip -o /tmp/w43_prof.o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DW3_PROF -c $C/conv_wino.hip -o /tmp/w32_prof.o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWR_PROF -c $C/conv_wgrad.hip -o /tmp/wr_prof.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib43_prof.so $(ls $C/*.o | grep -v "conv_wino43.o\|conv_wino.o\|conv_wgrad.o") /tmp/w43_prof.o /tmp/w32_prof.o /tmp/wr_prof.o
else
  ADH_LIB_PATH=$O/lib43_prof.so python3 $R/tools/prof_wino43.py "${@:2}"
fi
