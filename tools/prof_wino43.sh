#!/bin/bash
# Per-workgroup timeline of conv_wino43_kernel (s_memtime stamps, -DW4_PROF dev build): where a workgroup's 70 us go and
# how long a CU sits between two workgroups.
#   here:        bash tools/prof_wino43.sh build     (writes tools/micro/ablate/lib43_prof.so -- not tracked, travels with gpurun)
#   on the box:  bash tools/prof_wino43.sh run
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/adam-dehaze_amd/csrc
O=$R/tools/micro/ablate
if [ "$1" = build ]; then
  mkdir -p $O
  make -C $C > /dev/null
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DW4_PROF -c $C/conv_wino43.hip -o /tmp/w43_prof.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib43_prof.so $(ls $C/*.o | grep -v conv_wino43.o) /tmp/w43_prof.o
else
  ADH_LIB_PATH=$O/lib43_prof.so python3 $R/tools/prof_wino43.py "${@:2}"
fi
