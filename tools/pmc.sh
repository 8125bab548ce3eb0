#!/bin/bash
# usage: pmc.sh <tag> <bench args...>   (runs 3 counter passes, prints per-kernel sums)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/tools/bench_kernels.py "$@" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || exit 1
done
