# Round-end measurements on the GPU box (run via gpurun from the repo root).  Outputs under gpurun_out/r04_*; copy into profiles/.
#   part 1: bash tools/final_profiles.sh bench      bench line (headline fp32 + split_bf16x3 object + cpu_baseline + parity)
#   part 2: bash tools/final_profiles.sh stats      rocprofv3 --kernel-trace --stats of the same command, both contractions
#   part 3: bash tools/final_profiles.sh pmc        HBM traffic passes of the headline bench + MFMA-busy passes of the bf16 x 3 kernels
set -e
R=$GRAFT_REPO_ROOT
case "$1" in
bench)
  python bench.py > gpurun_out/r04_bench_n1.log 2>&1; tail -1 gpurun_out/r04_bench_n1.log > gpurun_out/r04_bench_n1.json
  python tools/wino_err.py > gpurun_out/r04_bf16x3_error.txt 2>&1
  ;;
stats)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-split > $R/gpurun_out/r04_stats.log 2>&1
  export ADH_CONTRACT=bf16x3
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_stats_bf16x3 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-split --no-forward-eval > $R/gpurun_out/r04_stats_bf16x3.log 2>&1
  ;;
pmc)
  bash tools/pmc_bench.sh
  cp gpurun_out/pmc_bench.json gpurun_out/r04_pmc_bench.json
  cd /tmp && export TMPDIR=/tmp
  export ADH_CONTRACT=bf16x3
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_b3_$i -- python3 $R/tools/bench_kernels.py --only conv --pass fwd --iters 3 > $R/gpurun_out/pmc_b3_$i.log 2>&1 || echo "pass $i failed"
  done
  cd $R
  python3 tools/pmc_summary.py gpurun_out/pmc_b3_1 gpurun_out/pmc_b3_2 gpurun_out/pmc_b3_3 gpurun_out/pmc_b3_4 > gpurun_out/r04_pmc_bf16x3_raw.txt 2>&1 || true
  ;;
esac
echo done
