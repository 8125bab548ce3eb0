# Round-end measurements on the GPU box (run via gpurun from the repo root).  Outputs under gpurun_out/r04b_*; copy into profiles/.
#   part 1: bash tools/final_profiles.sh bench      bench line (headline fp32 + split_bf16x3 object + cpu_baseline + parity)
#   part 2: bash tools/final_profiles.sh stats      rocprofv3 --kernel-trace --stats of the same command, both contractions
#   part 3: bash tools/final_profiles.sh pmc        HBM traffic passes of the headline bench + MFMA-busy passes of the bf16 x 3 kernels
set -e
R=$GRAFT_REPO_ROOT
case "$1" in
bench)
  python bench.py > gpurun_out/r04b_bench_n1.log 2>&1; tail -1 gpurun_out/r04b_bench_n1.log > gpurun_out/r04b_bench_n1.json
  python tools/wino_err.py > gpurun_out/r04b_bf16x3_error.txt 2>&1
  ;;
stats)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04b_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-split > $R/gpurun_out/r04b_stats.log 2>&1
  export ADH_CONTRACT=bf16x3
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04b_stats_bf16x3 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-split --no-forward-eval > $R/gpurun_out/r04b_stats_bf16x3.log 2>&1
  ;;
pmc)
  bash tools/pmc_bench.sh
  cp gpurun_out/pmc_bench.json gpurun_out/r04b_pmc_bench.json
  ;;
pmc2)
  # SQ counters of the F(4x4,3x3) / F(3x3,2x2) forward kernels, fp32 MFMA and bf16 x 3 contraction, at the headline shapes
  cd /tmp && export TMPDIR=/tmp
  for c in fp32 bf16x3; do
    export ADH_CONTRACT=$c
    for sel in conv96 conv384 down96to192 up384to96; do
      i=0
      for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
        i=$((i+1))
        rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${c}_${sel}_$i -- python3 $R/tools/bench_kernels.py --only $sel --pass fwd --iters 3 > $R/gpurun_out/pmc_${c}_${sel}_$i.log 2>&1 || echo "pass $c $sel $i failed"
      done
      echo "=== $c $sel (tools/bench_kernels.py --only $sel --pass fwd --iters 3: 4 launches per pass)" >> $R/gpurun_out/r04b_pmc_contract_raw.txt
      python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${c}_${sel}_1 $R/gpurun_out/pmc_${c}_${sel}_2 $R/gpurun_out/pmc_${c}_${sel}_3 $R/gpurun_out/pmc_${c}_${sel}_4 | grep -v "pack_weights" >> $R/gpurun_out/r04b_pmc_contract_raw.txt 2>&1 || true
    done
  done
  cd $R
  ;;
esac
echo done
