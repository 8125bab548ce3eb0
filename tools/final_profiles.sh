# Round-end measurements on the GPU box (run via gpurun from the repo root): bench line, rocprofv3 kernel stats of the same
# command, PMC traffic passes, the other BASELINE.json configs.  Outputs under gpurun_out/r03_*; copy into profiles/.
set -e
python bench.py > gpurun_out/r03_bench_n1.log 2>&1; tail -1 gpurun_out/r03_bench_n1.log > gpurun_out/r03_bench_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_stats.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_bench.sh
cp gpurun_out/pmc_bench.json gpurun_out/r03_pmc_bench.json
echo done
