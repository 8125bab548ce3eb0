set -e
python bench.py > gpurun_out/r02b_bench_n1.log 2>&1; tail -1 gpurun_out/r02b_bench_n1.log > gpurun_out/r02b_bench_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02b_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-forward-eval > $GRAFT_REPO_ROOT/gpurun_out/r02b_stats.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_bench.sh
echo done
