cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/k5t -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 1 --warmup 0 --iters 200 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/k5t.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find gpurun_out/k5t -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py "$F" 8
