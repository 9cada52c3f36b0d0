#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters, collected as the guide prescribes: separate --pmc
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only.  Run on the GPU box:
#   bash tools/collect_pmc.sh <outdir> [workload]        workload = cfg3 (default) | cfg2 | cfg4 | cfg5
# (the program itself follows `--`: python3 bench.py ..., nothing in between -- the profiler's preload has initialised the GPU)
OUT=${1:-gpurun_out/pmc}
WL=${2:-cfg3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
export TMPDIR=/tmp
cd /tmp
case $WL in
  cfg3) ARGS="--steps 1 --warmup 0 --iters 120 --no-cpu-baseline --no-general-path --no-cfg4-strong --no-baseline-configs --no-concurrent --no-single-process --no-alt-storage" ;;
  cfg2) ARGS="--workload cfg2 --steps 1 --warmup 0 --iters 40 --no-cpu-baseline --no-concurrent" ;;
  cfg4) ARGS="--workload cfg4 --steps 1 --warmup 0 --iters 70 --no-cpu-baseline" ;;
  cfg5) ARGS="--workload cfg5 --steps 1 --warmup 0 --iters 10 --no-cpu-baseline" ;;
  *) echo "unknown workload $WL"; exit 2 ;;
esac
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$OUT/$C" -o bench -- \
      python3 "$R/bench.py" $ARGS > "$R/$OUT/$C.log" 2>&1
  echo "$WL $C rc=$?"
done
cd "$R"
python3 tools/pmc_summary.py "$OUT" "$WL"
