#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters, collected as the guide prescribes: separate --pmc
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only.  Run on the GPU box:
#   bash tools/collect_pmc.sh <outdir>
OUT=${1:-gpurun_out/pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$OUT/$C" -o bench -- \
      python3 "$R/bench.py" --steps 1 --warmup 0 --iters 20 --no-cpu-baseline --no-general-path --no-cfg4-strong --no-concurrent > "$R/$OUT/$C.log" 2>&1
  echo "$C rc=$?"
done
cd "$R"
python3 tools/pmc_summary.py "$OUT"
