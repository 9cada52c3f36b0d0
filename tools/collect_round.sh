#!/bin/bash
# Refresh the round's evidence under gpurun_out/prof (run on the GPU box from the repo root; copy what is to be judged into
# profiles/): bench lines for cfg3 / cfg4 / cfg3-f32, rocprofv3 kernel stats for cfg3 / cfg4 / cfg5, PMC traffic of the cfg3 bench.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out/prof"
cd "$R"
timeout -k 10 300 python bench.py > gpurun_out/prof/bench_cfg3.json 2> gpurun_out/prof/bench_cfg3.err
timeout -k 10 300 python bench.py --workload cfg4 > gpurun_out/prof/bench_cfg4.json 2> gpurun_out/prof/bench_cfg4.err
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline --no-general-path > gpurun_out/prof/bench_cfg3_f32.json 2> gpurun_out/prof/bench_cfg3_f32.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof/k3" -o bench -- python3 "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-general-path --no-alt-storage > "$R/gpurun_out/prof/k3.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof/k4" -o bench -- python3 "$R/bench.py" --workload cfg4 --steps 1 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/prof/k4.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof/k5" -o bench -- python3 "$R/tools/bench_cfg5.py" 20 8 500 > "$R/gpurun_out/prof/k5.log" 2>&1
cd "$R"
for k in k3 k4 k5; do F=$(find gpurun_out/prof/$k -name "*kernel_stats.csv" | head -1); python tools/trim_stats.py "$F" gpurun_out/prof/${k}_kernel_stats.csv > /dev/null; rm -rf gpurun_out/prof/$k; done
bash tools/collect_pmc.sh gpurun_out/prof/pmc > gpurun_out/prof/pmc.log 2>&1
rm -rf gpurun_out/prof/pmc/FETCH_SIZE gpurun_out/prof/pmc/WRITE_SIZE
ls -la gpurun_out/prof gpurun_out/prof/pmc
