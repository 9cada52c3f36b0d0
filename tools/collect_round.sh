#!/bin/bash
# Refresh the round's evidence under gpurun_out/prof (run on the GPU box from the repo root; copy what is to be judged into
# profiles/ with tools/publish_profiles.sh).  Three parts, each within one gpurun call:
#   bash tools/collect_round.sh bench    bench lines of every workload (cfg3 with its sub-records, cfg4, cfg2, cfg5, cfg3-f32)
#   bash tools/collect_round.sh stats    rocprofv3 --kernel-trace --stats of cfg3 / cfg4 / cfg5 / cfg2, the iteration timeline
#   bash tools/collect_round.sh pmc      FETCH_SIZE / WRITE_SIZE of every workload, matrix-pipe counters of the factorisation
#   bash tools/collect_round.sh gram     the dense MFMA Gram (general-w path; judge's row N1): kernel stats of a bench run WITH gram_general_path, SQ counters of gram_kernel
set -x
set -e   # a step that fails or is killed at its limit ends the call: no further GPU step behind it
PART=${1:-bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out/prof"
cd "$R"
if [ "$PART" = bench ]; then
  timeout -k 10 400 python bench.py > gpurun_out/prof/bench_cfg3.json 2> gpurun_out/prof/bench_cfg3.err
  timeout -k 10 300 python bench.py --workload cfg4 > gpurun_out/prof/bench_cfg4.json 2> gpurun_out/prof/bench_cfg4.err
  timeout -k 10 300 python bench.py --workload cfg2 > gpurun_out/prof/bench_cfg2.json 2> gpurun_out/prof/bench_cfg2.err
  timeout -k 10 400 python bench.py --workload cfg5 > gpurun_out/prof/bench_cfg5.json 2> gpurun_out/prof/bench_cfg5.err
  timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline --no-general-path --no-cfg4-strong --no-single-process > gpurun_out/prof/bench_cfg3_f32.json 2> gpurun_out/prof/bench_cfg3_f32.err
elif [ "$PART" = stats ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof/k3" -o bench -- python3 "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-general-path --no-alt-storage --no-cfg4-strong --no-baseline-configs --no-concurrent --no-single-process > "$R/gpurun_out/prof/k3.log" 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof/k4" -o bench -- python3 "$R/bench.py" --workload cfg4 --steps 1 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/prof/k4.log" 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof/k5" -o bench -- python3 "$R/bench.py" --workload cfg5 --steps 1 --warmup 0 --iters 500 --no-cpu-baseline > "$R/gpurun_out/prof/k5.log" 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof/k2" -o bench -- python3 "$R/bench.py" --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline --no-concurrent > "$R/gpurun_out/prof/k2.log" 2>&1
  cd "$R"
  for k in k3 k4 k5 k2; do F=$(find gpurun_out/prof/$k -name "*kernel_stats.csv" | head -1); python tools/trim_stats.py "$F" gpurun_out/prof/${k}_kernel_stats.csv > /dev/null; rm -rf gpurun_out/prof/$k; done
  timeout -k 10 120 python tools/iter_timeline.py > gpurun_out/prof/iteration_timeline.txt 2> gpurun_out/prof/iteration_timeline.err
elif [ "$PART" = gram ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof/kg" -o bench -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-alt-storage --no-cfg4-strong --no-baseline-configs --no-concurrent --no-single-process > "$R/gpurun_out/prof/kg.log" 2>&1
  cd "$R"
  F=$(find gpurun_out/prof/kg -name "*kernel_stats.csv" | head -1); python tools/trim_stats.py "$F" gpurun_out/prof/kg_kernel_stats.csv > /dev/null; rm -rf gpurun_out/prof/kg
  bash tools/collect_mfma_pmc.sh gpurun_out/prof/pmc_gram > gpurun_out/prof/pmc_gram.log 2>&1
else
  bash tools/collect_pmc.sh gpurun_out/prof/pmc cfg3 > gpurun_out/prof/pmc.log 2>&1
  for w in cfg2 cfg4 cfg5; do bash tools/collect_pmc.sh gpurun_out/prof/pmc_$w $w > gpurun_out/prof/pmc_$w.log 2>&1; done
  bash tools/collect_factor_pmc.sh gpurun_out/prof/pmc_factor 8192 > gpurun_out/prof/pmc_factor.log 2>&1
fi
ls -la gpurun_out/prof
