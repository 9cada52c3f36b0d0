#!/bin/bash
# Copy the evidence tools/collect_round.sh left under gpurun_out/prof into profiles/ under this round's names.
#   bash tools/publish_profiles.sh r03
set -e
R=${1:?round tag, e.g. r03}
P=gpurun_out/prof
for w in cfg3 cfg4 cfg2 cfg5 cfg3_f32; do [ -s $P/bench_$w.json ] && cp $P/bench_$w.json profiles/${R}_bench_$w.json; done
[ -s $P/k3_kernel_stats.csv ] && cp $P/k3_kernel_stats.csv profiles/${R}_bench_cfg3_kernel_stats.csv
for k in 4 5 2; do [ -s $P/k${k}_kernel_stats.csv ] && cp $P/k${k}_kernel_stats.csv profiles/${R}_cfg${k}_kernel_stats.csv; done
[ -s $P/pmc/pmc_summary.json ] && cp $P/pmc/pmc_summary.json profiles/${R}_pmc_traffic.json
for w in cfg2 cfg4 cfg5; do [ -s $P/pmc_$w/pmc_summary.json ] && cp $P/pmc_$w/pmc_summary.json profiles/${R}_pmc_traffic_$w.json; done
[ -s $P/iteration_timeline.txt ] && cp $P/iteration_timeline.txt profiles/${R}_iteration_timeline.txt
[ -s $P/pmc_factor/factor_mfma_summary.json ] && cp $P/pmc_factor/factor_mfma_summary.json profiles/${R}_factor_mfma_pmc.json
[ -s $P/kg_kernel_stats.csv ] && cp $P/kg_kernel_stats.csv profiles/${R}_bench_cfg3_with_dense_gram_kernel_stats.csv
[ -s $P/pmc_gram/mfma_summary.json ] && cp $P/pmc_gram/mfma_summary.json profiles/${R}_gram_mfma_counters.json
ls -la profiles | grep "${R}_"
