#!/bin/bash
export LPVS_EXPERIMENTS=1   # the schedule knobs below are experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
# The factorisation of an 8192 x 8192 SPD matrix (tools/factor_time.py: best / all of 5) under every schedule knob of linalg.hip, one
# process per setting (the knobs are read once per process).  Run on the GPU box:  bash tools/factor_knob_sweep.sh > gpurun_out/r05/factor_knob_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
run() { echo -n "$1: "; env $1 timeout -k 10 120 python tools/factor_time.py 8192 5 2>/dev/null | tail -1; }
run "LPVS_DEFAULT=1"
for g in 1 2 3 4; do run "LPVS_FACTOR_GROUP=$g"; done
for s in 8 16; do run "LPVS_RU_STAGE=$s"; done
for c in 0 4 8 12 16 24; do run "LPVS_RESERVE_CUS=$c"; done
run "LPVS_LOOKAHEAD=0"; run "LPVS_LOOKAHEAD=1"
run "LPVS_PIVOT_ALONE=0"; run "LPVS_PIVOT_ALONE=1"
run "LPVS_CHAIN=split"; run "LPVS_PIVOT=regs"
run "LPVS_BAND_TILE=64"; run "LPVS_BAND_TILE=128"
run "LPVS_FACTOR_SCHEME=steps"
run "LPVS_DEFAULT=2"
