#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out/factor_ab4.txt"; : > "$O"
cd "$R"
run() { echo -n "$1  " >> "$O"; env $2 python tools/factor_time.py $3 4 2>&1 | grep factor >> "$O"; }
for n in 8192 4096 16384; do
run "stage16 group2     " "X=1" $n
run "stage8  group2     " "LPVS_RU_STAGE=8" $n
run "stage8  group3     " "LPVS_RU_STAGE=8 LPVS_FACTOR_GROUP=3" $n
run "stage8  group4     " "LPVS_RU_STAGE=8 LPVS_FACTOR_GROUP=4" $n
run "stage16 group4     " "LPVS_FACTOR_GROUP=4" $n
run "stage8  group2 r16 " "LPVS_RU_STAGE=8 LPVS_RESERVE_CUS=16" $n
done
LPVS_RU_STAGE=8 python tools/factor_check.py 2176 4096 8192 >> "$O" 2>&1
cat "$O"
