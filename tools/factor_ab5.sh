#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out/factor_ab5.txt"; : > "$O"
cd "$R"
run() { echo -n "$1  " >> "$O"; env $2 python tools/factor_time.py $3 4 2>&1 | grep factor >> "$O"; }
for n in 8192; do
run "band64            " "X=1" $n
run "band64 st8        " "LPVS_RU_STAGE=8" $n
run "band64 st8 group3 " "LPVS_RU_STAGE=8 LPVS_FACTOR_GROUP=3" $n
run "band64 reserve 4  " "LPVS_RESERVE_CUS=4" $n
run "band64 reserve 12 " "LPVS_RESERVE_CUS=12" $n
run "band64 reserve 16 " "LPVS_RESERVE_CUS=16" $n
run "band64 shares     " "LPVS_PIVOT_ALONE=0" $n
done
cat "$O"
