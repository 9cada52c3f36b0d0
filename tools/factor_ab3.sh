#!/bin/bash
# group schedule: panels per pass, pivot kernel, CU reservation at n = 8192 (and 4096 / 12288 / 16384 for the default), then a timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out/factor_ab3.txt"; : > "$O"
cd "$R"
run() { echo -n "$1  " >> "$O"; env $2 python tools/factor_time.py $3 4 2>&1 | grep factor >> "$O"; }
run "default            " "X=1" 8192
run "group=3            " "LPVS_FACTOR_GROUP=3" 8192
run "group=2            " "LPVS_FACTOR_GROUP=2" 8192
run "group=1            " "LPVS_FACTOR_GROUP=1" 8192
run "pivot=regs         " "LPVS_PIVOT=regs" 8192
run "reserve=0          " "LPVS_RESERVE_CUS=0" 8192
run "reserve=4          " "LPVS_RESERVE_CUS=4" 8192
run "reserve=16         " "LPVS_RESERVE_CUS=16" 8192
run "steps (round 2)    " "LPVS_FACTOR_SCHEME=steps" 8192
for n in 2048 4096 12288 16384 32768; do
  run "default            " "X=1" $n
  run "default KW=128     " "LPVS_KW=128" $n
  run "steps (round 2)    " "LPVS_FACTOR_SCHEME=steps" $n
done
python tools/factor_check.py 2176 4096 8192 >> "$O" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/ftl" -o tl -- python3 "$R/tools/factor_time.py" 8192 3 > "$R/gpurun_out/ftl.log" 2>&1
cd "$R"
F=$(find gpurun_out/ftl -name "*kernel_trace.csv" | head -1)
python tools/factor_timeline.py "$F" > gpurun_out/factor_timeline_groups.txt 2>&1
rm -rf gpurun_out/ftl
cat "$O"
