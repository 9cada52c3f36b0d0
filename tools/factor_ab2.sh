#!/bin/bash
# pair schedule: CUs left to the pivot chains (LPVS_RESERVE_CUS) at n = 8192 / 4096 / 12288, then a kernel timeline of the default
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out/factor_ab2.txt"; : > "$O"
cd "$R"
for n in 8192 4096 12288; do
  for r in 0 2 4 8 16; do
    LPVS_KW=128 LPVS_RESERVE_CUS=$r python tools/factor_time.py $n 4 2>&1 | grep factor | sed "s/^/reserve=$r  /" >> "$O"
  done
done
python tools/factor_check.py 2176 4096 8192 >> "$O" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/ftl" -o tl -- python3 "$R/tools/factor_time.py" 8192 3 > "$R/gpurun_out/ftl.log" 2>&1
cd "$R"
F=$(find gpurun_out/ftl -name "*kernel_trace.csv" | head -1)
python tools/factor_timeline.py "$F" > gpurun_out/factor_timeline_pairs2.txt 2>&1
rm -rf gpurun_out/ftl
cat "$O"
