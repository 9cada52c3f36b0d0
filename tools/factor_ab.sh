#!/bin/bash
# A/B of the factorisation schedules on the GPU box (from the repo root): one-panel steps (round 2), pairs with the separate gather + GEMM
# kernels, pairs with the fused chain tail; then a kernel timeline of the default at n = 8192.  Output: gpurun_out/factor_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out/factor_ab.txt"; : > "$O"
cd "$R"
for n in 2048 4096 8192 12288 16384; do
  echo "== n=$n" >> "$O"
  LPVS_FACTOR_SCHEME=steps python tools/factor_time.py $n 4 2>&1 | sed 's/^/steps        /' >> "$O"
  LPVS_CHAIN=split python tools/factor_time.py $n 4 2>&1 | sed 's/^/pairs+split  /' >> "$O"
  python tools/factor_time.py $n 4 2>&1 | sed 's/^/pairs+fused  /' >> "$O"
  if [ $n -ge 12288 ]; then
    LPVS_KW=128 python tools/factor_time.py $n 4 2>&1 | sed 's/^/pairs KW=128 /' >> "$O"
  fi
done
python tools/factor_check.py 2048 2176 4096 8192 >> "$O" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/ftl" -o tl -- python3 "$R/tools/factor_time.py" 8192 3 > "$R/gpurun_out/ftl.log" 2>&1
cd "$R"
F=$(find gpurun_out/ftl -name "*kernel_trace.csv" | head -1)
python tools/factor_timeline.py "$F" > gpurun_out/factor_timeline_pairs.txt 2>&1
rm -rf gpurun_out/ftl
tail -40 "$O"
