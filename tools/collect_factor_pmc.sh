#!/bin/bash
export LPVS_EXPERIMENTS=1   # the schedule knobs below are experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
# Matrix-pipe utilisation of the factorisation's kernels from SQ counters (own pass, kernel-trace only), for the round-2 schedule
# (LPVS_FACTOR_SCHEME=steps) and the default group schedule.  Run on the GPU box:  bash tools/collect_factor_pmc.sh <outdir> [n]
OUT=${1:-gpurun_out/pmc_factor}
N=${2:-8192}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
export TMPDIR=/tmp
cd /tmp
for V in steps groups; do
  if [ $V = steps ]; then export LPVS_FACTOR_SCHEME=steps; else unset LPVS_FACTOR_SCHEME; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
      --output-format csv -d "$R/$OUT/$V" -o f -- python3 "$R/tools/factor_time.py" $N 2 > "$R/$OUT/$V.log" 2>&1
  echo "$V rc=$?"
done
unset LPVS_FACTOR_SCHEME
cd "$R"
python3 - "$OUT" $N <<'PY'
import csv, glob, os, re, sys, json
out, n = sys.argv[1], int(sys.argv[2])
res = {}
for v in ("steps", "groups"):
    vals, dur, calls = {}, {}, {}
    for f in glob.glob(os.path.join(out, v, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"lpvs::.*?(\w+_kernel)", r.get("Kernel_Name", ""))
            if not m:
                continue
            k = m.group(1)
            vals.setdefault(k, {}); vals[k][r["Counter_Name"]] = vals[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for f in glob.glob(os.path.join(out, v, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"lpvs::.*?(\w+_kernel)", r.get("Kernel_Name", ""))
            if m:
                dur[m.group(1)] = dur.get(m.group(1), 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                calls[m.group(1)] = calls.get(m.group(1), 0) + 1
    rows = {}
    for k, c in vals.items():
        if "rank_update" not in k and "pivot_inverse" not in k and "panel" not in k:
            continue
        mf = c.get("SQ_INSTS_VALU_MFMA_F64", 0.0)
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0        # rocprofv3 sums the 8 XCDs
        rows[k] = {"launches": calls.get(k), "seconds_profiled": dur.get(k), "mfma_f64_instructions": mf,
                   "mfma_busy_cycles_per_instruction": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / mf if mf else None,
                   "mfma_busy_fraction_of_kernel_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else None,
                   "valu_instructions_per_mfma": c.get("SQ_INSTS_VALU", 0) / mf if mf else None,
                   "wave_cycles": c.get("SQ_WAVE_CYCLES"), "wait_any": c.get("SQ_WAIT_ANY"), "wait_inst_any": c.get("SQ_WAIT_INST_ANY"),
                   "active_inst_any": c.get("SQ_ACTIVE_INST_ANY"),
                   "achieved_TFLOPs_while_running": mf * 2048 / dur[k] * 1e-12 if dur.get(k) else None}
    res[v] = rows
json.dump({"n": n, "command": "tools/factor_time.py %d 2 under rocprofv3 --pmc (tools/collect_factor_pmc.sh)" % n, "schemes": res},
          open(os.path.join(out, "factor_mfma_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$R/$OUT/steps" "$R/$OUT/groups"
