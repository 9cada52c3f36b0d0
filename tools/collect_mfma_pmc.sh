#!/bin/bash
# MFMA utilisation of the dense Gram kernel (gram_kernel<KRS>: the general-w path, bench.py's `gram_general_path` -- NOT skipped here) from SQ
# counters (own pass, kernel-trace only).  Run on the GPU box:
#   bash tools/collect_mfma_pmc.sh <outdir>
OUT=${1:-gpurun_out/pmc_mfma}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv \
    -d "$R/$OUT/sq" -o bench -- python3 "$R/bench.py" --steps 1 --warmup 0 --iters 20 --no-cpu-baseline --no-cfg4-strong --no-baseline-configs --no-concurrent --no-single-process --no-alt-storage > "$R/$OUT/sq.log" 2>&1
echo "rc=$?"
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, json
out = sys.argv[1]
vals, dur, launches = {}, 0.0, 0
for f in glob.glob(os.path.join(out, "sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gram_kernel" in r.get("Kernel_Name", ""):
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for f in glob.glob(os.path.join(out, "sq", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gram_kernel" in r.get("Kernel_Name", ""):
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9; launches += 1     # (the counters are summed over the launches too)
cyc = vals.get("GRBM_GUI_ACTIVE", 0) / 8.0            # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
res = {"raw": vals, "gram_kernel_launches": launches, "gram_seconds_profiled": dur, "gpu_cycles": cyc,
       "achieved_TFLOPs_issued_while_profiled": vals.get("SQ_INSTS_VALU_MFMA_F64", 0) * 2048 / dur * 1e-12 if dur else None,   # 16x16x4: 2048 flop per instruction
       "effective_clock_GHz": cyc / dur * 1e-9 if dur else None,
       "mfma_busy_fraction": vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else None,
       "mfma_f64_instructions": vals.get("SQ_INSTS_VALU_MFMA_F64"),
       "cycles_per_mfma_per_simd": cyc * 1024 / vals["SQ_INSTS_VALU_MFMA_F64"] if vals.get("SQ_INSTS_VALU_MFMA_F64") else None}
json.dump(res, open(os.path.join(out, "mfma_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
