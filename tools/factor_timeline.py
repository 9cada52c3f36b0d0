#!/usr/bin/env python3
"""Per-step timeline of the factorisation from a rocprofv3 kernel trace: factor_timeline.py <kernel_trace.csv> [step]"""
import csv, re, sys
tr = [r for r in csv.DictReader(open(sys.argv[1])) if "lpvs" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
ru = [r for r in tr if "rank_update" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ru) // 2
start = int(ru[k]["Start_Timestamp"])
for r in tr:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if start - 5000 <= s <= start + 1700000:
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:40]
        print(f"{(s-start)/1e3:9.1f} -> {(e-start)/1e3:9.1f} us  {(e-s)/1e3:8.1f}  q={r.get('Queue_Id','?')} {name}")
