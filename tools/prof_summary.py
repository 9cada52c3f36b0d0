#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats csv output: prof_summary.py <kernel_stats.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows[:n]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} ms  avg {float(r['AverageNs'])/1e3:10.2f} us")
