"""Trim a rocprofv3 *_kernel_stats.csv to the library's own kernels (drops torch's input-generation kernels)
and shorten the names.  Usage: python tools/trim_stats.py in.csv out.csv"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "lpvs::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "PercentOfAllKernels", "MinNs", "MaxNs"])
    for r in keep:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"\(.*", "", name).replace("void ", "").replace("lpvs::", "")
        w.writerow([name, r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]),
                    "%.3f" % (100 * float(r["TotalDurationNs"]) / tot), r["MinNs"], r["MaxNs"]])
print(open(sys.argv[2]).read())
