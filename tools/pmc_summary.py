"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (library kernels only).
FETCH_SIZE / WRITE_SIZE are in KiB-units of the counter (x1024 bytes); on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section), so the corrected read figure doubles it."""
import csv, glob, os, re, sys, json
out = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "lpvs::" not in name or r.get("Counter_Name") != c:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"\(.*", "", name).replace("void ", "").replace("lpvs::", "")
            d = res.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "calls": {"FETCH_SIZE": 0, "WRITE_SIZE": 0}})
            d[c] += float(r["Counter_Value"]); d["calls"][c] += 1
rows = []
for k, d in sorted(res.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
    nf, nw = max(d["calls"]["FETCH_SIZE"], 1), max(d["calls"]["WRITE_SIZE"], 1)
    rows.append({"kernel": k, "launches": nf, "fetch_raw_bytes_per_launch": d["FETCH_SIZE"] * 1024 / nf,
                 "fetch_corrected_bytes_per_launch": 2 * d["FETCH_SIZE"] * 1024 / nf,
                 "write_bytes_per_launch": d["WRITE_SIZE"] * 1024 / nw})
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
SOURCES = {"cfg3": ("admm_device.h", "admm_one_launch.hip"), "cfg4": ("admm_device.h", "admm_one_launch.hip"),
           "cfg2": ("admm_device.h", "admm_small.hip"), "cfg5": ("admm_device.h", "admm_multi.hip")}     # bench.KERNEL_SOURCES
for rel in SOURCES[workload]:
    h.update(open(os.path.join(root, "lpvspectral.jl_amd", "csrc", rel), "rb").read())
stamp = h.hexdigest()[:16]
rows.insert(0, {"kernel": "__meta__", "kernel_sources_sha16": stamp, "workload": workload, "command": "bench.py, workload %s, one step of a few iterations (tools/collect_pmc.sh)" % workload})
json.dump(rows, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
rows = rows[1:]
for r in rows[:12]:
    print("%-28s launches %5d  fetch(raw) %10.1f MB  fetch(x2) %10.1f MB  write %10.1f MB" % (
        r["kernel"], r["launches"], r["fetch_raw_bytes_per_launch"] / 1e6, r["fetch_corrected_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6))
