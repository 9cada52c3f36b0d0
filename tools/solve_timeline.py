#!/usr/bin/env python3
"""Where a cfg3 solve's wall-clock goes on the HOST side: the calls of bench.solve() timed one by one (perf_counter, the device idle
before the constructor and synchronised by the library inside admm_run / params), next to the HIP-event phases the handle reports.
usage: solve_timeline.py [solves]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
y, X, V, w = bench.synth_signal(1 << 20, 512, 0, dev)
prox = L.SlicedSeparableSum.frequency_groups(bench.LAMBDA, len(w), 2 * bench.NV)
rows = []
for k in range(n + 2):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    p = L.Problem.lpv(y, X, V, w, bench.NV, True, False, device=0); t.append(time.perf_counter())
    p.set_prox(prox); t.append(time.perf_counter())
    p.admm_init(None, μ=bench.MU, tol=0.0); t.append(time.perf_counter())
    p.admm_run(2000); t.append(time.perf_counter())
    par = p.params(0); t.append(time.perf_counter())
    tm = p.timing(); t.append(time.perf_counter())
    p.close(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    if k >= 2:
        rows.append(np.diff(t) * 1e3)
        ph = tm
names = ["constructor", "set_prox", "admm_init", "admm_run", "params", "timing", "close", "sync"]
r = np.array(rows)
print("host wall-clock per call, ms (mean / min over %d solves):" % n)
for i, nm in enumerate(names):
    print(f"  {nm:12s} {r[:, i].mean():8.3f} {r[:, i].min():8.3f}")
print(f"  total        {r.sum(1).mean():8.3f} {r.sum(1).min():8.3f}")
print("HIP-event phases of the last solve, ms: " + ", ".join(f"{k} {ph[k]:.3f}" for k in ("basis_ms", "gram_ms", "reduce_rhs_ms", "factor_ms", "admm_ms", "xcorr_ms")))
print(f"  constructor - (basis + gram + reduce_rhs) = {r[:, 0].mean() - ph['basis_ms'] - ph['gram_ms'] - ph['reduce_rhs_ms']:.3f} ms (the constructor does not wait for the device)")
print(f"  admm_init + admm_run - (factor + admm)    = {r[:, 2].mean() + r[:, 3].mean() - ph['factor_ms'] - ph['admm_ms']:.3f} ms")
