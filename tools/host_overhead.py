#!/usr/bin/env python3
"""Host wall-clock of each API call of a cfg3 solve next to the HIP-event phase timings (what is not kernel time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
y, X, V, w = bench.synth_signal(1 << 20, 512, 0, torch.device("cuda"))
torch.cuda.synchronize()
for rep in range(4):
    t = [time.perf_counter()]
    p = L.Problem.lpv(y, X, V, w, 8, True, False); t.append(time.perf_counter())
    p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, 512, 16)); t.append(time.perf_counter())
    p.admm_init(None, μ=0.05, tol=0.0); t.append(time.perf_counter())
    p.admm_run(2000); t.append(time.perf_counter())
    prm = p.params(0); t.append(time.perf_counter())
    tm = p.timing(); p.close(); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print(f"rep{rep}: create {d[0]:.2f} (basis+gram+rhs events {tm['basis_ms']+tm['gram_ms']+tm['reduce_rhs_ms']:.2f})  set_prox {d[1]:.2f}  "
          f"init {d[2]:.2f} (factor event {tm['factor_ms']:.2f})  run {d[3]:.2f} (admm event {tm['admm_ms']:.2f})  params {d[4]:.2f}  close {d[5]:.2f}  total {sum(d):.2f}")
