#!/usr/bin/env python3
"""Variance of the multi-signal mat-vec launch time: repeated timings on one handle, and on fresh handles (fresh allocations) in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
Nf, Nv, N, ns = 1024, 16, 1 << 14, 8
g = torch.Generator(device="cuda").manual_seed(5)
X = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * (10.0 * N / 500)).values
V = torch.linspace(0, 1, N, dtype=torch.float64, device="cuda")
w = torch.tensor(2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf, dtype=torch.float64, device="cuda")
Y = torch.randn(N, ns, dtype=torch.float64, device="cuda", generator=g)
hold = []
for rep in range(4):
    p = L.Problem.lpv_multi(Y, X, V, w, Nv)
    p.set_prox(L.IndBallL0(32))
    p.admm_init(None, μ=0.05, tol=0.0)
    ts = [p.time_matvec(20)[0] for _ in range(5)]
    print(f"handle {rep}: " + " ".join(f"{t:.0f}" for t in ts), flush=True)
    if rep % 2 == 0:
        hold.append(p)            # keep every other handle alive so the next one lands elsewhere
    else:
        p.close()
