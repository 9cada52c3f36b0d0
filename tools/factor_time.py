#!/usr/bin/env python3
"""Factorisation time only (no accuracy check): factor_time.py n [reps]"""
import os; os.environ.setdefault("LPVS_EXPERIMENTS", "1")   # this tool flips experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device="cuda").manual_seed(n)
A = torch.randn(n + 64, n, dtype=torch.float64, device="cuda", generator=g)
G = (A.T @ A).contiguous(); b = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
del A
ts = []
for _ in range(reps):
    with L.Problem.gram(G, b) as p:
        p.set_prox(L.NormL1(0.1))
        p.admm_init(None, μ=0.05, tol=0.0)
        ts.append(p.timing()["factor_ms"])
print(f"n={n} factor ms: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
