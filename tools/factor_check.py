#!/usr/bin/env python3
"""Accuracy and time of M = (G + I/mu)^-1 against torch.linalg.inv (fp64) for several sizes.
usage: factor_check.py [n ...]   (LPVS_FACTOR=sweep64 selects the single-level sweep)"""
import os; os.environ.setdefault("LPVS_EXPERIMENTS", "1")   # this tool flips experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L

for n in [int(a) for a in sys.argv[1:]] or [1000, 1024, 1100, 2048, 6200, 8192]:
    g = torch.Generator(device="cuda").manual_seed(n)
    A = torch.randn(n + 64, n, dtype=torch.float64, device="cuda", generator=g)
    G = (A.T @ A).contiguous(); b = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    del A
    mu = 0.05
    with L.Problem.gram(G, b) as p:
        p.set_prox(L.NormL1(0.1))
        p.admm_init(None, μ=mu, tol=0.0)
        M = torch.as_tensor(p.get_inverse(1.0 / mu)).cuda()
        tm = p.timing()
    H = G + torch.eye(n, dtype=torch.float64, device="cuda") / mu
    R = M @ H - torch.eye(n, dtype=torch.float64, device="cuda")
    rel = float("nan")
    if n <= 4096:
        Mt = torch.linalg.inv(H)
        rel = ((M - Mt).norm() / Mt.norm()).item()
    print(f"n={n:6d}  factor {tm['factor_ms']:8.2f} ms   |M H - I|_max {R.abs().max().item():.2e}   |M - inv|/|inv| {rel:.2e}   asym {(M-M.T).abs().max().item():.1e}", flush=True)
