#!/usr/bin/env python3
"""Launch time of the multi-signal ADMM mat-vec (ns right-hand sides sharing M) at the cfg5 matrix size.
usage: matvec_multi.py [Nf] [Nv] [ns ...]     env: LPVS_MULTI_MATVEC=stream|dma|valu, LPVS_M_STORAGE=f64"""
import os; os.environ.setdefault("LPVS_EXPERIMENTS", "1")   # this tool flips experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L

Nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Nv = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = 1 << 14
for ns in [int(a) for a in sys.argv[3:]] or [8, 16]:
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * (10.0 * N / 500)).values
    V = torch.linspace(0, 1, N, dtype=torch.float64, device="cuda")
    w = torch.tensor(2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf, dtype=torch.float64, device="cuda")
    Y = torch.randn(N, ns, dtype=torch.float64, device="cuda", generator=g)
    with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
        p.set_prox(L.IndBallL0(32))
        p.admm_init(None, μ=0.05, tol=0.0)
        us, nbytes = p.time_matvec(50)
        it, _, _ = p.admm_run(50)
        tm = p.timing()
        z = p.params(0)
    print(f"n={p.n} ns={ns} storage={os.environ.get('LPVS_M_STORAGE', 'split')} kernel={os.environ.get('LPVS_MULTI_MATVEC', 'stream')}: "
          f"M={nbytes/1e6:.1f} MB  matvec {us:.1f} us  {nbytes/us*1e-3:.0f} GB/s   admm {tm['admm_ms']/tm['admm_iters']*1e3:.1f} us/iter  "
          f"checksum {float(np.abs(z).sum()):.12e}", flush=True)
