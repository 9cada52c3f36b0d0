#!/usr/bin/env python3
"""Mat-vec (symv_tile_kernel) launch time and streamed-bytes rate vs matrix size: shows where M stops fitting the
memory-side cache.  usage: matvec_bw.py [n ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L

for n in [int(a) for a in sys.argv[1:]] or [2048, 4096, 6144, 7168, 8192, 12288]:
    g = torch.Generator(device="cuda").manual_seed(n)
    A = torch.randn(n + 64, n, dtype=torch.float64, device="cuda", generator=g)
    G = (A.T @ A).contiguous(); b = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    del A
    with L.Problem.gram(G, b) as p:
        p.set_prox(L.NormL1(0.1))
        p.admm_init(None, μ=0.05, tol=0.0)
        us, nbytes = p.time_matvec(300)
        it, _, _ = p.admm_run(400)
        tm = p.timing()
    print(f"n={n:6d}  M={nbytes/1e6:8.1f} MB  matvec {us:7.2f} us  {nbytes/us*1e-3:8.1f} GB/s   admm {tm['admm_ms']/tm['admm_iters']*1e3:7.2f} us/iter  factor {tm['factor_ms']:.2f} ms", flush=True)
