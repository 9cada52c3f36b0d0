import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L, bench
lg, Nf, Nv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
y, X, V, w = bench.synth_signal(1 << lg, Nf, 0, torch.device("cuda"))
p = L.Problem.lpv(y, X, V, w, Nv)
p.set_prox(L.IndBallL0(32)); p.admm_init(None, μ=0.05, tol=0.0)
t0 = time.perf_counter(); it, nxz, conv = p.admm_run(300); t1 = time.perf_counter()
print(f"n={p.n}: ball prox path {1e6*(t1-t0)/it:.1f} us/iter")
