#!/usr/bin/env python3
"""ADMM iteration time at the cfg3 size for the two iteration schemes (LPVS_ITERATION=two: mat-vec + update launches; default: one
launch per iteration with fixed-point accumulation) and two prox operators (frequency-grouped lasso as cfg3; plain L1).
usage: iter_time.py [log2N] [Nf] [Nv] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Nf = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Nv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
y, X, V, w = bench.synth_signal(1 << lg, Nf, 0, "cuda")
res = {}
for prox_name, prox in (("group", L.SlicedSeparableSum.frequency_groups(2.0, Nf, 2 * Nv)), ("l1", L.NormL1(0.5))):
    for mode in ("two", "one"):
        if mode == "two":
            os.environ["LPVS_ITERATION"] = "two"
        else:
            os.environ.pop("LPVS_ITERATION", None)
        with L.Problem.lpv(y, X, V, w, Nv, True, False) as p:
            p.set_prox(prox)
            p.admm_init(None, μ=0.05, tol=0.0)
            p.admm_run(50)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            it, nxz, conv = p.admm_run(iters)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            x, z, u = p.admm_get()
        res[(prox_name, mode)] = z
        print(f"{prox_name:6s} {mode}: {dt / iters * 1e6:7.2f} us per iteration  ({it} iterations, nxz {nxz:.6e}, nnz {np.count_nonzero(z)})", flush=True)
    a, b = res[(prox_name, "two")], res[(prox_name, "one")]
    print(f"{prox_name:6s} rel-L2(z_one - z_two) = {np.linalg.norm(a - b) / np.linalg.norm(a):.2e}, same support: {np.array_equal(a != 0, b != 0)}")
