#!/usr/bin/env python3
"""cfg5 (8 channels, n = 32768, IndBallL0(32)) in ONE process, one handle: the multi-signal product's two walks (LPVS_MULTI_WALK=runs: round 3's
segments of ~8 tiles, per-tile column-sum records; default: column panels with the column sums in LDS) and the x-update correction on / off --
product launch time (HIP events, 30 back-to-back launches), iteration time over 300 iterations, and the iterates of the two walks against each other.
usage: cfg5_ab.py [iters]"""
import os; os.environ.setdefault("LPVS_EXPERIMENTS", "1")   # this tool flips experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N, Nf, Nv, ns = 1 << 20, 1024, 16, 8
Y, X, V, w = bench.synth_channels(N, Nf, ns, torch.device("cuda"))
res = {}
with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
    p.set_prox(L.IndBallL0(32))
    for rep in range(2):
        for walk in ("runs", "panel"):
            for xc in ("0", "2"):
                os.environ["LPVS_MULTI_WALK"] = walk   # (the library default is "runs")
                os.environ["LPVS_XUPDATE_CORRECTION"] = xc
                p.admm_init(None, μ=0.05, tol=0.0)
                mv_us, mv_bytes = p.time_matvec(30)
                p.admm_init(None, μ=0.05, tol=0.0)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                p.admm_run(iters)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                res[(walk, xc)] = p.admm_get()
                print(f"rep {rep} walk {walk:5s} correction {xc}: product {mv_us:7.1f} us per launch ({mv_bytes/1e9:.2f} GB of tiles), "
                      f"{iters} iterations {(t1-t0)*1e3:7.1f} ms = {(t1-t0)/iters*1e3:.4f} ms per iteration; kernel {p.matvec_info()['kernel']}", flush=True)
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for xc in ("0", "2"):
    a, b = res[("runs", xc)], res[("panel", xc)]
    print(f"correction {xc}: panel vs runs after {iters} iterations: x {rel(b[0], a[0]):.2e} z {rel(b[1], a[1]):.2e} u {rel(b[2], a[2]):.2e}; same support {np.array_equal(a[1] != 0, b[1] != 0)}")
a, b = res[("panel", "0")], res[("panel", "2")]
print(f"panel walk: correction 2 vs 0: z {rel(b[1], a[1]):.2e}; same support {np.array_equal(a[1] != 0, b[1] != 0)}")
