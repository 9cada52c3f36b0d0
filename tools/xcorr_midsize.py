#!/usr/bin/env python3
"""The x-update correction at a size the extended-precision oracle runs in seconds (n = 2048): device iterates with and without it against
oracle.admm_gram_ld and oracle.admm_gram on the device Gram.  usage: xcorr_midsize.py [log2N] [Nf] [iters]"""
import os; os.environ.setdefault("LPVS_EXPERIMENTS", "1")   # this tool flips experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
from oracle import oracle as o
import bench
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Nf = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
y, X, V, w = bench.synth_signal(1 << lg, Nf, 0, torch.device("cuda"))
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
snaps = [iters // 4, iters // 2, iters]
res = {}
for sc in ("0", "d512", "2"):
    os.environ["LPVS_XUPDATE_CORRECTION"] = sc
    with L.Problem.lpv(y, X, V, w, 8) as p:
        G, b = p.get_gram()
        p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, Nf, 16))
        p.admm_init(None, μ=0.05, tol=0.0)
        done, out = 0, []
        for c in snaps:
            p.admm_run(c - done); done = c
            out.append(p.admm_get())
        res[sc] = out
        kern = p.matvec_info()["kernel"]
t0 = time.time(); ld = o.admm_gram_ld(G, b, o.GroupL2(5.0, 16), snaps, mu=0.05); t1 = time.time()
print(f"n = {len(b)}, kernel {kern}, extended-precision oracle {t1 - t0:.1f} s, nnz {[int(np.count_nonzero(ld[c][1])) for c in snaps]}")
for k, c in enumerate(snaps):
    ro = o.admm_gram(G, b, o.GroupL2(5.0, 16), iters=c, tol=0.0, mu=0.05)
    print(f"{c:5d} iterations: f64 oracle vs exact z {rel(ro['z'], ld[c][1]):.2e} | " + " | ".join(
        f"device ({sc}) vs exact x {rel(res[sc][k][0], ld[c][0]):.2e} z {rel(res[sc][k][1], ld[c][1]):.2e} u {rel(res[sc][k][2], ld[c][2]):.2e}" for sc in res))
