#!/usr/bin/env python3
"""cfg5 at its judged size (8 channels, N = 2^20, Nf = 1024, Nv = 16 -> n = 32768, IndBallL0(32), 2000 iterations): what the x-update correction
(the constant forcing E w of an explicit inverse; default OFF for handles with several right-hand sides until round 6) moves --
rel-L2 of x, z, u between a corrected and an uncorrected run per channel and count, the supports, and what a correction costs.
(VERDICT round 5, next #1b.)  No CPU oracle runs this size; tools/cfg5_midsize_vs_oracle.py holds the same kernel chain to the oracle at n = 8192.
usage: cfg5_xcorr_fullsize.py [counts ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench

counts = [int(c) for c in sys.argv[1:]] or [200, 500, 1000, 2000]
C5 = bench.CFG5
Y, X, V, w = bench.synth_channels(1 << C5["log2n"], C5["Nf"], C5["channels_per_gpu"], torch.device("cuda"))
rel = lambda p, q: float(np.linalg.norm(p - q) / max(np.linalg.norm(q), 1e-300))
legs = {}
for leg in ("off", "on"):
    with L.Problem.lpv_multi(Y, X, V, w, C5["Nv"]) as p:
        p.set_option("xupdate_correction", leg)
        p.set_prox(L.IndBallL0(C5["r"]))
        p.admm_init(None, μ=C5["mu"], tol=0.0)
        out, done = {}, 0
        torch.cuda.synchronize(); t0 = time.time()
        for c in counts:
            p.admm_run(c - done); done = c
            out[c] = p.admm_get()
        dt = time.time() - t0
        tm = p.timing()
        legs[leg] = out
        print(f"xupdate_correction={leg}: {counts[-1]} iterations in {dt:.3f} s (with the read-backs); admm {tm.get('admm_ms', 0):.1f} ms, "
              f"corrections {tm['xcorr_count']} in {tm.get('xcorr_ms', 0):.2f} ms; kernel {p.matvec_info()['kernel']}", flush=True)
ns = Y.shape[1]
for c in counts:
    for q in range(ns):
        (x0, z0, u0), (x1, z1, u1) = legs["off"][c], legs["on"][c]
        same = np.array_equal(z0[:, q] != 0, z1[:, q] != 0)
        print(f"  {c:5d} iterations, channel {q}: uncorrected vs corrected  x {rel(x0[:, q], x1[:, q]):.2e}  z {rel(z0[:, q], z1[:, q]):.2e}  u {rel(u0[:, q], u1[:, q]):.2e}  support {'identical' if same else 'DIFFERS'}")
