#!/usr/bin/env python3
"""Where the time of the one-launch ADMM iteration goes: its duration over a range of problem sizes (Nf at Nv = 8, N = 2^20) against
the bytes it streams, fitted as  T = T0 + bytes / BW.  T0 is what a launch costs besides its bytes (boundary between dependent
launches, ramp, prologue, tail); BW the rate the tile stream settles at.   usage: iter_fit.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rows = []
for Nf in (128, 192, 256, 384, 512, 640, 768, 1024):
    Nv = 8
    y, X, V, w = bench.synth_signal(1 << 20, Nf, 0, "cuda")
    with L.Problem.lpv(y, X, V, w, Nv, True, False) as p:
        p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, Nf, 2 * Nv))
        p.admm_init(None, μ=0.05, tol=0.0)
        info = p.matvec_info()
        _, nbytes = p.time_matvec(3)
        p.admm_run(100)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p.admm_run(iters)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    rows.append((2 * Nf * Nv, nbytes, dt * 1e6))
    print(f"n = {2*Nf*Nv:6d}  {info['kernel']:24s} {nbytes/1e6:8.1f} MB  {dt*1e6:7.2f} us per iteration  {nbytes/dt*1e-12:5.2f} TB/s", flush=True)
B = np.array([r[1] for r in rows]); T = np.array([r[2] for r in rows])
A = np.stack([np.ones_like(B), B], axis=1)
(t0, slope), *_ = np.linalg.lstsq(A, T, rcond=None)
print(f"fit: T = {t0:.2f} us + bytes / {1e-6/slope:.2f} TB/s   (max |residual| {np.abs(A @ [t0, slope] - T).max():.2f} us)")
