import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L, bench
CFG4 = bench.CFG4
for nwin in (32, 64, 384, 512, 768):
    n = 1 << CFG4["log2n"] if "log2n" in CFG4 else 1 << 16
    y, t, f = bench.synth_windows(nwin, n, CFG4["Nf"], torch.device("cuda")) if hasattr(bench, "synth_windows") else (None, None, None)
    def solve(lo, hi, out, k):
        out[k] = L.windowpsd_sparse_batched(y, t, f, n, 0, None, λ=CFG4["lam"], μ=CFG4["mu"], tol=0.0, iters=2000, win_lo=lo, win_hi=hi, device=0)
    for parts in (1, 2, 1, 2):
        out = [None] * parts
        solve(0, 8, [None], 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=solve, args=(nwin * k // parts, nwin * (k + 1) // parts, out, k)) for k in range(parts)]
        [q.start() for q in th]; [q.join() for q in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"nwin {nwin} in {parts} part(s) in flight: {dt*1e3:.1f} ms", flush=True)
