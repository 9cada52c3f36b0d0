import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L, bench
CFG4 = bench.CFG4
nwin, n = 1024, 1 << 16
y, t, f = bench.synth_windows(nwin, n, CFG4["Nf"], torch.device("cuda"))
def solve(lo, hi):
    L.windowpsd_sparse_batched(y, t, f, n, 0, None, λ=CFG4["lam"], μ=CFG4["mu"], tol=0.0, iters=2000, win_lo=lo, win_hi=hi, device=0)
solve(0, 8)
for chunk, parts in ((1024, 1), (256, 2), (320, 2), (384, 2), (384, 3), (448, 2), (512, 2), (341, 2), (192, 2), (1024, 1)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lo = 0
    while lo < nwin:
        hi = min(nwin, lo + chunk)
        th = [threading.Thread(target=solve, args=(lo + (hi - lo) * k // parts, lo + (hi - lo) * (k + 1) // parts)) for k in range(parts)]
        [q.start() for q in th]; [q.join() for q in th]
        lo = hi
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"1024 windows in chunks of {chunk}, {parts} part(s) in flight: {dt*1e3:.1f} ms = {nwin/dt:.0f} windows/s", flush=True)
