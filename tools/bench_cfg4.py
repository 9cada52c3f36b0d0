"""cfg4: ls_windowpsd(estimator=ls_sparse_spectral) on 2^16-sample windows, Nf=256 (zero frequency first,
Nreg=511), L1 lambda=0.2, mu=1e-4, 2000 iterations per window (tol=0).  Usage: bench_cfg4.py [nwin] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n = 1 << 16
Lh = nwin * n
g = torch.Generator(device="cuda").manual_seed(4)
t = torch.arange(Lh, dtype=torch.float64, device="cuda")
f = np.arange(256) / 512.0
y = (torch.sin(2 * np.pi * f[33] * t) + 0.5 * torch.cos(2 * np.pi * f[100] * t) + 0.1 * torch.randn(Lh, dtype=torch.float64, device="cuda", generator=g))
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    x, S, its = L.windowpsd_sparse_batched(y, t, f, n, 0, None, λ=0.2, μ=1e-4, tol=0.0, iters=iters)
    dt = time.perf_counter() - t0
    print(f"rep{rep}: {nwin} windows x 2^16, Nf=256, {iters} iters: {dt:.3f} s -> {nwin/dt:.1f} windows/s, {nwin*iters/dt:.0f} window-iters/s; "
          f"argmax S = {int(np.argmax(S))+1}, iters {its.min()}..{its.max()}")
