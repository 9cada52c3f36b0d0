import sys, time, numpy as np
sys.path.insert(0, '.')
import lpvspectral_jl_amd as L
rng = np.random.default_rng(0)
for (nw, nlen, Nf, Nv) in ((40, 5000, 64, 8), (64, 600, 12, 4), (200, 2000, 32, 4)):
    N = nw * nlen
    X = np.sort(rng.random(N) * 10.0 * nw); V = np.tile(np.linspace(0, 1, nlen), nw)
    w = 2 * np.pi * np.arange(1, Nf + 1) * 0.5
    Y = np.cos(w[3] * X) * (1 + V) + 0.1 * rng.standard_normal(N)
    for fl in (1, 2, 4):
        L.ls_windowpsd_lpv(Y, X, V, w, Nv, nw, 0, λ=0.02, in_flight=fl)
        t0 = time.perf_counter(); S = L.ls_windowpsd_lpv(Y, X, V, w, Nv, nw, 0, λ=0.02, in_flight=fl); t1 = time.perf_counter()
        print(f"{nw} windows x {nlen} samples, n = {2*Nf*Nv}: library (batched factorisations / solves), in_flight={fl}: {(t1-t0)*1e3:.1f} ms")
    t0 = time.perf_counter(); S2 = L.ls_windowpsd_lpv(Y, X, V, w, Nv, nw, 0, λ=0.02, covariance=False, in_flight=2); t1 = time.perf_counter()
    print(f"   wrapper's per-window loop (single-handle solves, 2 in flight): {(t1-t0)*1e3:.1f} ms   rel diff {np.linalg.norm(S-S2)/np.linalg.norm(S2):.1e}")
