"""cfg5 shape: ns channels sharing (X,V), N=2^log2n, Nf=1024, Nv=16 (n=32768), IndBallL0(32).
Usage: bench_cfg5.py [log2n] [ns] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
N, Nf, Nv = 1 << lg, 1024, 16
g = torch.Generator(device="cuda").manual_seed(5)
X = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * (10.0 * N / 500)).values
V = torch.linspace(0, 1, N, dtype=torch.float64, device="cuda")
w = torch.tensor(2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf, dtype=torch.float64, device="cuda")
Y = torch.stack([sum((1.0 + 0.3 * k) * torch.cos(w[(37 * q + 101 * k) % Nf] * X + 0.1 * k) * (1 + V * (k % 2)) for k in range(3 + q % 4))
                 + 0.1 * torch.randn(N, dtype=torch.float64, device="cuda", generator=g) for q in range(ns)], dim=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
p = L.Problem.lpv_multi(Y, X, V, w, Nv)
t1 = time.perf_counter()
p.set_prox(L.IndBallL0(32))
p.admm_init(None, μ=0.05, tol=0.0)
t2 = time.perf_counter()
it, nxz, conv = p.admm_run(iters)
t3 = time.perf_counter()
P = p.params(0)
tm = p.timing()
print(f"cfg5 shape N=2^{lg} n={p.n} ns={ns}: create {t1-t0:.2f}s (gram {tm['gram_ms']/1e3:.2f}s, {tm['gram_flops']/tm['gram_ms']*1e-9:.1f} TF alg, "
      f"{tm['gram_issued_flops']/tm['gram_ms']*1e-9:.1f} TF issued) factor {t2-t1:.2f}s admm {t3-t2:.2f}s ({(t3-t2)/it*1e3:.3f} ms/iter for {ns} signals) "
      f"total {t3-t0:.2f}s -> {ns/(t3-t0):.3f} signals/s; nnz per signal {[int(np.count_nonzero(P[:, q])) for q in range(ns)]}")
p.close()
