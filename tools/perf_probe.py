"""Timing probe of the headline path (LPV group lasso) at a chosen size; prints the phase timings
measured by HIP events inside the library.  Usage: python tools/perf_probe.py [log2N] [Nf] [Nv] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
Nf = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Nv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
N = 1 << lg
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * (10.0 * N / 500)).values
V = torch.linspace(0, 1, N, dtype=torch.float64, device="cuda")
w = torch.tensor(2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf, dtype=torch.float64, device="cuda")
y = (2 * V ** 2 * torch.cos(w[40] * X) + 2 / (5 * V + 1) * torch.cos(w[204] * X) + 0.1 * torch.randn(N, dtype=torch.float64, device="cuda", generator=g))
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.time()
    p = L.Problem.lpv(y, X, V, w, Nv)
    t1 = time.time()
    p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, Nf, 2 * Nv))
    p.admm_init(None, μ=0.05, tol=0.0)
    t2 = time.time()
    it, nxz, conv = p.admm_run(iters)
    t3 = time.time()
    tm = p.timing()
    n = p.n
    print(f"rep{rep} N=2^{lg} n={n}: create {t1-t0:.3f}s init(factor) {t2-t1:.3f}s admm({it}) {t3-t2:.3f}s nxz={nxz:.3e}")
    print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})
    print(f"    gram {tm['gram_flops']/tm['gram_ms']*1e-9:.2f} TFLOP/s algorithmic ({tm['gram_issued_flops']/tm['gram_ms']*1e-9:.2f} issued); admm {tm['admm_ms']/max(tm['admm_iters'],1)*1e3:.1f} us/iter "
          f"({n*n*8/ (tm['admm_ms']/max(tm['admm_iters'],1)*1e-3)*1e-12:.2f} TB/s of M)")
    p.close()
