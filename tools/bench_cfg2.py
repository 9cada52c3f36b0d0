"""cfg2: ls_sparse_spectral NormL1(0.01), N=2^18, Nf=512 (n=1024), 5000 ADMM iterations (tol=0), one MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
N, Nf, iters = 1 << 18, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 5000
g = torch.Generator(device="cuda").manual_seed(2)
t = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * N).values
f = torch.tensor(np.arange(1, Nf + 1) / 1024.0, dtype=torch.float64, device="cuda")
amp = [(2, 16), (1, 99), (.5, 256), (.25, 299), (.1, 479)]
y = sum(a * torch.sin(2 * np.pi * f[i] * t + 0.3 * i) for a, i in amp) + 0.1 * torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    p = L.Problem.fourier(y, t, f)
    t1 = time.perf_counter()
    p.set_prox(L.NormL1(0.01)); p.admm_init(None, μ=0.05, tol=0.0)
    t2 = time.perf_counter()
    it, nxz, conv = p.admm_run(iters)
    t3 = time.perf_counter()
    x = p.params(0); tm = p.timing(); p.close()
    print(f"rep{rep}: create {1e3*(t1-t0):.2f} ms (panel {tm['basis_ms']:.2f}, gram {tm['gram_ms']:.2f} = {tm['gram_flops']/tm['gram_ms']*1e-9:.1f} TF, rhs {tm['reduce_rhs_ms']:.2f}) "
          f"factor {1e3*(t2-t1):.2f} ms admm {1e3*(t3-t2):.1f} ms = {(t3-t2)/it*1e6:.2f} us/iter ({it/(t3-t2):.0f} iters/s) total {1e3*(t3-t0):.1f} ms; peaks {sorted(np.argsort(-np.abs(x))[:5]+1)}")
