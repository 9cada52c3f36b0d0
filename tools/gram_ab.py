"""A/B timing of Gram k-loop schedules in ONE process, interleaved rounds (guide rule 24).
Usage: python tools/gram_ab.py [log2N] [rounds] [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = sys.argv[3:] or ["kr", "krs"]
N, Nf, Nv = 1 << lg, 512, 8
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * (10.0 * N / 500)).values
V = torch.linspace(0, 1, N, dtype=torch.float64, device="cuda")
w = torch.tensor(2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf, dtype=torch.float64, device="cuda")
y = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
res = {v: [] for v in variants}
ref = None
for r in range(rounds):
    for v in variants:
        os.environ["LPVS_GRAM_FORM"] = v
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            tm = p.timing()
            if r == 0:
                G, b = p.get_gram()
                if ref is None: ref = G
                else: print(v, "max|G-ref| =", np.abs(G - ref).max(), "of", np.abs(ref).max())
        res[v].append(tm["gram_ms"])
for v in variants:
    t = np.array(res[v]); fl = tm["gram_flops"]
    print(f"variant {v}: median {np.median(t):.2f} ms  min {t.min():.2f} ms  -> {fl/np.median(t)*1e-9:.2f} TFLOP/s (median)")
