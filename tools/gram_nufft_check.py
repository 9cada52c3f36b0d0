#!/usr/bin/env python3
"""The structured Gram with its slot sums by non-uniform FFT (nufft.hip, default) against direct evaluation (LPVS_NUDFT=direct,
a second process) at the cfg3 size: time and max |G - G_direct| / max|G|.  usage: gram_nufft_check.py [log2N] [Nf] [Nv] [out.npy]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Nf = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Nv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
y, X, V, w = bench.synth_signal(1 << lg, Nf, 0, "cuda")
for rep in range(2):
    with L.Problem.lpv(y, X, V, w, Nv, True, False) as p:
        tm = p.timing()
        G, b = p.get_gram()
print(f"N=2^{lg} Nf={Nf} Nv={Nv}: form {tm['gram_form']}  basis {tm['basis_ms']:.2f} ms  gram {tm['gram_ms']:.2f} ms  rhs {tm['reduce_rhs_ms']:.2f} ms  LPVS_NUDFT={os.environ.get('LPVS_NUDFT', 'nufft')}")
if len(sys.argv) > 4:
    if os.path.exists(sys.argv[4]):
        d = np.load(sys.argv[4])
        G0, b0 = d["G"], d["b"]
        print("max|G - G_ref| / max|G| = %.3e   rel-Frobenius = %.3e   max|b - b_ref| / max|b| = %.3e" % (
            np.abs(G - G0).max() / np.abs(G0).max(), np.linalg.norm(G - G0) / np.linalg.norm(G0), np.abs(b - b0).max() / np.abs(b0).max()))
    else:
        np.savez(sys.argv[4], G=G, b=b)
