#!/usr/bin/env python3
"""cfg3 at the judged size, the bench's own iteration count: admm_iter_mixed_kernel against the CPU oracle's Gram-form ADMM (Cholesky
x-update) on the Gram read back from the device, at several iteration counts.  (The test-suite holds 200 iterations to 1e-9:
tests/test_gpu_judged_size.py; this prints how the difference grows up to the bench's 2000.)  ~4 CPU-minutes on two threads.
usage: cfg3_vs_oracle.py [iteration counts ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
from oracle import oracle as o
import bench
counts = [int(a) for a in sys.argv[1:]] or [200, 500, 1000, 2000]
y, X, V, w = bench.synth_signal(1 << 20, 512, 0, torch.device("cuda"))
dev = {}
with L.Problem.lpv(y, X, V, w, 8) as p:
    G, b = p.get_gram()
    p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, 512, 16))
    p.admm_init(None, μ=0.05, tol=0.0)
    assert p.matvec_info()["kernel"] == "admm_iter_mixed_kernel"
    done = 0
    for c in counts:
        p.admm_run(c - done); done = c
        dev[c] = p.admm_get()
    # the same with the inverse streamed as doubles
    p.set_option("storage", "f64")
    p.admm_init(None, μ=0.05, tol=0.0)
    done = 0
    dev64 = {}
    for c in counts:
        p.admm_run(c - done); done = c
        dev64[c] = p.admm_get()
rel = lambda a, b_: np.linalg.norm(a - b_) / np.linalg.norm(b_)
t0 = time.time()
for c in counts:            # (the oracle restarts for every count: its factorisation dominates, ~40 s each on two threads)
    ro = o.admm_gram(G, b, o.GroupL2(5.0, 16), iters=c, tol=0.0, mu=0.05)
    x, z, u = dev[c]; x8, z8, u8 = dev64[c]
    print(f"{c:5d} iterations: mixed storage vs oracle  x {rel(x, ro['x']):.2e} z {rel(z, ro['z']):.2e} u {rel(u, ro['u']):.2e} same support {np.array_equal(z != 0, ro['z'] != 0)} | "
          f"8-byte storage vs oracle z {rel(z8, ro['z']):.2e} | mixed vs 8-byte z {rel(z, z8):.2e}   nnz {np.count_nonzero(ro['z'])}   [{time.time() - t0:.0f} s]", flush=True)
