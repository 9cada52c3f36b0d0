#!/usr/bin/env python3
"""How many bits does the packed inverse need?  Takes cfg3's M = (G + I/mu)^-1 from the device and compares storage formats by the
error they put into the offset-form product  M~ v  (v = the iteration's (z-u)/mu at the end of a 2000-iteration solve, and random v):
   rel40 : the shipped format, every element rounded to 40 significant bits
   fixB  : off-diagonal tiles as B-bit fixed point against a per-tile (or per tile-row) scale, diagonal tiles rel40
usage: quant_study.py [log2N] [Nf] [Nv]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Nf = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Nv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
y, X, V, w = bench.synth_signal(1 << lg, Nf, 0, "cuda")
with L.Problem.lpv(y, X, V, w, Nv, True, False) as p:
    p.set_prox(L.SlicedSeparableSum.frequency_groups(bench.LAMBDA, Nf, 2 * Nv))
    p.admm_init(None, μ=bench.MU, tol=0.0)
    p.admm_run(2000)
    x, z, u = p.admm_get()
    M = p.get_inverse(1.0 / bench.MU)
n = M.shape[0]
T = 128
nb = -(-n // T)
v = (z - u) / bench.MU
rng = np.random.default_rng(0)
vr = rng.standard_normal(n)
print(f"n={n}  max|M|={np.abs(M).max():.3e}  |diag| {np.abs(np.diag(M)).min():.3e}..{np.abs(np.diag(M)).max():.3e}  |v|={np.linalg.norm(v):.3e} |x|={np.linalg.norm(x):.3e}")
tm = np.array([[np.abs(M[i*T:(i+1)*T, j*T:(j+1)*T]).max() for j in range(nb)] for i in range(nb)])
off = tm[~np.eye(nb, dtype=bool)]
print("log2(tile max / global max): diagonal tiles", np.round(np.log2(np.diag(tm) / tm.max()), 1)[:8], "... off-diagonal min/median/max",
      np.round(np.log2(np.array([off.min(), np.median(off), off.max()]) / tm.max()), 1))
# dynamic range inside off-diagonal tiles: median |entry| relative to the tile max
i, j = 5, 2
blk = np.abs(M[i*T:(i+1)*T, j*T:(j+1)*T])
print("inside tile (5,2): log2(median/max) = %.1f, log2(min/max) = %.1f" % (np.log2(np.median(blk) / blk.max()), np.log2(blk.min() / blk.max() + 1e-300)))

def rel_round(A, bits):
    m, e = np.frexp(A)
    return np.ldexp(np.round(np.ldexp(m, bits)) , e - bits)

def fix_round(A, bits, mode):
    out = A.copy()
    for i in range(nb):
        for j in range(nb):
            if i == j: continue
            blk = A[i*T:(i+1)*T, j*T:(j+1)*T]
            if mode == "tile":
                s = np.abs(blk).max()
                q = np.ldexp(1.0, int(np.ceil(np.log2(s))) - (bits - 1))
                out[i*T:(i+1)*T, j*T:(j+1)*T] = np.round(blk / q) * q
            else:   # one scale per row of the tile
                s = np.abs(blk).max(axis=1, keepdims=True)
                q = np.ldexp(1.0, np.ceil(np.log2(s)).astype(int) - (bits - 1))
                out[i*T:(i+1)*T, j*T:(j+1)*T] = np.round(blk / q) * q
    return out

M40 = rel_round(M, 40)
ref = M @ v; refr = M @ vr
def report(name, Mq):
    e = np.linalg.norm((Mq - M) @ v) / np.linalg.norm(x)
    er = np.linalg.norm((Mq - M) @ vr) / np.linalg.norm(refr)
    print(f"{name:14s}: |dM v|/|x| = {e:.3e}   random v: |dM v|/|M v| = {er:.3e}", flush=True)
report("rel40", M40)
report("rel37", rel_round(M, 37))
for bits in (40, 38, 36, 32):
    Mq = fix_round(M40, bits, "tile")
    report(f"fix{bits}/tile", Mq)
    Mq = fix_round(M40, bits, "row")
    report(f"fix{bits}/row", Mq)
