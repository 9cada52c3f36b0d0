#!/usr/bin/env python3
"""Adjudicates the cases of tests/test_gpu_offfamily.py: for each case (or the ones named) the device's default numerics and its
variants (36-bit reads, uniform 6-byte elements, doubles, no x-update correction) against BOTH CPU forms of the Gram-form ADMM on the
device Gram -- the f64 oracle (Cholesky) and the same algorithm in x87 extended precision, which says which f64 side carries a difference.
Prints max over x, z of rel-L2, and u, per leg; cond(G + I/mu) from the extreme eigenvalues.
usage: offfamily_probe.py [--iters 600] [--variants] [case-id substrings ...]"""
import os; os.environ.setdefault("LPVS_EXPERIMENTS", "1")   # this tool flips experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lpvspectral_jl_amd as L
from oracle import oracle as o
import test_gpu_offfamily as T

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=T.ITERS)
ap.add_argument("--variants", action="store_true")
ap.add_argument("--refine", action="store_true", help="legs with the offset vector refined at lpvs_admm_init / early corrections (environment knobs)")
ap.add_argument("--save", default=None, help="write the extended-precision iterates of every case run as the fixture tests/golden/offfamily_exact_iterates.npz")
ap.add_argument("sel", nargs="*")
a = ap.parse_args()
rel = T.rel
saved = {}
for case, cid in zip(T.CASES, T.IDS):
    if a.sel and not any(s in cid for s in a.sel):
        continue
    n, vkind, norm, mu, density, wkind, kind = case
    y, X, V, w, Nf = T.make_inputs(n, vkind, wkind, seed=1000 + T.CASES.index(case))
    legs = [("default", {})]
    if a.variants:
        legs += [("36-bit reads", dict(storage="mixed")), ("6-byte", dict(storage="split")), ("doubles", dict(storage="f64")),
                 ("default, uncorrected", dict(xupdate_correction="off")), ("two-launch", dict(iteration="two"))]
    if a.refine:
        legs += [("xb refined at init", dict(env=dict(LPVS_XB_REFINE="1"))), ("early corrections q512", dict(env=dict(LPVS_XUPDATE_CORRECTION="q512"))),
                 ("refined + q512", dict(env=dict(LPVS_XB_REFINE="1", LPVS_XUPDATE_CORRECTION="q512"))), ("refined, uncorrected", dict(xupdate_correction="off")),
                 ("refined + d128", dict(env=dict(LPVS_XB_REFINE="1", LPVS_XUPDATE_CORRECTION="d128")))]
    dev = {}
    for name, opts in legs:
        opts = dict(opts)
        env = opts.pop("env", {})
        os.environ.update(env)
        with L.Problem.lpv(y, X, V, w, T.NV, norm, False) as p:
            G, b = p.get_gram()
            lam = T.penalties(G, b, Nf, mu, density)[kind]
            prox, oprox = {"group": (L.SlicedSeparableSum.frequency_groups(lam, Nf, 2 * T.NV), o.GroupL2(lam, 2 * T.NV)),
                           "l1": (L.NormL1(lam), o.NormL1(lam)), "l0": (L.NormL0(lam), o.NormL0(lam))}[kind]
            for k, v in opts.items():
                p.set_option(k, v)
            p.set_prox(prox)
            p.admm_init(None, μ=mu, tol=0.0)
            info = p.matvec_info()
            p.admm_run(a.iters)
            dev[name] = (p.admm_get(), info["kernel"], info["storage"][:60], p.timing()["xcorr_count"])
        for k in env:
            os.environ.pop(k)
    ev = np.linalg.eigvalsh(G)
    cond = (ev[-1] + 1 / mu) / (max(ev[0], 0) + 1 / mu)
    t0 = time.time(); ro = o.admm_gram(G, b, oprox, iters=a.iters, tol=0.0, mu=mu); t1 = time.time()
    ld = o.admm_gram_ld(G, b, oprox, [a.iters], mu=mu)[a.iters]; t2 = time.time()
    nz = np.count_nonzero(ld[1])
    saved[cid + "/sha256"] = T.fingerprint(G, b)
    for k, v in zip("xzu", ld):
        saved[cid + "/" + k] = v
    print(f"{cid}: lambda {lam:.3g}, cond(G + I/mu) {cond:.2e}, nnz {nz}/{n}, |x| {np.linalg.norm(ld[0]):.3g} |u| {np.linalg.norm(ld[2]):.3g}; oracle {t1 - t0:.1f} s, extended {t2 - t1:.1f} s")
    print(f"    f64 oracle vs exact: x {rel(ro['x'], ld[0]):.2e} z {rel(ro['z'], ld[1]):.2e} u {rel(ro['u'], ld[2]):.2e}  support {'identical' if np.array_equal(ro['z'] != 0, ld[1] != 0) else 'DIFFERS'}")
    su = max(np.linalg.norm(ld[0]), np.linalg.norm(ld[2]))            # the state's scale: an error of u enters z = prox(x + u) on the same footing as one of x
    for name, ((x, z, u), kern, st, nc) in dev.items():
        print(f"    {name:22s} vs exact: x {rel(x, ld[0]):.2e} z {rel(z, ld[1]):.2e} u {rel(u, ld[2]):.2e} (of the state's scale {np.linalg.norm(u - ld[2]) / su:.2e}) | vs f64 oracle: x {rel(x, ro['x']):.2e} z {rel(z, ro['z']):.2e} u {rel(u, ro['u']):.2e} "
              f"| support {'identical' if np.array_equal(z != 0, ld[1] != 0) else 'DIFFERS'} | {kern}, {nc} corrections", flush=True)
if a.save:
    np.savez_compressed(a.save, iters=np.array(a.iters), **saved)
    print("saved", a.save, os.path.getsize(a.save), "bytes,", len(saved) // 4, "cases")
