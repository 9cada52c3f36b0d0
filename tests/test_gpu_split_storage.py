"""Reduced storage of the tile-packed inverse streamed by the ADMM mat-vec of large single-signal problems, against the 8-byte
storage (LPVS_M_STORAGE=f64) and the CPU oracle.  GPU only.
  split: float head + 16-bit tail, 40 significant bits, 6 bytes per element (LPVS_M_STORAGE=split)
  mixed: the default -- split for the diagonal tiles, 36-bit fixed point with a per-row step (4.53 bytes) for tiles whose entries
         are all small against max|M| (decided per tile when packing; admm.hip)

The split form adds a relative error <= 2^-40 = 9.1e-13 per element of M = (G + I/mu)^-1.  Applied naively (x = M~ (b + v))
that error is amplified by cond(G + I/mu): measured 5.3e-9 rel-L2 in z at the cfg3 size, above the 1e-9 parity bound.  The
library therefore runs the x-update in offset form, x = M b + M~ (z-u)/mu with M b computed once from the full-precision
inverse: measured 1.2e-10 at cfg3 (2000 iterations) and 6e-11 against the oracle at n = 2176 -- both storages are held to
the 1e-9 bound here, with identical supports and stopping iterations."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def _lpv_problem(N, Nf, Nv, seed):
    rng = np.random.default_rng(seed)
    X = np.sort(10 * rng.random(N) * N / 500); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf / 4
    y = 2 * V ** 2 * np.cos(w[12] * X) + 2 / (5 * V + 1) * np.cos(w[60] * X) + 0.1 * rng.standard_normal(N)
    return y, X, V, w


def _solve(L, y, X, V, w, Nv, prox, iters, tol, storage):
    if storage:
        os.environ["LPVS_M_STORAGE"] = storage
    try:
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            p.set_prox(prox)
            p.admm_init(None, μ=0.05, tol=tol)
            info = p.matvec_info()
            us, nbytes = p.time_matvec(5)
            it, nxz, conv = p.admm_run(iters)
            x, z, u = p.admm_get()
            G, b = p.get_gram()
    finally:
        os.environ.pop("LPVS_M_STORAGE", None)
    return dict(x=x, z=z, u=u, it=it, nxz=nxz, conv=conv, info=info, nbytes=nbytes, G=G, b=b)


@pytest.mark.parametrize("kind", ["group", "l1", "ball"])
def test_split_and_f64_storage_against_oracle(L, oracle, kind):
    Nf, Nv = 136, 8                                              # n = 2176: tile-packed path, a ragged last tile row
    y, X, V, w = _lpv_problem(3000, Nf, Nv, 5)
    prox, oprox = {"group": (L.SlicedSeparableSum.frequency_groups(3.0, Nf, 2 * Nv), oracle.GroupL2(3.0, 2 * Nv)),
                   "l1": (L.NormL1(1.0), oracle.NormL1(1.0)),
                   "ball": (L.IndBallL0(20), oracle.IndBallL0(20))}[kind]
    rs = _solve(L, y, X, V, w, Nv, prox, 300, 0.0, "split")
    rm = _solve(L, y, X, V, w, Nv, prox, 300, 0.0, None)         # the default: mixed
    rd = _solve(L, y, X, V, w, Nv, prox, 300, 0.0, "f64")
    npk = 2176 * (2176 + 128) // 2
    assert rs["info"]["kernel"] == "symv_tile_split_kernel" and rs["nbytes"] == 6 * npk
    # few samples: hardly any tile of this inverse is small enough for fixed point, and the handle falls back to the 6-byte format
    assert rm["info"]["kernel"] == "symv_tile_split_kernel" and rm["nbytes"] == 6 * npk
    assert rd["info"]["kernel"] == "symv_tile_kernel<double>" and rd["nbytes"] == 8 * npk
    ro = oracle.admm_gram(rs["G"], rs["b"], oprox, iters=300, tol=0.0, mu=0.05)
    for r in (rs, rm, rd):
        assert r["it"] == 300
        assert rel(r["z"], ro["z"]) <= 1e-9 and rel(r["x"], ro["x"]) <= 1e-9 and rel(r["u"], ro["u"]) <= 1e-9, (kind, rel(r["z"], ro["z"]))
        assert np.array_equal(r["z"] != 0, ro["z"] != 0)
    print(f"{kind}: rel-L2(z) split vs oracle {rel(rs['z'], ro['z']):.2e}, f64 vs oracle {rel(rd['z'], ro['z']):.2e}, split vs f64 {rel(rs['z'], rd['z']):.2e}")


def test_split_storage_single_matvec_accuracy_and_stopping_iteration(L, oracle):
    Nf, Nv = 128, 8
    y, X, V, w = _lpv_problem(3000, Nf, Nv, 6)
    prox = L.SlicedSeparableSum.frequency_groups(3.0, Nf, 2 * Nv)
    a = _solve(L, y, X, V, w, Nv, prox, 1, 0.0, "split")
    m = _solve(L, y, X, V, w, Nv, prox, 1, 0.0, None)
    b = _solve(L, y, X, V, w, Nv, prox, 1, 0.0, "f64")
    assert rel(a["x"], b["x"]) <= 5e-12, rel(a["x"], b["x"])      # one mat-vec: 2^-40 per element, no cancellation to speak of
    assert rel(m["x"], b["x"]) <= 5e-12, rel(m["x"], b["x"])      # mixed: the fixed-point tiles add about a quarter to that
    ro = oracle.admm_gram(a["G"], a["b"], oracle.GroupL2(3.0, 2 * Nv), iters=5000, tol=1e-6, mu=0.05)
    for st in ("split", None):
        c = _solve(L, y, X, V, w, Nv, prox, 5000, 1e-6, st)
        assert c["conv"] and c["it"] == ro["iters"] and rel(c["z"], ro["z"]) <= 1e-9


def test_split_storage_extreme_values_roundtrip(L):
    """Values spanning many binades, exact zeros, negative numbers, ties: the decoded matrix reproduces M to 2^-40 relative.
    (M is read back through one mat-vec per unit vector block: x = M e_j.)"""
    rng = np.random.default_rng(7)
    n = 2048
    # a diagonally dominant SPD matrix whose inverse has entries over ~12 orders of magnitude
    d = np.logspace(-3, 3, n)
    B = rng.standard_normal((n, 8)) * 1e-2
    G = np.diag(d) + B @ B.T
    bvec = rng.standard_normal(n)
    outs = {}
    for storage in ("split", "mixed", "f64"):
        if storage:
            os.environ["LPVS_M_STORAGE"] = storage
        try:
            with L.Problem.gram(G, bvec) as p:
                p.set_prox(L.NormL1(1e-3))
                p.admm_init(None, μ=1.0, tol=0.0)
                if storage == "mixed":                            # a diagonally dominant inverse: every off-diagonal tile is fixed point
                    saved = 6 * 2048 * (2048 + 128) // 2 - p.time_matvec(1)[1]       # (diagonal tiles may be too: 98304 -> 75264 B)
                    assert 120 * (98304 - 74240) <= saved <= 120 * (98304 - 74240) + 16 * (98304 - 75264), saved
                p.admm_run(1)
                outs[storage] = p.admm_get()[0]                  # x after the first iteration = M b
        finally:
            os.environ.pop("LPVS_M_STORAGE", None)
    Minv = np.linalg.inv(G + np.eye(n))
    xe = Minv @ bvec
    assert rel(outs["f64"], xe) <= 1e-11
    assert rel(outs["split"], xe) <= 1e-11 and rel(outs["split"], outs["f64"]) <= 3e-12
    assert rel(outs["mixed"], xe) <= 1e-11 and rel(outs["mixed"], outs["f64"]) <= 3e-12


def test_split_vs_f64_at_cfg3_fullsize(L):
    """The judged size: what the 40-bit storage costs in the iterates after 2000 iterations (measured, printed)."""
    import bench, torch
    y, X, V, w = bench.synth_signal(1 << 20, 512, 0, torch.device("cuda"))
    prox = L.SlicedSeparableSum.frequency_groups(5.0, 512, 16)
    out = {}
    for st in ("mixed", "split", "f64"):
        os.environ["LPVS_M_STORAGE"] = st
        try:
            with L.Problem.lpv(y, X, V, w, 8) as p:
                p.set_prox(prox)
                p.admm_init(None, μ=0.05, tol=0.0)
                us, nb = p.time_matvec(300)
                p.admm_run(2000)
                out[st] = (p.admm_get()[1], us, nb)
        finally:
            os.environ.pop("LPVS_M_STORAGE", None)
    for st in ("mixed", "split"):
        r = rel(out[st][0], out["f64"][0])
        print(f"cfg3 N=2^20 2000 iterations: rel-L2(z {st} vs f64) = {r:.3e}; mat-vec {out[st][1]:.2f} us ({out[st][2] * 1e-6:.1f} MB, {out[st][2] / out[st][1] * 1e-6:.0f} GB/s) "
              f"vs {out['f64'][1]:.2f} us ({out['f64'][2] / out['f64'][1] * 1e-6:.0f} GB/s)")
        assert np.array_equal(out[st][0] != 0, out["f64"][0] != 0)
        assert r <= 1e-9, r                                      # measured 1.2e-10 (split, offset form; 5.3e-9 without it)
