"""Slot sums of the structured Gram by non-uniform FFT (csrc/nufft.hip, the default when the slot frequencies are multiples of one
step and N >= 4096) against their direct evaluation (csrc/nudft.hip, LPVS_NUDFT=direct) and against the oracle.  GPU only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gram(L, y, X, V, w, Nv, mode, monkeypatch):
    if mode:
        monkeypatch.setenv("LPVS_NUDFT", mode)
    else:
        monkeypatch.delenv("LPVS_NUDFT", raising=False)
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        G, b = p.get_gram()
        form = p.timing()["gram_form"]
    return G, b, form


@pytest.mark.parametrize("N,Nf,Nv,a_over_D,shuffle", [(4096, 24, 3, 1.0, False),     # w = D (1..Nf): s0 = 2, right-hand sides by NUFFT too
                                                      (20000, 130, 4, 0.0, True),      # default_freqs-like grid from zero, unsorted samples
                                                      (70001, 300, 2, 1.5, False),     # s0 = 3 (odd): right-hand sides stay direct; ragged chunk
                                                      (33000, 64, 8, 2.0, True),
                                                      (9000, 1000, 2, 1.0, False)])    # fine grid of 8192 cells: two grids per workgroup, half twiddle table
def test_nufft_gram_equals_direct_evaluation(L, N, Nf, Nv, a_over_D, shuffle, monkeypatch):
    rng = np.random.default_rng(N)
    X = rng.random(N) * 400.0 - 100.0                    # negative abscissae too
    if not shuffle:
        X = np.sort(X)
    V = rng.random(N) * 2 - 0.5
    D = 0.731
    w = D * (a_over_D + np.arange(Nf))
    y = rng.standard_normal(N)
    Gd, bd, fd = _gram(L, y, X, V, w, Nv, "direct", monkeypatch)
    Gn, bn, fn = _gram(L, y, X, V, w, Nv, None, monkeypatch)
    assert fd == "ap" and fn == "ap-nufft"
    assert np.abs(Gn - Gd).max() <= 1e-12 * np.abs(Gd).max(), np.abs(Gn - Gd).max() / np.abs(Gd).max()
    assert np.abs(bn - bd).max() <= 1e-12 * np.abs(bd).max(), np.abs(bn - bd).max() / np.abs(bd).max()
    assert np.array_equal(Gn, Gn.T)
    # fixed-point accumulation: no dependence on the order the hardware serves the scatter in
    G2, b2, _ = _gram(L, y, X, V, w, Nv, None, monkeypatch)
    assert np.array_equal(Gn, G2) and np.array_equal(bn, b2)


def test_nufft_gram_against_the_oracle_and_multi_signal_rhs(L, oracle, monkeypatch):
    rng = np.random.default_rng(3)
    N, Nf, Nv, ns = 6000, 20, 3, 3
    X = np.sort(rng.random(N) * 60.0); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) / 8
    Y = rng.standard_normal((N, ns))
    monkeypatch.delenv("LPVS_NUDFT", raising=False)
    with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
        G, _ = p.get_gram()
        b = p.get_rhs()
        assert p.timing()["gram_form"] == "ap-nufft"
    Phi = oracle.lpv_regressor(X, V, w, Nv)
    tol = 1e-12 + 4.5e-16 * np.abs(w).max() * np.abs(X).max()
    for q in range(ns):
        Go, bo = oracle.gram(Phi, Y[:, q])
        assert np.abs(G - Go).max() <= tol * np.abs(Go).max()
        assert np.abs(b[:, q] - bo).max() <= 10 * tol * max(np.abs(bo).max(), 1.0)


def test_small_problems_and_split_slot_layouts_keep_the_direct_sums(L, monkeypatch):
    monkeypatch.delenv("LPVS_NUDFT", raising=False)
    rng = np.random.default_rng(4)
    for N, w in ((1000, 0.5 * (1 + np.arange(12.0))),                 # too few samples for the grid to pay
                 (8000, 0.37 + 0.5 * np.arange(12.0))):                # 2a is not a multiple of D: two slot families
        X = np.sort(rng.random(N) * 30.0); V = rng.random(N)
        with L.Problem.lpv(rng.standard_normal(N), X, V, w, 3) as p:
            assert p.timing()["gram_form"] == "ap"


@pytest.mark.parametrize("ns,window", [(1, False), (2, True), (3, True)])
def test_window_engine_nufft_equals_direct_sums(L, ns, window, monkeypatch):
    """Batched windows (psd: one signal, csd-like: several signals sharing every window's Gram): slot sums per window by NUFFT
    against the direct sums -- same spectra to 1e-10 after 200 ADMM iterations, same result from a window shard, bit for bit."""
    rng = np.random.default_rng(ns)
    n, nwin, Nf = 4096, 6, 96
    Ltot = n * nwin
    t = np.arange(Ltot) * 0.5 + 0.01 * rng.random(Ltot)                 # jittered sampling
    f = np.arange(Nf) / (2.0 * Nf)
    Y = np.stack([np.sin(2 * np.pi * f[7 + 3 * q] * t) + 0.3 * rng.standard_normal(Ltot) for q in range(ns)], axis=1)
    W = np.hanning(n) + 0.1 if window else None
    from lpvspectral_jl_amd import _lib, api
    eng = dict(estimator=_lib.EST_SPARSE, lam=0.0, prox=(_lib.PROX_L1, 0.3, 0), μ=0.01, tol=0.0, iters=200, sign=_lib.LINEAR_LEAST_SQUARES)
    Ys = [np.ascontiguousarray(Y[:, q]) for q in range(ns)]
    res = {}
    for mode in ("direct", None):
        if mode:
            monkeypatch.setenv("LPVS_NUDFT", mode)
        else:
            monkeypatch.delenv("LPVS_NUDFT", raising=False)
        res[mode] = api.windows_estimate(Ys, t, f, n, 0, W, eng)
        assert api.windowpsd_last_timing()["gram_form"] == ("ap" if mode else "ap-nufft")
    xd, xn = res["direct"][0], res[None][0]
    assert np.abs(xn - xd).max() <= 1e-10 * np.abs(xd).max(), np.abs(xn - xd).max() / np.abs(xd).max()
    part = api.windows_estimate(Ys, t, f, n, 0, W, eng, win_lo=2, win_hi=5)[0]
    assert np.array_equal(part, xn[:, 2:5])


def test_non_finite_inputs_propagate_as_with_direct_sums(L, monkeypatch):
    """A NaN in the signal poisons b (and nothing else); a NaN in a window's samples poisons that window's coefficients only -- with
    the NUFFT exactly as with the direct sums (a fixed-point grid cannot hold a NaN: the library detects it and says so in floating point)."""
    from lpvspectral_jl_amd import _lib, api
    rng = np.random.default_rng(9)
    N, Nf, Nv = 6000, 20, 3
    X = np.sort(rng.random(N) * 60.0); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) / 8
    y = rng.standard_normal(N); y[1234] = np.nan
    monkeypatch.delenv("LPVS_NUDFT", raising=False)
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        G, b = p.get_gram()
        assert p.timing()["gram_form"] == "ap-nufft"
    assert np.isfinite(G).all() and np.isnan(b).all()
    n, nwin, Nfw = 4096, 4, 96
    t = np.arange(n * nwin) * 0.5
    f = np.arange(Nfw) / (2.0 * Nfw)
    yw = np.sin(2 * np.pi * f[9] * t) + 0.1 * rng.standard_normal(n * nwin)
    yw[n + 17] = np.inf                                                  # window 1
    eng = dict(estimator=_lib.EST_SPARSE, lam=0.0, prox=(_lib.PROX_L1, 0.3, 0), μ=0.01, tol=0.0, iters=20, sign=_lib.LINEAR_LEAST_SQUARES)
    res = {}
    for mode in ("direct", None):
        if mode:
            monkeypatch.setenv("LPVS_NUDFT", mode)
        else:
            monkeypatch.delenv("LPVS_NUDFT", raising=False)
        res[mode] = api.windows_estimate([yw], t, f, n, 0, None, eng)[0][0]
    for x in res.values():
        assert not np.isfinite(x[1]).any() and np.isfinite(x[[0, 2, 3]]).all()
