"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs, at sizes the oracle finishes in seconds.

Tolerances (fp64 path):
  * regressor / basis tables: |dev - oracle| <= 4e-16 * scale (device sincos/exp vs glibc: <= 2 ulp)
  * Gram G, b: relative 1e-12 of max|G| (different summation order over N)
  * ADMM iterates at equal iteration count vs the Gram-form oracle: rel-L2 <= 1e-9, identical support,
    identical iteration count at tol > 0; vs the faithful CG oracle: rel-L2 <= 1e-6
  * index / window bookkeeping, iteration counts at tol = 0: bit-exact
"""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def lpv_signal(N, seed, xmax=10.0):
    rng = np.random.default_rng(seed)
    X = np.sort(xmax * rng.random(N))
    V = np.linspace(0, 1, N)
    fd = [lambda v: 2 * v ** 2, lambda v: 2 / (5 * v + 1), lambda v: 3 * np.exp(-10 * (v - 0.5) ** 2)]
    w = 2 * np.pi * np.array([2.0, 10.0, 20.0])
    dep = np.stack([fd[i](V) for i in range(3)], 1)
    Y = (dep * np.cos(w[None, :] * X[:, None] - 0.5 * dep)).sum(1) + 0.1 * rng.standard_normal(N)
    return Y, X, V


def sines(N, seed, nonuniform=True):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.random(N)) * N if nonuniform else np.arange(N, dtype=float)
    f = np.arange(1, 65) / 128.0
    y = sum(a * np.sin(2 * np.pi * f[i] * t + p) for a, i, p in [(2, 5, 0.3), (1, 20, 1.1), (0.5, 40, 2.0)])
    return y + 0.1 * rng.standard_normal(N), t, f


# ------------------------------------------------------------------ a2: Fourier regressor
@pytest.mark.parametrize("zero", [False, True])
def test_fourier_regressor(L, oracle, zero):
    rng = np.random.default_rng(0)
    t = np.sort(rng.random(777)) * 3e5           # phases up to ~1e6 rad: exercises the large-argument path
    f = np.arange(0 if zero else 1, 40) / 80.0
    A, zf = L.get_fourier_regressor(t, f)
    Ao, zo = oracle.get_fourier_regressor(t, f)
    assert zf == zo and A.shape == Ao.shape
    assert np.abs(A - Ao).max() <= 4e-16 / np.sqrt(2 * len(f)) * 4


def test_fourier_regressor_device_resident_inputs(L, oracle):
    import torch
    t = np.arange(500) * 0.1
    f = L.default_freqs(t)
    A, zf = L.get_fourier_regressor(torch.tensor(t, device="cuda"), torch.tensor(f, device="cuda"))
    Ao, _ = oracle.get_fourier_regressor(t, f)
    assert A.shape == (500, 2 * len(f) - 1) and np.abs(A - Ao).max() < 1e-15


def test_zero_frequency_must_be_first(L):
    with pytest.raises(ValueError):
        L.get_fourier_regressor(np.arange(10.0), np.array([1.0, 0.0, 2.0]))
    with pytest.raises(ValueError):
        L.ls_sparse_spectral(np.ones(10), np.arange(10.0), np.array([1.0, 0.0, 2.0]), iters=1)


# ------------------------------------------------------------------ a3-a5: LPV basis + regressor
@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("coulomb", [False, True])
def test_basis_activation(L, oracle, normalize, coulomb):
    rng = np.random.default_rng(2)
    V = rng.standard_normal(333)
    K = L.basis_activation_func(V, 7, normalize, coulomb)
    Ko = oracle.basis_activation(V, 7, normalize, coulomb)
    assert K.shape == Ko.shape and np.abs(K - Ko).max() <= 1e-15


@pytest.mark.parametrize("permuted", [True, False])
@pytest.mark.parametrize("Nv", [3, 8])
def test_lpv_regressor(L, oracle, permuted, Nv):
    Y, X, V = lpv_signal(201, 4, xmax=3e4)
    w = 2 * np.pi * np.arange(2, 16, 2.0)
    Phi = L.lpv_regressor(X, V, w, Nv, True, False, permuted)
    Po = oracle.lpv_regressor(X, V, w, Nv, True, False, permuted)
    assert Phi.shape == Po.shape and np.abs(Phi - Po).max() <= 2e-15


# ------------------------------------------------------------------ Gram (a6/a8 setup)
@pytest.mark.parametrize("N,Nf,Nv", [(500, 12, 50), (1000, 40, 8), (4099, 24, 8), (300, 7, 3), (2500, 70, 2)])
def test_gram_lpv(L, oracle, N, Nf, Nv):
    Y, X, V = lpv_signal(N, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf
    with L.Problem.lpv(Y, X, V, w, Nv) as p:
        G, b = p.get_gram()
    Phi = oracle.lpv_regressor(X, V, w, Nv)
    Go, bo = Phi.T @ Phi, Phi.T @ Y
    assert G.shape == Go.shape
    assert np.abs(G - Go).max() <= 1e-12 * np.abs(Go).max()
    assert np.abs(b - bo).max() <= 1e-12 * np.abs(bo).max()
    assert np.array_equal(G, G.T)


@pytest.mark.parametrize("N,zero,weighted", [(1000, True, False), (1000, False, True), (3001, True, True), (130, False, False)])
def test_gram_fourier(L, oracle, N, zero, weighted):
    y, t, f = sines(N, N)
    if zero:
        f = np.concatenate([[0.0], f])
    W = np.random.default_rng(9).random(N) + 0.5 if weighted else None
    with L.Problem.fourier(y, t, f, W) as p:
        G, b = p.get_gram()
    A, _ = oracle.get_fourier_regressor(t, f)
    Go = A.T @ ((W[:, None] if weighted else 1.0) * A)
    bo = A.T @ ((W if weighted else 1.0) * y)
    assert np.abs(G - Go).max() <= 1e-12 * np.abs(Go).max()
    assert np.abs(b - bo).max() <= 1e-12 * np.abs(bo).max()


def test_gram_dense_and_explicit(L):
    rng = np.random.default_rng(3)
    A = rng.standard_normal((700, 150)); y = rng.standard_normal(700)
    with L.Problem.dense(A, y) as p:
        G, b = p.get_gram()
    assert np.abs(G - A.T @ A).max() <= 1e-12 * np.abs(A.T @ A).max()
    assert np.abs(b - A.T @ y).max() <= 1e-12 * np.abs(A.T @ y).max()
    with L.Problem.gram(A.T @ A, A.T @ y) as p:
        G2, b2 = p.get_gram()
    assert np.array_equal(G2, A.T @ A) and np.array_equal(b2, A.T @ y)


# ------------------------------------------------------------------ x-update solver
@pytest.mark.parametrize("n", [64, 200, 513])
def test_ridge_solve_matches_numpy(L, n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((3 * n, n)) * rng.random(n)[None, :]
    y = rng.standard_normal(3 * n)
    G, b = A.T @ A, A.T @ y
    with L.Problem.gram(G, b) as p:
        x = p.solve_ridge(20.0)
    xo = np.linalg.solve(G + 20.0 * np.eye(n), b)
    assert rel(x, xo) <= 1e-11


def test_not_positive_definite_is_reported(L):
    G = -np.eye(8)
    with L.Problem.gram(G, np.ones(8)) as p:
        with pytest.raises(L.NumericError):
            p.solve_ridge(0.5)


# ------------------------------------------------------------------ ADMM (a6-a13)
def _device_admm(L, prob, proxg, sign=1, **kw):
    prob.set_prox(proxg)
    prob.admm_init(None, μ=kw.get("μ", 0.05), tol=kw.get("tol", 1e-5), linear_sign=sign)
    it, nxz, conv = prob.admm_run(kw.get("iters", 100))
    x, z, u = prob.admm_get()
    return dict(x=x, z=z, u=u, iters=it, nxz=nxz, conv=conv)


PROXES = [("l1", 1.0), ("l0", 5.0), ("ball", 6), ("group", 2.0)]


@pytest.mark.parametrize("kind,param", PROXES)
def test_admm_fourier_matches_gram_oracle(L, oracle, kind, param):
    y, t, f = sines(600, 11)
    A, _ = oracle.get_fourier_regressor(t, f)
    Go, bo = oracle.gram(A, y)
    mk = {"l1": (L.NormL1, oracle.NormL1), "l0": (L.NormL0, oracle.NormL0), "ball": (L.IndBallL0, oracle.IndBallL0)}
    if kind == "group":
        gd = L.SlicedSeparableSum.frequency_groups(param, len(f), 2)
        go = oracle.GroupL2(param, 2)
    else:
        gd, go = mk[kind][0](param), mk[kind][1](param)
    ro = oracle.admm_gram(Go, bo, go, iters=60, tol=0.0, mu=0.05, history=True)
    with L.Problem.fourier(y, t, f) as p:
        rd = _device_admm(L, p, gd, iters=60, tol=0.0, μ=0.05)
    assert rd["iters"] == ro["iters"] == 60 and not rd["conv"]
    assert rel(rd["z"], ro["z"]) <= 1e-9 and rel(rd["x"], ro["x"]) <= 1e-9 and rel(rd["u"], ro["u"]) <= 1e-9
    assert np.array_equal(rd["z"] != 0, ro["z"] != 0)
    assert abs(rd["nxz"] - ro["nxz"][-1]) <= 1e-7 * ro["nxz"][-1] + 1e-12 * np.linalg.norm(ro["x"])
    assert 0 < np.count_nonzero(ro["z"]) <= 12


def test_admm_stops_at_reference_iteration(L, oracle):
    y, t, f = sines(500, 12)
    A, _ = oracle.get_fourier_regressor(t, f)
    Go, bo = oracle.gram(A, y)
    ro = oracle.admm_gram(Go, bo, oracle.NormL1(1.0), iters=5000, tol=1e-9, mu=0.05, history=True)
    assert 50 < ro["iters"] < 5000
    with L.Problem.fourier(y, t, f) as p:
        p.set_prox(L.NormL1(1.0)); p.admm_init(None, μ=0.05, tol=1e-9)
        it1, _, c1 = p.admm_run(7)                     # chunked like the host wrapper
        it, nxz, conv = p.admm_run(5000)
        x, z, u = p.admm_get()
        it2, _, _ = p.admm_run(50)                     # no-op after convergence
    assert it1 == 7 and not c1
    assert conv and it == ro["iters"] and it2 == it
    assert rel(z, ro["z"]) <= 1e-9


def test_admm_vs_faithful_cg_oracle_lpv(L, oracle):
    """Headline path at test size (the reference's own test problem, test/test_lasso.jl:17-32)."""
    Y, X, V = lpv_signal(500, 0)
    w_test = 2 * np.pi * np.arange(2, 26, 2.0)
    Nv = 50
    po, ro = oracle.ls_sparse_spectral_lpv(Y, X, V, w_test, Nv, lam=5, tol=1e-8, iters=2000)
    buf = io.StringIO()
    import lpvspectral_jl_amd.api as api
    with L.Problem.lpv(Y, X, V, w_test, Nv) as p:
        g = L.SlicedSeparableSum.frequency_groups(5, len(w_test), 2 * Nv)
        x, z = api._admm_on_problem(p, None, g, 1, iters=2000, tol=1e-8, printerval=100, μ=0.05, out=buf)
        params = p.params(0)
    lines = buf.getvalue().strip().splitlines()
    assert lines[0].startswith("100 ||x-z||₂ ")
    assert lines[-1].startswith("%d ||x-z||₂ " % ro["iters"])         # same stopping iteration
    assert rel(z, ro["z"]) <= 1e-6
    assert rel(params, po) <= 1e-6
    assert np.array_equal(z != 0, ro["z"] != 0)
    assert set(np.argsort(-oracle.psd(params, len(w_test)))[:3] + 1) == {1, 5, 10}


def test_ls_sparse_spectral_lpv_api(L, oracle):
    Y, X, V = lpv_signal(400, 7)
    w = 2 * np.pi * np.arange(2, 26, 2.0)
    se = L.ls_sparse_spectral_lpv(Y, X, V, w, 8, λ=5, iters=300, tol=0, printerval=1000)
    po, ro = oracle.ls_sparse_spectral_lpv(Y, X, V, w, 8, lam=5, iters=300, tol=0)
    assert isinstance(se, L.SpectralExt) and se.Σ is None and se.x.shape == (len(w) * 8,)
    assert rel(se.x, po) <= 1e-6
    G, bo = oracle.gram(oracle.lpv_regressor(X, V, w, 8), Y)
    rg = oracle.admm_gram(G, bo, oracle.GroupL2(5, 16), iters=300, tol=0)
    assert rel(se.x, oracle.lpv_unpermute(rg["z"], len(w), 8)) <= 1e-9
    assert np.allclose(L.psd(se).ravel(), oracle.psd(se.x, len(w)))
    with pytest.raises(NotImplementedError):
        L.ls_sparse_spectral_lpv(Y, X, V, w, 8, coulomb=True)
    with pytest.raises(AssertionError):
        L.ls_sparse_spectral_lpv(Y, X, V, w, 8, μ=1.5)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("zero", [False, True])
def test_ls_sparse_spectral_api(L, oracle, weighted, zero):
    y, t, f = sines(700, 21)
    if zero:
        f = np.concatenate([[0.0], f]); y = y + 0.7
    W = np.ones(700) if weighted else None
    x, fr = L.ls_sparse_spectral(y, t, f, W, λ=1.0, iters=100, tol=0, μ=0.05, printerval=1000)
    xo, _, ro = oracle.ls_sparse_spectral(y, t, f, W, lam=1.0, iters=100, tol=0, mu=0.05)
    assert x.shape == (len(f),) and np.array_equal(fr, f)
    assert rel(x, xo) <= 1e-6
    assert np.array_equal(x != 0, xo != 0) and 3 <= np.count_nonzero(x) <= 8
    if zero:
        assert x[0].imag == 0


def test_admm_plugin_boundary_generic_objects(L, oracle):
    """ADMM(x, proxf, proxg) on caller-supplied LeastSquares / Quadratic objects (src/lasso.jl:136)."""
    rng = np.random.default_rng(5)
    A = rng.standard_normal((400, 90)); xs = np.zeros(90); xs[[3, 40, 77]] = [2, -1, 3]
    y = A @ xs + 0.01 * rng.standard_normal(400)
    x, z = L.ADMM(np.zeros(90), L.LeastSquares(A, y, iterative=True), L.NormL1(5.0), iters=500, tol=1e-9, printerval=10000)
    ro = oracle.admm_ls(A, y, oracle.NormL1(5.0), iters=500, tol=1e-9)
    assert rel(z, ro["z"]) <= 1e-6 and set(np.nonzero(z)[0]) == {3, 40, 77}
    Q, q = A.T @ A, -(A.T @ y)
    x2, z2 = L.ADMM(np.zeros(90), L.Quadratic(Q, q, iterative=True), L.NormL1(5.0), iters=500, tol=1e-9, printerval=10000)
    assert rel(z2, z) <= 1e-9
    with pytest.raises(NotImplementedError):
        L.ADMM(np.zeros(90), L.LeastSquares(A, y), object(), iters=1)


def test_callback_and_printing(L):
    y, t, f = sines(300, 2)
    calls = []
    L.ls_sparse_spectral(y, t, f, λ=0.05, iters=250, tol=0, printerval=100, cb=lambda x, z: calls.append((x.copy(), z.copy())))
    assert len(calls) == 2 and calls[0][0].shape == (2 * len(f),)


# ------------------------------------------------------------------ dense estimators + windows (a16, a18)
def test_ls_spectral_known_answers_on_device(L, oracle):
    """test/runtests.jl:185-195 through the device path: the default grid (501 frequencies, Nreg = 1001 > N = 1000,
    vanishing Nyquist sine column) goes through the dual form, the weighted method through the primal one."""
    t = np.arange(1000) * 0.1
    y = np.sin(2 * np.pi * t)
    x, freqs = L.ls_spectral(y, t)
    p = np.abs(x) ** 2
    assert len(freqs) == 501 and abs(p.max() - 2.0 * len(freqs)) < 1e-4 and p.argmax() + 1 == 101
    xo, _ = oracle.ls_spectral(y, t)
    assert rel(x, xo) <= 1e-6                           # same minimiser as the SVD solve
    f = L.default_freqs(t)
    x, _ = L.ls_spectral(y, t, f, np.ones(len(y)))
    p = np.abs(x) ** 2
    assert abs(p.max() - 2.0 * len(f)) < 1e-4 and p.argmax() + 1 == 101
    f2 = f[:-1]                                          # drop Nyquist: tall, full-rank system (primal form)
    x2, _ = L.ls_spectral(y, t, f2)
    p2 = np.abs(x2) ** 2
    assert abs(p2.max() - 2.0 * len(f2)) < 1e-4 and p2.argmax() + 1 == 101
    xo2, _ = oracle.ls_spectral(y, t, f2)
    assert rel(x2, xo2) <= 1e-8


def test_ls_windowcsd_and_cohere_known_answers(L):
    """test/runtests.jl:203-208."""
    t = np.arange(1000) * 0.1
    y = np.sin(2 * np.pi * t)
    x, freqs = L.ls_windowcsd(y, y, t, noverlap=0)
    a = np.abs(x)
    assert abs(a.max() - 2.0 * len(freqs)) < 1e-4 and a.argmax() + 1 == 11
    c, _ = L.ls_cohere(y, y, t)
    assert np.all(c == 1)


def test_ls_windowpsd_known_answers_on_device(L, oracle):
    t = np.arange(1000) * 0.1
    y = np.sin(2 * np.pi * t)
    S, fr = L.ls_windowpsd(y, t, noverlap=0)
    assert S.argmax() + 1 == 13 and len(fr) == 63
    S16, _ = L.ls_windowpsd(y, t, nw=16, noverlap=0)
    assert np.abs(S16).argmax() + 1 == 7
    So, _ = oracle.ls_windowpsd(y, t, noverlap=0)
    assert rel(S, So) <= 1e-6


def test_ls_windowpsd_sparse_estimator(L, oracle):
    Y, X, V = lpv_signal(500, 0)
    fr = np.arange(1, 22.01, 0.5)
    S, _ = L.ls_windowpsd(Y, X, fr, nw=2, estimator=L.ls_sparse_spectral, λ=0.2, tol=1e-10, printerval=10000, iters=3000, μ=0.0001)
    So, _ = oracle.ls_windowpsd(Y, X, fr, nw=2, estimator=lambda y, t, f, W, **k: oracle.ls_sparse_spectral(y, t, f, W, **k),
                                lam=0.2, tol=1e-10, iters=3000, mu=0.0001)
    assert rel(S, So) <= 1e-6


def test_ls_spectral_lpv_top3(L, oracle):
    Y, X, V = lpv_signal(500, 0)
    w_test = 2 * np.pi * np.arange(2, 26, 2.0)
    se = L.ls_spectral_lpv(Y, X, V, w_test, 50, λ=0.02)
    assert set(np.argsort(-L.psd(se).ravel())[:3] + 1) == {1, 5, 10}
    xo = oracle.ls_spectral_lpv(Y, X, V, w_test, 50, lam=0.02)
    assert rel(se.x, xo) <= 1e-6
    Sw = L.ls_windowpsd_lpv(Y, X, V, w_test, 50, λ=0.02)
    assert set(np.argsort(-Sw)[:3] + 1) == {1, 5, 10}
    # its windows two in flight (the default) or one after the other: the same solves, summed in window order; the call above went through
    # the library's own driver (lpvs_windowpsd_lpv_f64), with `covariance=False` spelled out it takes the wrapper's per-window loop
    assert np.array_equal(Sw, L.ls_windowpsd_lpv(Y, X, V, w_test, 50, λ=0.02, in_flight=1))
    Sww = L.ls_windowpsd_lpv(Y, X, V, w_test, 50, λ=0.02, covariance=False)
    assert rel(Sw, Sww) <= 1e-9, rel(Sw, Sww)                    # (the batched refinement and the single-handle one sum in different orders)
    So = sum(np.abs(oracle.ls_spectral_lpv(Y[i * 50:(i + 1) * 50], X[i * 50:(i + 1) * 50], V[i * 50:(i + 1) * 50], w_test, 50, lam=0.02).reshape(-1, len(w_test)).sum(axis=0)) ** 2
             for i in range(10))
    assert rel(Sw, So) <= 1e-6
    # a well-posed case (125 samples per window, 96 unknowns): the library's driver with 1, 2 and 5 windows in flight, the wrapper's
    # per-window loop and the oracle
    S4 = L.ls_windowpsd_lpv(Y, X, V, w_test, 4, 4, λ=1e-4, in_flight=2)
    assert np.array_equal(S4, L.ls_windowpsd_lpv(Y, X, V, w_test, 4, 4, λ=1e-4, in_flight=1))
    assert np.array_equal(S4, L.ls_windowpsd_lpv(Y, X, V, w_test, 4, 4, λ=1e-4, in_flight=5))
    S4w = L.ls_windowpsd_lpv(Y, X, V, w_test, 4, 4, λ=1e-4, covariance=False)      # the wrapper's per-window loop (single-handle solves)
    assert rel(S4, S4w) <= 1e-10, rel(S4, S4w)                                      # (batched and single-handle refinement sum in different orders)
    So4 = sum(np.abs(oracle.ls_spectral_lpv(Y[i * 125:(i + 1) * 125], X[i * 125:(i + 1) * 125], V[i * 125:(i + 1) * 125], w_test, 4, lam=1e-4).reshape(-1, len(w_test)).sum(axis=0)) ** 2
              for i in range(4))
    assert rel(S4, So4) <= 1e-6
    with L.default_options(gram_form="krs"):                     # option defaults reach the worker threads
        Sk = L.ls_windowpsd_lpv(Y, X, V, w_test, 50, λ=0.02, in_flight=3)
    assert rel(Sk, Sw) <= 1e-8
    # covariance (src/lsfft.jl:252-254) against a direct numpy evaluation in the reference's [re; im] order
    Ar = oracle.lpv_regressor(X, V, w_test, 50, permuted=False)
    xr = np.concatenate([se.x.real, se.x.imag])
    e = Ar @ xr - Y
    Sig = np.var(e, ddof=1) * np.linalg.inv(Ar.T @ Ar + 0.02 * np.eye(Ar.shape[1]))
    assert se.Σ.shape == Sig.shape and np.abs(se.Σ - Sig).max() <= 1e-6 * np.abs(Sig).max()


def test_ls_windowpsd_lpv_batch_of_64_windows(L, oracle):
    """src/lsfft.jl:267-277 through the batch machinery (lpvs_windowpsd_lpv_f64: the windows' Grams into one batch, ONE blocked sweep for
    all factorisations, batched refined ridge solves): 64 windows of 600 samples, 96 unknowns each, against the oracle's
    ls_spectral_lpv per window, S summed in window order; the per-window fraction of variance explained against a direct numpy
    evaluation, and the reference's warning (:255-256) for windows below 0.9."""
    import ctypes as C
    import warnings
    from lpvspectral_jl_amd._lib import lib, out_ptr, check
    nw, nwin_len, Nv = 64, 600, 4
    Y, X, V = lpv_signal(nw * nwin_len, 3, xmax=10.0 * nw)                          # (every window spans 10 units of X, as the 500-sample record of the reference's test)
    w_test = 2 * np.pi * np.arange(2, 26, 2.0)
    Nf = len(w_test)
    S = np.zeros(Nf); fva = np.zeros(nw)
    Yh, Xh, Vh = (np.ascontiguousarray(a, dtype=np.float64) for a in (Y, X, V))
    check(lib().lpvs_windowpsd_lpv_f64(out_ptr(Yh), out_ptr(Xh), out_ptr(Vh), len(Yh), out_ptr(w_test), Nf, Nv, nwin_len, 0, 0.02, 1, 0, 0, 3, out_ptr(S), out_ptr(fva)))
    So = np.zeros(Nf); fo = np.zeros(nw)
    for i in range(nw):
        sl = slice(i * nwin_len, (i + 1) * nwin_len)
        xo = oracle.ls_spectral_lpv(Y[sl], X[sl], V[sl], w_test, Nv, lam=0.02)
        So += np.abs(xo.reshape(-1, Nf).sum(axis=0)) ** 2                        # abs2.(sum(reshape_params(x, Nf), dims = 2)), window order
        Ar = oracle.lpv_regressor(X[sl], V[sl], w_test, Nv, permuted=False)
        e = Ar @ np.concatenate([xo.real, xo.imag]) - Y[sl]
        fo[i] = 1 - np.var(e, ddof=1) / np.var(Y[sl], ddof=1)
    assert rel(S, So) <= 1e-8, rel(S, So)
    assert np.abs(fva - fo).max() <= 1e-8, np.abs(fva - fo).max()
    # the wrapper: same S whatever in_flight, and one warning per window below 0.9
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        Sw = L.ls_windowpsd_lpv(Y, X, V, w_test, Nv, nw, 0, λ=0.02)
        S1 = L.ls_windowpsd_lpv(Y, X, V, w_test, Nv, nw, 0, λ=0.02, in_flight=1)
    assert np.array_equal(Sw, S) and np.array_equal(S1, S)
    nlow = int((fo < 0.9).sum())
    assert sum("Fraction of variance explained" in str(r.message) for r in rec) == 2 * nlow
    # overlapping windows, ragged tail dropped (Windows3 / arraysplit)
    S2 = np.zeros(Nf)
    k = C.c_int64(0)
    check(lib().lpvs_window_count(len(Yh), 700, 350, C.byref(k)))
    check(lib().lpvs_windowpsd_lpv_f64(out_ptr(Yh), out_ptr(Xh), out_ptr(Vh), len(Yh), out_ptr(w_test), Nf, Nv, 700, 350, 0.02, 1, 0, 0, 2, out_ptr(S2), None))
    So2 = np.zeros(Nf)
    for i in range(int(k.value)):
        sl = slice(i * 350, i * 350 + 700)
        So2 += np.abs(oracle.ls_spectral_lpv(Y[sl], X[sl], V[sl], w_test, Nv, lam=0.02).reshape(-1, Nf).sum(axis=0)) ** 2
    assert int(k.value) == (len(Yh) - 700) // 350 + 1 and rel(S2, So2) <= 1e-8


# ------------------------------------------------------------------ batched windows (cfg4 engine)
@pytest.mark.parametrize("noverlap,zero", [(0, True), (100, False)])
def test_windowpsd_batched_equals_sequential_and_oracle(L, oracle, noverlap, zero):
    rng = np.random.default_rng(8)
    Lh, n = 4000, 500
    t = np.cumsum(0.5 + rng.random(Lh))                       # non-equidistant
    f = (np.arange(0 if zero else 1, 40)) / 100.0
    y = 1.5 * np.sin(2 * np.pi * 0.11 * t) + 0.7 * np.cos(2 * np.pi * 0.29 * t + 0.4) + 0.1 * rng.standard_normal(Lh) + (0.5 if zero else 0)
    W = L.hanning(n)
    kw = dict(λ=0.5, μ=0.05, tol=1e-9, iters=3000)
    xb, Sb, its = L.windowpsd_sparse_batched(y, t, f, n, noverlap, W, **kw)
    os.environ["LPVS_M_STORAGE"] = "f64"
    try:
        xb8, Sb8, its8 = L.windowpsd_sparse_batched(y, t, f, n, noverlap, W, **kw)
    finally:
        del os.environ["LPVS_M_STORAGE"]
    assert np.array_equal(its8, its)
    k = len(L.Windows2(y, t, n, noverlap))
    assert xb.shape == (k, len(f)) and its.shape == (k,)
    S_seq = np.zeros(len(f))
    for i, (yi, ti) in enumerate(L.Windows2(y, t, n, noverlap)):
        xs, _ = L.ls_sparse_spectral(yi, ti, f, W, printerval=100000, **kw)       # sequential device path
        xo, _, ro = oracle.ls_sparse_spectral(yi, ti, f, W, lam=0.5, mu=0.05, tol=1e-9, iters=3000)
        assert rel(xb[i], xs) <= 1e-9 and rel(xb[i], xo) <= 1e-6    # batch: 6-byte copy of the inverse (offset form); handle path: doubles
        assert rel(xb8[i], xs) <= 1e-12                             # with doubles in the batch too only the summation order differs
        assert its[i] == ro["iters"] and np.array_equal(xb[i] != 0, xo != 0)
        S_seq += np.abs(xs) ** 2
    assert rel(Sb, S_seq) <= 1e-9 and rel(Sb8, S_seq) <= 1e-12
    # the drop-in driver uses the batch and divides by k^2 (src/lsfft.jl:125)
    S1, _ = L.ls_windowpsd(y, t, f, nw=Lh // n, noverlap=noverlap, window_func=L.hanning, estimator=L.ls_sparse_spectral, **kw)
    S2, _ = L.ls_windowpsd(y, t, f, nw=Lh // n, noverlap=noverlap, window_func=L.hanning, estimator=L.ls_sparse_spectral, batched=False,
                           printerval=100000, **kw)
    assert rel(S1, Sb / k ** 2) <= 1e-14 and rel(S1, S2) <= 1e-9
    # sharding: two disjoint window ranges reproduce the whole, bit for bit
    xa, Sa, _ = L.windowpsd_sparse_batched(y, t, f, n, noverlap, W, win_lo=0, win_hi=k // 2, **kw)
    xc, Sc, _ = L.windowpsd_sparse_batched(y, t, f, n, noverlap, W, win_lo=k // 2, win_hi=k, **kw)
    assert np.array_equal(np.vstack([xa, xc]), xb)


# ------------------------------------------------------------------ shared-regressor batch (cfg5 engine)
@pytest.mark.parametrize("Nf,Nv,prox", [(12, 8, "group"), (140, 8, "group"), (140, 8, "ball"), (12, 4, "l1")])
def test_lpv_multi_equals_single_signal_runs(L, Nf, Nv, prox):
    """ns signals sharing (X, V, w): one Gram, ns right-hand sides; every column must equal the single-signal solve, with
    its own stopping iteration: bit for bit on the plain mat-vec path (same kernels, same order); on the tile-packed path
    the multi-signal product runs on the matrix cores (different summation order): rel-L2 <= 1e-12, identical support."""
    rng = np.random.default_rng(13)
    N, ns = 1500, 3
    X = np.sort(10 * rng.random(N)); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf
    Y = np.stack([np.cos(w[(3 * q + 1) % Nf] * X) * (1 + q * V) + 0.3 * np.sin(w[(5 * q + 2) % Nf] * X) + 0.05 * rng.standard_normal(N)
                  for q in range(ns)], axis=1)
    n = 2 * Nf * Nv
    g = {"group": None, "ball": L.IndBallL0(6), "l1": L.NormL1(2.0)}[prox]
    kw = dict(λ=3.0, iters=400, tol=1e-7, μ=0.05, printerval=100000)
    ses = L.ls_sparse_spectral_lpv_multi(Y, X, V, w, Nv, proxg=g, **kw)
    assert len(ses) == ns
    ses8 = None
    if n >= 2048:                                # the same with 8-byte storage of the inverse
        os.environ["LPVS_M_STORAGE"] = "f64"
        try:
            ses8 = L.ls_sparse_spectral_lpv_multi(Y, X, V, w, Nv, proxg=g, **kw)
        finally:
            del os.environ["LPVS_M_STORAGE"]
    its = []
    for q in range(ns):
        se = L.ls_sparse_spectral_lpv(Y[:, q].copy(), X, V, w, Nv, proxg=g, **kw)
        if n >= 2048:
            # both paths stream the 6-byte copy of M by default (parity bound 1e-9, measured ~1e-11); with doubles
            # (LPVS_M_STORAGE=f64) only the summation order differs
            same = lambda a, b: np.array_equal(np.abs(a) > 0, np.abs(b) > 0)
            assert rel(ses[q].x, se.x) <= 1e-9 and same(ses[q].x, se.x), (q, rel(ses[q].x, se.x))
            os.environ["LPVS_M_STORAGE"] = "f64"
            try:
                se8 = L.ls_sparse_spectral_lpv(Y[:, q].copy(), X, V, w, Nv, proxg=g, **kw)
            finally:
                del os.environ["LPVS_M_STORAGE"]
            assert rel(ses8[q].x, se8.x) <= 1e-12 and same(ses8[q].x, se8.x), (q, rel(ses8[q].x, se8.x))
            assert rel(ses[q].x, se8.x) <= 1e-9 and same(ses[q].x, se8.x), (q, rel(ses[q].x, se8.x))
        else:
            assert np.array_equal(ses[q].x, se.x), (q, rel(ses[q].x, se.x))
        its.append(np.count_nonzero(se.x))
    assert (n >= 2048) == (Nf == 140)            # both the packed-symmetric and the plain mat-vec paths are covered
    with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
        p.set_prox(L.SlicedSeparableSum.frequency_groups(3.0, Nf, 2 * Nv) if g is None else g)
        p.admm_init(None, μ=0.05, tol=1e-3)
        it, nxz, conv = p.admm_run(5000)
        per = [p.admm_status(q) for q in range(ns)]
    assert conv and all(c for _, _, c in per) and it == max(i for i, _, _ in per)
    assert len({i for i, _, _ in per}) > 1       # signals stop at their own iteration


# ------------------------------------------------------------------ edge cases
def test_tiny_and_ragged_problems(L, oracle):
    """Sizes below one tile / one stage, single frequency, N < n (fat), odd everything."""
    rng = np.random.default_rng(21)
    # Fourier: N = 5 samples, 1 and 3 frequencies (with and without the zero frequency)
    for f in (np.array([0.3]), np.array([0.0, 0.2, 0.45]), np.array([0.1, 0.2, 0.45])):
        t = np.sort(rng.random(5)) * 10; y = rng.standard_normal(5)
        A, zf = L.get_fourier_regressor(t, f)
        Ao, zo = oracle.get_fourier_regressor(t, f)
        assert zf == zo and np.abs(A - Ao).max() < 1e-15
        with L.Problem.fourier(y, t, f) as p:
            G, b = p.get_gram()
        assert np.abs(G - Ao.T @ Ao).max() <= 1e-13 and np.abs(b - Ao.T @ y).max() <= 1e-13
        x, _ = L.ls_sparse_spectral(y, t, f, λ=0.01, iters=50, tol=0, printerval=1000)
        xo, _, _ = oracle.ls_sparse_spectral(y, t, f, lam=0.01, iters=50, tol=0)
        assert rel(x, xo) <= 1e-8
    # LPV: N = 7 < n = 2*3*2 = 12 (fat), Nv = 2
    N = 7
    X = np.sort(rng.random(N)) * 3; V = rng.random(N); Y = rng.standard_normal(N)
    w = np.array([1.0, 2.5, 4.0])
    se = L.ls_sparse_spectral_lpv(Y, X, V, w, 2, λ=0.1, iters=80, tol=0, printerval=1000)
    po, ro = oracle.ls_sparse_spectral_lpv(Y, X, V, w, 2, lam=0.1, iters=80, tol=0)
    assert rel(se.x, po) <= 1e-6
    # group prox with a length that does not divide 128 (n = 2*5*3 = 30, groups of 6)
    N = 300
    X = np.sort(rng.random(N)) * 10; V = np.linspace(0, 1, N)
    w = 2 * np.pi * np.arange(1, 6.0)
    Y = np.cos(w[1] * X) * V + 0.05 * rng.standard_normal(N)
    se = L.ls_sparse_spectral_lpv(Y, X, V, w, 3, λ=2.0, iters=200, tol=0, printerval=1000)
    G, b = oracle.gram(oracle.lpv_regressor(X, V, w, 3), Y)
    rg = oracle.admm_gram(G, b, oracle.GroupL2(2.0, 6), iters=200, tol=0)
    assert rel(se.x, oracle.lpv_unpermute(rg["z"], 5, 3)) <= 1e-9


def test_windows_edge_cases_on_device(L):
    t = np.arange(40.0); y = np.sin(t)
    f = np.array([0.0, 0.1, 0.2])
    x, S, its = L.windowpsd_sparse_batched(y, t, f, 50, 0, None, iters=5)        # L < n: no windows
    assert x.shape == (0, 3) and np.all(S == 0) and len(its) == 0
    x, S, its = L.windowpsd_sparse_batched(y, t, f, 40, 0, None, iters=5, tol=0)  # exactly one window
    assert x.shape == (1, 3) and its[0] == 5
    with pytest.raises(L.DomainError):
        L.windowpsd_sparse_batched(y, t, f, 10, 10, None, iters=5)
    with pytest.raises(ValueError):
        L.windowpsd_sparse_batched(y, t, np.array([0.1, 0.0]), 10, 0, None, iters=5)
    with pytest.raises(AssertionError):
        L.windowpsd_sparse_batched(y, t, f, 10, 0, None, iters=5, μ=2.0)


def test_warm_start_and_reinit(L, oracle):
    """init=true warm start (src/lasso.jl:92-97) and re-initialising a handle with another mu."""
    y, t, f = sines(400, 33)
    x1, _ = L.ls_sparse_spectral(y, t, f, init=True, λ=1.0, iters=40, tol=0, printerval=1000)
    xo, _, _ = oracle.ls_sparse_spectral(y, t, f, init=True, lam=1.0, iters=40, tol=0)
    assert rel(x1, xo) <= 1e-6
    with L.Problem.fourier(y, t, f) as p:
        p.set_prox(L.NormL1(1.0))
        p.admm_init(None, μ=0.05, tol=0); p.admm_run(30); _, z1, _ = p.admm_get()
        p.admm_init(None, μ=0.5, tol=0); p.admm_run(30); _, z2, _ = p.admm_get()      # new factorisation
        p.admm_init(None, μ=0.05, tol=0); p.admm_run(30); _, z3, _ = p.admm_get()     # back: identical to the first
    assert np.array_equal(z1, z3) and not np.array_equal(z1, z2)


def test_ball_prox_with_ties(L, oracle):
    """IndBallL0 when several |v| are exactly equal at the cut: lowest indices win (the reference's
    partialsortperm leaves the order unspecified; the oracle uses the same rule)."""
    n = 64
    G = np.zeros((n, n)); b = np.zeros(n)
    b[[3, 10, 20, 30, 40, 50]] = [5.0, 2.0, -2.0, 2.0, -2.0, 1.0]     # x = mu*b on the first step: four equal magnitudes
    for r in (2, 3, 4):
        with L.Problem.gram(G, b) as p:
            p.set_prox(L.IndBallL0(r)); p.admm_init(None, μ=0.5, tol=0); p.admm_run(1)
            x, z, u = p.admm_get()
        ro = oracle.admm_gram(G, b, oracle.IndBallL0(r), iters=1, tol=0, mu=0.5)
        assert np.array_equal(z != 0, ro["z"] != 0) and np.count_nonzero(z) == r      # same winners: lowest indices
        assert np.allclose(z, ro["z"], rtol=1e-14, atol=0)
        assert list(np.nonzero(z)[0]) == [3, 10, 20, 30][:r]


# ------------------------------------------------------------------ Gram forms
def test_gram_form_selection_and_agreement(L, oracle):
    """auto picks the structured form only for arithmetic-progression w; all forms agree with the oracle Gram."""
    import os
    rng = np.random.default_rng(77)
    N, Nv = 700, 5
    X = np.sort(rng.random(N)) * 50; V = rng.random(N); Y = rng.standard_normal(N)
    w_ap = 2 * np.pi * np.arange(2, 40, 2.0) / 7.0
    w_jit = w_ap * (1 + 1e-9 * rng.standard_normal(len(w_ap)))        # off the progression by far more than rounding
    def gram(w, form=None):
        if form: os.environ["LPVS_GRAM_FORM"] = form
        try:
            with L.Problem.lpv(Y, X, V, w, Nv) as p:
                return p.get_gram()
        finally:
            os.environ.pop("LPVS_GRAM_FORM", None)
    for w in (w_ap, w_jit):
        Phi = oracle.lpv_regressor(X, V, w, Nv)
        Go, bo = Phi.T @ Phi, Phi.T @ Y
        tol = 1e-12 + 4.5e-16 * float(w.max() * X.max())
        for form in (None, "krs", "kr") + (("ap",) if w is w_ap else ()):
            G, b = gram(w, form)
            assert np.abs(G - Go).max() <= tol * np.abs(Go).max(), form
            assert np.abs(b - bo).max() <= tol * np.abs(bo).max() * 10, form
    assert np.array_equal(gram(w_ap)[0], gram(w_ap, "ap")[0])          # auto == structured on the progression
    assert np.array_equal(gram(w_jit)[0], gram(w_jit, "krs")[0])       # auto == dense otherwise
    with pytest.raises(ValueError):
        gram(w_jit, "ap")
    # frequencies that are a progression only up to rounding (as produced by 2*pi*(1:Nf)*c): first-order residual correction
    w_r = 2 * np.pi * (np.arange(40) + 1.0) * (25.0 / 512)
    Xb = np.sort(rng.random(N)) * 2e4
    with L.Problem.lpv(Y, Xb, V, w_r, Nv) as p:
        Ga, _ = p.get_gram()
    os.environ["LPVS_GRAM_FORM"] = "krs"
    try:
        with L.Problem.lpv(Y, Xb, V, w_r, Nv) as p:
            Gk, _ = p.get_gram()
    finally:
        os.environ.pop("LPVS_GRAM_FORM", None)
    assert not np.array_equal(Ga, Gk)
    assert np.abs(Ga - Gk).max() <= (1e-12 + 4.5e-16 * float(w_r.max() * Xb.max())) * np.abs(Gk).max()


# ---- row-sharded Gram + one exchange step ----------------------------------------------------------
def _rowshard_signal(N=6000, Nf=24, seed=5):
    rng = np.random.default_rng(seed)
    X = np.sort(rng.uniform(0, 10 * N / 500, N)); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf / 8
    y = 2 * V ** 2 * np.cos(w[3] * X) + 2 / (5 * V + 1) * np.cos(w[11] * X) + 0.1 * rng.standard_normal(N)
    return y, X, V, w


@pytest.mark.parametrize("form", ["auto", "krs"])
def test_row_shards_sum_to_the_whole_gram(L, form, monkeypatch):
    """Partial problems over row shards built with the GLOBAL ranges: their Grams / right-hand sides add up to the
    whole problem's (same basis centres), for the structured and the dense Gram form."""
    monkeypatch.setenv("LPVS_GRAM_FORM", form)
    y, X, V, w = _rowshard_signal()
    Nv = 4
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        G, b = p.get_gram()
    ranges = L.lpv_ranges(X, V)
    assert np.array_equal(ranges, [V.min(), V.max(), np.abs(V).max(), np.abs(X).max()])
    cuts = [0, 1777, 4001, len(y)]
    Gs, bs = np.zeros_like(G), np.zeros_like(b)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        with L.Problem.lpv_rows(y[lo:hi], X[lo:hi], V[lo:hi], w, Nv, ranges) as p:
            Gr, br = p.get_gram()
            Gs += Gr; bs += br
    tol = 1e-12 + 4.5e-16 * np.abs(w).max() * np.abs(X).max()
    assert np.abs(Gs - G).max() <= tol * np.abs(G).max()
    assert np.abs(bs - b).max() <= tol * np.abs(b).max() * 10


def _rowshard_rank(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share the one GPU of the test box
    import lpvspectral_jl_amd as L2
    y, X, V, w = _rowshard_signal()
    lo, hi = L2.sharding.shard_range(len(y), world, rank)
    se = L2.ls_sparse_spectral_lpv_rowsharded(y[lo:hi], X[lo:hi], V[lo:hi], w, 4, λ=2.0, dist=dist, iters=300, tol=0.0, printerval=1000)
    dist.barrier()
    q.put((rank, np.asarray(se.x)))
    dist.destroy_process_group()


def test_row_sharded_solve_two_ranks_matches_single_process(L):
    """ls_sparse_spectral_lpv_rowsharded on 2 ranks (gloo, host-staged all-reduce; RCCL on a multi-GPU node) returns
    on every rank the coefficients of the single-process solve of the whole signal."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    y, X, V, w = _rowshard_signal()
    ref = L.ls_sparse_spectral_lpv(y, X, V, w, 4, λ=2.0, iters=300, tol=0.0, printerval=1000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rowshard_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    scale = np.abs(ref.x).max()
    assert np.array_equal(res[0][1], res[1][1])                      # replicated ADMM: identical on both ranks
    assert np.abs(res[0][1] - ref.x).max() <= 1e-9 * scale
    assert np.array_equal(np.abs(res[0][1]) > 0, np.abs(ref.x) > 0)  # same support


@pytest.mark.parametrize("zero_first,weighted", [(True, False), (False, True), (True, True)])
def test_fourier_structured_gram_agrees_with_dense(L, oracle, zero_first, weighted, monkeypatch):
    """Fourier problems on a uniform frequency grid take the structured Gram (nudft.hip); it must agree with the dense
    MFMA panel form and with the oracle, with and without the zero frequency / weights."""
    rng = np.random.default_rng(17)
    N, Nf = 5000, 40
    t = np.sort(rng.random(N) * 300.0)
    f = (np.arange(Nf) if zero_first else np.arange(1, Nf + 1)) / 97.0
    y = np.sin(2 * np.pi * f[7] * t) + 0.2 * rng.standard_normal(N)
    W = rng.random(N) + 0.5 if weighted else None
    res = {}
    for form in ("auto", "panel"):
        monkeypatch.setenv("LPVS_GRAM_FORM", form)
        with L.Problem.fourier(y, t, f, W) as p:
            res[form] = p.get_gram() + (p.timing()["gram_form"],)
    assert res["auto"][2] == "ap" and res["panel"][2] == "panel"
    A, zf = oracle.get_fourier_regressor(t, f)
    Go, bo = oracle.gram(A, y, W)
    tol = 1e-12 + 4.5e-16 * 2 * np.pi * f.max() * t.max()
    for form in ("auto", "panel"):
        G, b, _ = res[form]
        assert np.abs(G - Go).max() <= tol * np.abs(Go).max(), form
        assert np.abs(b - bo).max() <= 10 * tol * np.abs(bo).max(), form
    monkeypatch.setenv("LPVS_GRAM_FORM", "ap")                      # forcing the structured form on a non-uniform grid is an error
    with pytest.raises(ValueError):
        L.Problem.fourier(y, t, np.sort(rng.random(Nf)) + 0.01, W)


def test_tls_spectral_known_answer_and_oracle(L, oracle):
    """src/lsfft.jl:85-99 / test/runtests.jl:194-195: findmax(abs2(x)) ~ (2 length(freqs), 101) for y = sin(2 pi t); and
    agreement with the SVD form on a noisy record."""
    t = np.arange(0, 100, 0.1)
    y = np.sin(2 * np.pi * t)
    x, f = L.tls_spectral(y, t)
    p = np.abs(x) ** 2
    assert len(f) == 500 and abs(p.max() - 1000.0) < 1e-4 and p.argmax() + 1 == 101
    rng = np.random.default_rng(8)
    t2 = np.sort(rng.random(800) * 80)
    f2 = np.arange(1, 41) / 16.0
    y2 = np.sin(2 * np.pi * f2[9] * t2) + 0.3 * rng.standard_normal(800)
    x2, _ = L.tls_spectral(y2, t2, f2)
    xo, _ = oracle.tls_spectral(y2, t2, f2)
    assert rel(x2, xo) <= 1e-9


@pytest.mark.parametrize("ns,prox", [(3, "ball"), (8, "group"), (8, "ball"), (12, "group")])
def test_multi_signal_tile_product_variants_agree(L, oracle, ns, prox, monkeypatch):
    """The multi-signal tile product (symv_tile_mfma_ws_kernel) in its variants -- 4x4x4 four-block MFMA (ns <= 8) or the 16-column MFMA,
    one P1 record per tile or per run of tiles (segments of 2 / 4 tiles here; 8 at cfg5) -- through both consumers of the partials
    (symv_reduce_kernel for IndBallL0, admm_fused_update2_kernel for the group prox): same supports, same stopping iterations, iterates
    equal to summation order."""
    rng = np.random.default_rng(31)
    N, Nf, Nv = 3000, 192, 16                                   # n = 6144: 48 row blocks, 1176 tiles
    X = np.sort(10 * rng.random(N)); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf
    Y = np.stack([np.cos(w[(3 * q + 1) % Nf] * X) * (1 + q * V) + 0.3 * np.sin(w[(5 * q + 2) % Nf] * X) + 0.05 * rng.standard_normal(N)
                  for q in range(ns)], axis=1)
    g = L.IndBallL0(6) if prox == "ball" else None
    kw = dict(λ=3.0, iters=300, tol=1e-6, μ=0.05, printerval=100000, proxg=g)
    out = {}
    # (round 5: the column-panel walk -- P2 sums in registers across the rows a workgroup walks in a panel, one record per workgroup and panel;
    # opt-in, LPVS_MULTI_WALK=panel, and only where the run walk and the four-block MFMA apply: it falls back to runs otherwise)
    for name, env in (("tiles", {"LPVS_MULTI_RUNS": "0"}), ("runs2", {"LPVS_MULTI_RUNS": "2"}), ("runs4", {"LPVS_MULTI_RUNS": "4"}),
                      ("panel", {"LPVS_MULTI_RUNS": "2", "LPVS_MULTI_WALK": "panel"}), ("split/panel", {"LPVS_MULTI_RUNS": "2", "LPVS_MULTI_WALK": "panel", "LPVS_M_STORAGE": "split"}),
                      ("mfma16/tiles", {"LPVS_MULTI_RUNS": "0", "LPVS_MULTI_MFMA": "16"}), ("mfma16/runs2", {"LPVS_MULTI_RUNS": "2", "LPVS_MULTI_MFMA": "16"}),
                      ("split/runs2", {"LPVS_MULTI_RUNS": "2", "LPVS_M_STORAGE": "split"}), ("split/mfma16", {"LPVS_MULTI_RUNS": "0", "LPVS_MULTI_MFMA": "16", "LPVS_M_STORAGE": "split"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ses = L.ls_sparse_spectral_lpv_multi(Y, X, V, w, Nv, **kw)
        for k in env:
            monkeypatch.delenv(k)
        out[name] = np.stack([se.x for se in ses], axis=1)
    ref = out["split/mfma16"]                                    # (round 2's kernel on round 2's storage)
    assert 0 < np.count_nonzero(ref) < ref.size
    for name, x in out.items():
        assert np.array_equal(x != 0, ref != 0), name
        assert rel(x, ref) <= (1e-11 if name.startswith("split") else 1e-9), (name, rel(x, ref))   # (mixed storage where the tiles qualify: 36 bits)
    q = ns - 1
    se = L.ls_sparse_spectral_lpv(Y[:, q].copy(), X, V, w, Nv, **kw)     # and the single-signal solve of the last column (scalar tile product)
    assert np.array_equal(se.x != 0, out["runs2"][:, q] != 0) and rel(out["runs2"][:, q], se.x) <= 1e-9
    if ns == 8 and prox == "ball":
        # ... and the ORACLE on the kernel cfg5 runs (4x4x4 MFMA, partials per run, one-pass top-r selection): Gram-form ADMM on the
        # device Gram at equal iteration counts, every channel: rel-L2 <= 1e-9, identical support, identical stopping iteration
        monkeypatch.setenv("LPVS_MULTI_RUNS", "2")
        with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
            G, _ = p.get_gram(); B = p.get_rhs()
            p.set_prox(L.IndBallL0(6))
            p.admm_init(None, μ=0.05, tol=1e-6)
            assert p.matvec_info()["kernel"] == "symv_tile_mfma_ws_kernel" and p.matvec_info()["signals_per_pass"] == 8
            p.admm_run(300)
            per = [p.admm_status(c) for c in range(ns)]
            _, z, _ = p.admm_get()
        for c in (0, 5):
            ro = oracle.admm_gram(G, B[:, c], oracle.IndBallL0(6), iters=300, tol=1e-6, mu=0.05)
            assert ro["iters"] == per[c][0], (c, ro["iters"], per[c])
            assert np.array_equal(ro["z"] != 0, z[:, c] != 0) and rel(z[:, c], ro["z"]) <= 1e-9, (c, rel(z[:, c], ro["z"]))


def test_panel_walk_falls_back_where_its_records_do_not_fit(L, monkeypatch):
    """the opt-in column-panel walk of the multi-signal product indexes its P2 records as (flush index) * 4 + c into the ntiles records
    a signal owns; with LPVS_MULTI_RUNS=2 a short triangle (45 row blocks: ntiles = 1035 < (256 + 12) * 4 flush records) would overrun them.  The plan
    is now checked against the record area and the run walk taken instead: at the smallest sizes around the limit the panel request gives the run
    walk's iterates (bit for bit where it falls back, to summation order where the panel walk still applies)."""
    rng = np.random.default_rng(41)
    N, Nv, ns = 3000, 16, 8
    for Nf in (180, 192):                                            # n = 5760 (45 row blocks: falls back), 6144 (48: the panel walk runs)
        X = np.sort(10 * rng.random(N)); V = np.linspace(0, 1, N)
        w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf
        Y = np.stack([np.cos(w[(3 * q + 1) % Nf] * X) * (1 + q * V) + 0.3 * np.sin(w[(5 * q + 2) % Nf] * X) + 0.05 * rng.standard_normal(N) for q in range(ns)], axis=1)
        out = {}
        for name, env in (("runs", {"LPVS_MULTI_RUNS": "2"}), ("panel", {"LPVS_MULTI_RUNS": "2", "LPVS_MULTI_WALK": "panel"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
                p.set_prox(L.IndBallL0(6))
                p.admm_init(None, μ=0.05, tol=0.0)
                assert p.matvec_info()["kernel"] == "symv_tile_mfma_ws_kernel"
                p.admm_run(120)
                out[name] = p.admm_get()
            for k in env:
                monkeypatch.delenv(k)
        for a, b in zip(out["panel"], out["runs"]):
            assert np.all(np.isfinite(a)) and rel(a, b) <= 1e-12, (Nf, rel(a, b))
        if Nf == 180:
            assert all(np.array_equal(a, b) for a, b in zip(out["panel"], out["runs"]))      # the fall-back IS the run walk
