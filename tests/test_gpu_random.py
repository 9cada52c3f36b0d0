"""Seeded randomized parity sweep: random sizes / options, device path (through the C-ABI) vs the CPU oracle.
Each case checks the materialised regressor, the Gram / right-hand side and a short ADMM trajectory."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.mark.parametrize("seed", range(12))
def test_random_lpv_case(L, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(20, 900))
    Nf = int(rng.integers(1, 30))
    Nv = int(rng.integers(2, 12))
    normalize = bool(rng.integers(0, 2))
    X = np.sort(rng.random(N)) * 10 ** rng.uniform(0, 4)              # phases up to ~1e6 rad
    V = rng.standard_normal(N) if seed % 2 else np.linspace(-1, 2, N)
    w = np.sort(rng.random(Nf)) * 30 + 0.1                            # NOT equidistant
    Y = rng.standard_normal(N)
    Phi = L.lpv_regressor(X, V, w, Nv, normalize, False, True)
    Po = oracle.lpv_regressor(X, V, w, Nv, normalize, False, True)
    assert np.abs(Phi - Po).max() <= 4e-15 * max(1.0, np.abs(Po).max())
    with L.Problem.lpv(Y, X, V, w, Nv, normalize) as p:
        G, b = p.get_gram()
        lam = float(10 ** rng.uniform(-2, 0.5)) * np.sqrt(N)
        mu = float(10 ** rng.uniform(-2, 0))
        p.set_prox(L.SlicedSeparableSum.frequency_groups(lam, Nf, 2 * Nv))
        p.admm_init(None, μ=mu, tol=0.0)
        it, nxz, _ = p.admm_run(25)
        x, z, u = p.admm_get()
    Go, bo = Po.T @ Po, Po.T @ Y
    # the reference rounds every phase, fl(w*x) (error <= |w x| 2^-53 rad per term); the structured Gram form (used when w
    # is an arithmetic progression, e.g. Nf <= 2) evaluates the exact phase, so the comparison carries that term
    tol = 1e-12 + 4.5e-16 * float(np.abs(w).max() * np.abs(X).max())
    assert np.abs(G - Go).max() <= tol * np.abs(Go).max() and np.abs(b - bo).max() <= tol * max(np.abs(bo).max(), np.abs(Po).max() * np.abs(Y).sum() * 1e-3)
    ro = oracle.admm_gram(Go, bo, oracle.GroupL2(lam, 2 * Nv), iters=25, tol=0.0, mu=mu, history=True)
    scale = max(np.linalg.norm(ro["x"]), mu * np.linalg.norm(bo))     # natural size of the iterates (x_1 ~ mu*b)
    assert it == 25
    for a_, b_ in ((x, ro["x"]), (u, ro["u"]), (z, ro["z"])):
        assert np.linalg.norm(a_ - b_) <= 1e-8 * scale
    assert np.array_equal(z != 0, ro["z"] != 0)


@pytest.mark.parametrize("seed", range(12))
def test_random_fourier_case(L, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    N = int(rng.integers(8, 1500))
    Nf = int(rng.integers(1, 70))
    zero = bool(rng.integers(0, 2))
    weighted = bool(rng.integers(0, 2))
    t = np.cumsum(rng.random(N) + 0.01) * 10 ** rng.uniform(-1, 2)
    f = np.sort(rng.random(Nf)) * 0.5 + 1e-3
    if zero:
        f = np.concatenate([[0.0], f])
    y = rng.standard_normal(N)
    W = rng.random(N) + 0.1 if weighted else None
    A, zf = L.get_fourier_regressor(t, f)
    Ao, zo = oracle.get_fourier_regressor(t, f)
    assert zf == zo and np.abs(A - Ao).max() <= 4e-15
    kinds = [(L.NormL1, oracle.NormL1, 10 ** rng.uniform(-2, 0)), (L.NormL0, oracle.NormL0, 10 ** rng.uniform(-3, -1)),
             (L.IndBallL0, oracle.IndBallL0, int(rng.integers(1, max(2, A.shape[1] // 2))))]
    gd_c, go_c, par = kinds[seed % 3]
    mu = float(10 ** rng.uniform(-2, 0))
    x, _ = L.ls_sparse_spectral(y, t, f, W, proxg=gd_c(par), iters=30, tol=0.0, μ=mu, printerval=10 ** 6)
    xo, _, ro = oracle.ls_sparse_spectral(y, t, f, W, proxg=go_c(par), iters=30, tol=0.0, mu=mu)
    Gq = Ao.T @ ((W[:, None] if weighted else 1.0) * Ao)
    bq = Ao.T @ ((W if weighted else 1.0) * y)
    scale = max(np.linalg.norm(ro["x"]), mu * np.linalg.norm(bq))
    assert np.linalg.norm(x - xo) <= 1e-6 * scale                                    # vs the faithful CG form
    with L.Problem.fourier(y, t, f, W) as p:
        G, b = p.get_gram()
    assert np.abs(G - Gq).max() <= 1e-12 * np.abs(Gq).max()
