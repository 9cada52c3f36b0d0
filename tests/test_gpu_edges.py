"""Edge cases of the structured (NUDFT) Gram and of the batched-window engine: single frequencies, slot padding,
sample counts that straddle the kernel's 16-sample blocks and chunk boundaries, ragged window tails, empty inputs."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tol(w, x):
    return 1e-12 + 4.5e-16 * np.abs(w).max() * np.abs(x).max()


@pytest.mark.parametrize("N,Nf,Nv", [(37, 1, 2), (37, 3, 2), (1000, 7, 3), (8193, 9, 2), (16385, 17, 4)])
def test_lpv_structured_gram_small_and_ragged_sizes(L, oracle, N, Nf, Nv):
    rng = np.random.default_rng(N + Nf)
    X = np.sort(rng.random(N) * 12.0); V = rng.random(N) * 2 - 0.5
    w = 0.7 + 1.3 * np.arange(Nf)                       # arithmetic progression, not starting at its step
    y = rng.standard_normal(N)
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        G, b = p.get_gram()
        assert p.timing()["gram_form"] in ("ap", "ap-nufft")
    Phi = oracle.lpv_regressor(X, V, w, Nv)
    Go, bo = oracle.gram(Phi, y)
    assert np.abs(G - Go).max() <= _tol(w, X) * np.abs(Go).max()
    assert np.abs(b - bo).max() <= 10 * _tol(w, X) * max(np.abs(bo).max(), 1.0)


@pytest.mark.parametrize("f", [np.array([0.0]), np.array([0.25]), np.arange(5) / 8.0, np.arange(1, 10) / 20.0])
def test_fourier_structured_gram_tiny_grids(L, oracle, f):
    rng = np.random.default_rng(len(f))
    N = 531
    t = np.sort(rng.random(N) * 50)
    y = rng.standard_normal(N)
    with L.Problem.fourier(y, t, f) as p:
        G, b = p.get_gram()
        assert p.timing()["gram_form"] in ("ap", "ap-nufft")
    A, zf = oracle.get_fourier_regressor(t, f)
    Go, bo = oracle.gram(A, y)
    tol = _tol(2 * np.pi * f, t)
    assert G.shape == Go.shape
    assert np.abs(G - Go).max() <= tol * np.abs(Go).max()
    assert np.abs(b - bo).max() <= 10 * tol * max(np.abs(bo).max(), 1.0)


def test_batched_windows_ragged_tail_and_overlap(L):
    """L not a multiple of the hop, overlapping windows, hanning weights: the batched engine (structured per-window Gram)
    equals the window-by-window loop."""
    rng = np.random.default_rng(5)
    Ltot, nw = 5000, 9
    t = np.arange(Ltot) * 0.5
    y = np.sin(2 * np.pi * 0.11 * t) + 0.5 * np.sin(2 * np.pi * 0.31 * t) + 0.2 * rng.standard_normal(Ltot)
    freqs = np.arange(0, 40) / 80.0
    kw = dict(nw=nw, noverlap=100, window_func=L.hanning, estimator=L.ls_sparse_spectral, λ=0.5, iters=300, tol=0.0,
              printerval=1000, out=io.StringIO())
    S1, f1 = L.ls_windowpsd(y, t, freqs, batched=True, **kw)
    S2, f2 = L.ls_windowpsd(y, t, freqs, batched=False, **kw)
    assert np.array_equal(f1, f2)
    assert np.abs(S1 - S2).max() <= 1e-9 * np.abs(S2).max()


def test_empty_and_mismatched_inputs_raise(L):
    e = np.zeros(0)
    with pytest.raises((ValueError, AssertionError)):
        L.Problem.lpv(e, e, e, np.array([1.0, 2.0]), 2)
    with pytest.raises((ValueError, AssertionError)):
        L.Problem.fourier(e, e, np.array([0.1]))
    with pytest.raises(AssertionError):
        L.Problem.lpv(np.zeros(5), np.zeros(4), np.zeros(5), np.array([1.0]), 2)
    with pytest.raises(ValueError):                       # zero frequency not first, src/lsfft.jl:22
        L.Problem.fourier(np.zeros(8), np.arange(8.0), np.array([0.1, 0.0]))


@pytest.mark.parametrize("knobs", [{}, {"LPVS_KW": "256"}, {"LPVS_KW": "256", "LPVS_LOOKAHEAD": "0"}, {"LPVS_LOOKAHEAD": "0"},
                                   {"LPVS_PIVOT": "sweep64"}, {"LPVS_FACTOR": "sweep64"}, {"LPVS_FACTOR_SCHEME": "steps"},
                                   {"LPVS_CHAIN": "split"}, {"LPVS_PIVOT": "regs"}, {"LPVS_FACTOR_GROUP": "1"}, {"LPVS_FACTOR_GROUP": "2"},
                                   {"LPVS_FACTOR_GROUP": "3"}, {"LPVS_FACTOR_GROUP": "4", "LPVS_RU_STAGE": "8"}, {"LPVS_RESERVE_CUS": "0"},
                                   {"LPVS_BAND_TILE": "128"}, {"LPVS_BAND_TILE": "64", "LPVS_FACTOR_GROUP": "4"}, {"LPVS_PIVOT_ALONE": "0"}])
@pytest.mark.parametrize("n", [1024, 2100, 2300, 2500])
def test_factorisation_variants_give_the_inverse(L, knobs, n, monkeypatch):
    """Every factorisation variant (128 / 256-wide outer blocks incl. the ragged last block, with and without look-ahead,
    one-workgroup or swept pivot block, single-level sweep; from np = 2048 the default is the GROUP schedule -- up to four 128-wide
    panels per pass of the trailing update, pivot block inverted on the matrix cores, fused pivot-chain tail: np = 2176 is 17 pivot
    blocks (groups 4 + 4 + 4 + 4 + 1), 2304 = 18, 2560 = 20; LPVS_FACTOR_GROUP = 1 / 2 / 3 panels per pass, LPVS_PIVOT=regs the
    register kernel for the pivot block, LPVS_FACTOR_SCHEME=steps / LPVS_CHAIN=split the one-panel schedule / the separate gather +
    GEMM kernels, LPVS_RESERVE_CUS=0 no CU mask, LPVS_BAND_TILE=64|128 the band launches' tile size, LPVS_PIVOT_ALONE=0 the pivot
    kernel without its LDS padding) returns (G + I/mu)^-1: |M H - I| at rounding level."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n + 50, n))
    G = A.T @ A
    b = rng.standard_normal(n)
    with L.Problem.gram(G, b) as p:
        M = p.get_inverse(20.0)
    H = G + 20.0 * np.eye(n)
    assert np.abs(M @ H - np.eye(n)).max() <= 5e-13
    assert np.array_equal(M, M.T)


@pytest.mark.parametrize("knobs", [{}, {"LPVS_LOOKAHEAD": "1"}, {"LPVS_KW": "256"}, {"LPVS_FACTOR_SCHEME": "steps"}, {"LPVS_CHAIN": "split"},
                                   {"LPVS_FACTOR_GROUP": "3"}, {"LPVS_PIVOT": "regs"}, {"LPVS_BAND_TILE": "128"}])
def test_deep_lookahead_factorisation_gives_the_inverse(L, knobs, monkeypatch):
    """np >= 6144 takes the depth-2 look-ahead schedule (three panel buffers, second band of the trailing update); a ragged
    size just above that threshold must agree with the depth-one schedule and with the definition of the inverse."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    n = 6200
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n + 50, n))
    G = A.T @ A
    b = rng.standard_normal(n)
    with L.Problem.gram(G, b) as p:
        M = p.get_inverse(20.0)
    H = G + 20.0 * np.eye(n)
    assert np.abs(M @ H - np.eye(n)).max() <= 5e-12
    assert np.array_equal(M, M.T)


@pytest.mark.parametrize("Nf,Nv", [(12, 4), (128, 8)])
def test_iterates_do_not_depend_on_chunking(L, Nf, Nv):
    """lpvs_admm_run in one call or in chunks (what printerval / cb do) gives the same iterates bit for bit, on the plain
    (n < 2048: hipGraph replay) and on the tile-packed (n >= 2048: deferred convergence commit) path; an early stop lands on
    the same iteration."""
    rng = np.random.default_rng(Nf)
    N = 3000
    X = np.sort(rng.random(N) * 60); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) / 8
    y = np.cos(w[3] * X) * (1 + V) + 0.1 * rng.standard_normal(N)
    g = L.SlicedSeparableSum.frequency_groups(2.0, Nf, 2 * Nv)
    res = []
    for chunks in ([300], [100, 100, 100], [1, 7, 150, 142]):
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            p.set_prox(g)
            p.admm_init(None, μ=0.05, tol=0.0)
            for c in chunks:
                it, nxz, conv = p.admm_run(c)
            res.append((it, nxz) + p.admm_get())
    for r in res[1:]:
        assert r[0] == res[0][0] == 300 and r[1] == res[0][1]
        for a, b in zip(r[2:], res[0][2:]):
            assert np.array_equal(a, b)
    stops = []
    for chunks in ([4000], [250] * 16):
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            p.set_prox(g)
            p.admm_init(None, μ=0.05, tol=1e-4)
            for c in chunks:
                it, nxz, conv = p.admm_run(c)
                if conv:
                    break
            stops.append((it, conv, nxz) + p.admm_get())
    assert stops[0][1] and stops[1][1] and stops[0][0] == stops[1][0] and stops[0][2] == stops[1][2]
    assert np.array_equal(stops[0][3], stops[1][3]) and np.array_equal(stops[0][4], stops[1][4])


def test_structured_gram_at_the_admission_boundary(L, monkeypatch):
    """Frequencies that deviate from an arithmetic progression by eps with max|eps| * max|x| just below / above the 1e-7 admission
    bound: below, the structured form is taken and its first-order correction keeps it within 1e-12 of the dense form (the neglected
    second order is (eps x)^2 / 2 <= 5e-15); above, the dense form runs."""
    rng = np.random.default_rng(77)
    N, Nf, Nv = 4000, 10, 3
    X = np.sort(rng.random(N) * 50.0); V = rng.random(N)
    y = rng.standard_normal(N)
    base = 0.3 + 0.9 * np.arange(Nf)
    pert = rng.uniform(-1, 1, Nf); pert[0] = pert[-1] = 0.0; pert /= np.abs(pert).max()
    res = {}
    for tag, scale in (("below", 0.9e-7 / X.max()), ("above", 3e-7 / X.max())):
        w = base + scale * pert
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            G, b = p.get_gram()
            form = p.timing()["gram_form"]
        monkeypatch.setenv("LPVS_GRAM_FORM", "krs")
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            Gd, bd = p.get_gram()
        monkeypatch.delenv("LPVS_GRAM_FORM")
        res[tag] = (form, np.abs(G - Gd).max() / np.abs(Gd).max(), np.abs(b - bd).max() / np.abs(bd).max())
    assert res["below"][0] in ("ap", "ap-nufft") and res["above"][0] in ("krs", "kr")
    assert res["below"][1] <= 1e-12 and res["below"][2] <= 1e-11, res
    assert res["above"][1] == 0.0
