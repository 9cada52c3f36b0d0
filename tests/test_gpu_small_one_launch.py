"""One launch per ADMM iteration below np = 2048 (``admm_small_iter_kernel``, csrc/admm_small.hip): the full-matrix path of cfg2-sized problems
(n = 1024 / 1023) -- every workgroup redoes the update and takes the stopping decision itself, x and u double-buffered by launch
parity.  Against ``oracle.admm_gram`` (src/lasso.jl:136-171 on the Gram form of :98) with the kernel name asserted; the reference's
stopping rule in the iteration it belongs to (both parities of the stopping launch, inside and at the end of a chunk); any chunking
of ``lpvs_admm_run`` and the graph replay give the same bits; the two-launch scheme agrees to rounding.  GPU only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _fourier_case(rng, N, Nf, zero):
    t = np.sort(rng.random(N)) * N
    f = (np.arange(Nf) if zero else np.arange(1, Nf + 1)) / (2.0 * Nf)
    y = 2 * np.sin(2 * np.pi * f[16] * t + 0.3) + np.sin(2 * np.pi * f[99] * t) + 0.25 * np.sin(2 * np.pi * f[300] * t + 1.0) + 0.1 * rng.standard_normal(N)
    return y, t, f


@pytest.fixture(scope="module", params=[False, True], ids=["n1024", "n1023_zero_frequency"])
def case(request, L):
    rng = np.random.default_rng(71)
    y, t, f = _fourier_case(rng, 1500, 512, request.param)        # (few samples per unknown: a wide spectrum of G, hundreds of iterations to converge)
    with L.Problem.fourier(y, t, f) as p:
        G, b = p.get_gram()
        n = p.n
    assert n == (1023 if request.param else 1024)
    return dict(y=y, t=t, f=f, G=G, b=b, n=n)


def _run(L, c, prox, iters, tol, chunks=None, mode=None, x0=None):
    with L.Problem.fourier(c["y"], c["t"], c["f"]) as p:
        if mode is not None:
            p.set_option("iteration", mode)
        p.set_prox(prox)
        p.admm_init(x0, μ=0.05, tol=tol)
        info = p.matvec_info()
        it = conv = nxz = None
        for k in (chunks or [iters]):
            it, nxz, conv = p.admm_run(k)
        x, z, u = p.admm_get()
    return dict(x=x, z=z, u=u, it=it, nxz=nxz, conv=conv, info=info)


def _proxes(L, oracle, c):
    b = c["b"]
    lam1 = float(np.quantile(np.abs(b), 0.1))
    glen = 31 if c["n"] == 1023 else 16                                   # 1023 = 33 x 31
    gb = np.linalg.norm(b.reshape(-1, glen), axis=1)
    xr = np.linalg.solve(c["G"] + np.eye(c["n"]) / 0.05, b)
    return {"l1": (L.NormL1(lam1), oracle.NormL1(lam1)),
            "l0": (L.NormL0(float(np.quantile(np.abs(xr), 0.3) ** 2 / 0.1)), oracle.NormL0(float(np.quantile(np.abs(xr), 0.3) ** 2 / 0.1))),
            "group": (L.SlicedSeparableSum.frequency_groups(float(np.quantile(gb, 0.1)), c["n"] // glen, glen), oracle.GroupL2(float(np.quantile(gb, 0.1)), glen))}


@pytest.mark.parametrize("kind", ["l1", "l0", "group"])
def test_small_one_launch_iteration_against_oracle(L, oracle, case, kind):
    prox, oprox = _proxes(L, oracle, case)[kind]
    r = _run(L, case, prox, 300, 0.0)
    assert r["info"]["kernel"] == "admm_small_iter_kernel" and r["info"]["one_launch_iteration"], r["info"]
    ro = oracle.admm_gram(case["G"], case["b"], oprox, iters=300, tol=0.0, mu=0.05, history=True)
    assert r["it"] == 300 == ro["iters"]
    errs = {k: rel(r[k], ro[k]) for k in ("x", "z", "u")}
    print(f"n={case['n']} {kind}: rel-L2 vs oracle x {errs['x']:.2e} z {errs['z']:.2e} u {errs['u']:.2e}; nnz {np.count_nonzero(ro['z'])}")
    assert max(errs.values()) <= 1e-9, errs
    assert np.array_equal(r["z"] != 0, ro["z"] != 0) and 0 < np.count_nonzero(ro["z"]) < ro["z"].size
    assert abs(r["nxz"] - ro["nxz"][-1]) <= 1e-9 * max(ro["nxz"][-1], np.linalg.norm(ro["x"]))
    # the two-launch scheme (symv_kernel + prox kernel): the same iterates to rounding
    r2 = _run(L, case, prox, 300, 0.0, mode="two")
    assert r2["info"]["kernel"] == "symv_kernel" and not r2["info"].get("one_launch_iteration", False)
    assert rel(r["z"], r2["z"]) <= 1e-12 and np.array_equal(r["z"] != 0, r2["z"] != 0)


def test_small_one_launch_stops_in_the_reference_iteration(L, oracle, case):
    """tol > 0 (src/lasso.jl:164): the stopping iteration of the oracle at both parities, in the middle of a chunk, as a chunk's last
    iteration and as its first; the state returned is that of the stopping iteration."""
    prox, oprox = _proxes(L, oracle, case)["l1"]
    full = oracle.admm_gram(case["G"], case["b"], oprox, iters=700, tol=0.0, mu=0.05, history=True)
    nx = full["nxz"]
    for k0 in (120, 301):
        k = next(k for k in range(k0, 690, 2) if nx[k] < 0.999 * nx[k - 1] and nx[:k].min() > np.sqrt(nx[k] * nx[k - 1]))
        tol = np.sqrt(nx[k] * nx[k - 1])                                   # first crossing: iteration k + 1 (1-based)
        ro = oracle.admm_gram(case["G"], case["b"], oprox, iters=700, tol=tol, mu=0.05)
        assert ro["iters"] == k + 1
        for chunks in ([700], [k, 700], [k + 1, 700], [k - 1, 1, 1, 700], [50] * 14):
            r = _run(L, case, prox, 700, tol, chunks=chunks)
            assert r["info"]["kernel"] == "admm_small_iter_kernel"
            assert r["conv"] and r["it"] == k + 1, (chunks, r["it"], k + 1)
            for q in ("x", "z", "u"):
                assert rel(r[q], ro[q]) <= 1e-9, (chunks, q, rel(r[q], ro[q]))
            assert np.array_equal(r["z"] != 0, ro["z"] != 0)


def test_small_one_launch_is_independent_of_chunking_and_graph_replay(L, oracle, case):
    """1200 iterations at once (two graph replays of 250 + direct launches), in ragged chunks and one iteration at a time: the same bits;
    a warm start (x0) enters through the first launch of the first chunk."""
    prox, _ = _proxes(L, oracle, case)["group"]
    a = _run(L, case, prox, 1200, 0.0)
    b = _run(L, case, prox, 1200, 0.0, chunks=[1, 2, 3, 594, 600])
    c = _run(L, case, prox, 1200, 0.0, chunks=[1] * 40 + [1160])
    for q in ("x", "z", "u"):
        assert np.array_equal(a[q], b[q]) and np.array_equal(a[q], c[q]), q
    assert a["it"] == b["it"] == c["it"] == 1200
    x0 = 0.1 * np.random.default_rng(5).standard_normal(case["n"])
    w1 = _run(L, case, prox, 200, 0.0, x0=x0)
    w2 = _run(L, case, prox, 200, 0.0, x0=x0, mode="two")
    assert rel(w1["z"], w2["z"]) <= 1e-12 and not np.array_equal(w1["z"], _run(L, case, prox, 200, 0.0)["z"])


def test_small_one_launch_multi_signal_and_resume(L, oracle):
    """Several right-hand sides sharing a small regressor (blockIdx.y = signal, each with its own stopping iteration) and re-entry from
    saved iterates (lpvs_admm_set_state) in the middle of a run."""
    rng = np.random.default_rng(8)
    N, Nf, Nv = 6000, 24, 4
    X = np.sort(rng.random(N) * 10); V = np.linspace(0, 1, N)
    w = 2 * np.pi * np.arange(1, Nf + 1)
    Y = np.stack([np.cos(w[3] * X) * (1 + V), 2 * np.cos(w[10] * X - 0.4) * V, np.cos(w[17] * X)], axis=1) + 0.05 * rng.standard_normal((N, 3))
    lam = 30.0
    with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
        G, _ = p.get_gram(); B = p.get_rhs()
        p.set_prox(L.SlicedSeparableSum.frequency_groups(lam, Nf, 2 * Nv))
        p.admm_init(None, μ=0.05, tol=1e-7)
        assert p.matvec_info()["kernel"] == "admm_small_iter_kernel"
        p.admm_run(5000)
        x, z, u = p.admm_get()
        its = [p.admm_status(q)[0] for q in range(3)]
    for q in range(3):
        ro = oracle.admm_gram(G, B[:, q], oracle.GroupL2(lam, 2 * Nv), iters=5000, tol=1e-7, mu=0.05)
        assert ro["iters"] < 5000 and its[q] == ro["iters"], (q, its[q], ro["iters"])
        assert rel(z[:, q], ro["z"]) <= 1e-9 and np.array_equal(z[:, q] != 0, ro["z"] != 0)
    assert len(set(its)) > 1                                              # the signals stop in different iterations
    y1 = Y[:, 0]
    with L.Problem.lpv(y1, X, V, w, Nv) as p:
        p.set_prox(L.SlicedSeparableSum.frequency_groups(lam, Nf, 2 * Nv))
        p.admm_init(None, μ=0.05, tol=0.0)
        p.admm_run(77)
        st = p.admm_get()
        p.admm_run(123)
        ref = p.admm_get()
        p.admm_init(None, μ=0.05, tol=0.0)
        p.admm_set_state(*st, iters=77)
        it, _, _ = p.admm_run(123)
        assert it == 200
        for a_, b_ in zip(p.admm_get(), ref):
            assert np.array_equal(a_, b_)


def _spd_problem(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((3 * n // 2, n)) / np.sqrt(n)
    k = min(9, n)
    xs = np.zeros(n); xs[rng.choice(n, k, replace=False)] = rng.standard_normal(k) * 3
    y = A @ xs + 0.01 * rng.standard_normal(A.shape[0])
    return A.T @ A, A.T @ y


@pytest.mark.parametrize("n", [5, 100, 128, 129, 640, 1000, 1025, 1100, 1900, 2047])
def test_small_one_launch_edge_sizes(L, oracle, n):
    """Sizes around the kernel's boundaries: one row block (np = 128), exactly a block, one past it, the largest size of the
    np <= 1024 instances and the first of the others, the largest padded size below the tile-packed path (np = 1920), and n = 2047
    (np = 2048: the tile-packed path takes over -- the same result from a different kernel).  Explicit-Gram problems, L1 and a group
    prox whose groups do not align with anything, tol > 0 (every workgroup forms the norm) and tol = 0 (only workgroup 0 does)."""
    G, b = _spd_problem(n, 100 + n)
    glen = next(g for g in (7, 5, 4, 3, 1) if n % g == 0)
    lam = float(np.quantile(np.abs(b), 0.5))
    gb = np.linalg.norm(b.reshape(-1, glen), axis=1)
    for prox, oprox in ((L.NormL1(lam), oracle.NormL1(lam)),
                        (L.SlicedSeparableSum.frequency_groups(float(np.quantile(gb, 0.5)), n // glen, glen), oracle.GroupL2(float(np.quantile(gb, 0.5)), glen))):
        for tol in (0.0, 1e-6):
            with L.Problem.gram(G, b) as p:
                p.set_prox(prox)
                p.admm_init(None, μ=0.05, tol=tol)
                info = p.matvec_info()
                it, nxz, conv = p.admm_run(400)
                x, z, u = p.admm_get()
            assert (info["kernel"] == "admm_small_iter_kernel") == (n <= 1920), (n, info)
            ro = oracle.admm_gram(G, b, oprox, iters=400, tol=tol, mu=0.05, history=True)
            k = ro["iters"]
            unambiguous = tol == 0.0 or k == 400 or min(abs(ro["nxz"][k - 1] - tol), abs(ro["nxz"][k - 2] - tol) if k >= 2 else 1.0) > 1e-6 * tol
            if unambiguous:
                assert it == k, (n, tol, it, k)
                for a, q in ((x, "x"), (z, "z"), (u, "u")):
                    assert rel(a, ro[q]) <= 1e-9, (n, tol, q, rel(a, ro[q]))
                assert np.array_equal(z != 0, ro["z"] != 0), (n, tol)
