"""The CPU oracle held DIRECTLY against the kernels bench.py times (src/lasso.jl:136-171 on the Gram form of :51 / :119-121):

  admm_iter_mixed_kernel<.., BATCH = false>   cfg3's dominant kernel: 36-bit fixed-point tiles, 64-bit fixed-point atomics, prox in the prologue
  symv_tile_mixed_kernel + admm_fused_update2  the two-launch scheme of the same storage (LPVS_ITERATION=two)
  admm_iter_mixed_kernel<.., BATCH = true>    cfg4's dominant kernel (one launch per iteration for a batch of windows), with and
                                              without non-temporal tile loads (the NT instance is what 1024 windows run)
  admm_iter_mixed_kernel<.., F32 = true>      the _f32 handles' one-launch iteration

Every case asserts the kernel that ran (``matvec_info`` / ``windowpsd_last_timing``) before comparing: a handle that fell back to
another storage would otherwise pass for the wrong reason.  Sizes are the smallest at which the mixed storage holds (n = 2048 needs
N = 2^18 samples for a diagonally dominant inverse); the oracle side is ``oracle.admm_gram`` / ``oracle.admm_quadratic`` on the Gram
read back from the device, so the comparison isolates the iteration (storage, atomics, prox, stopping rule) from the Gram.
Tolerances: rel-L2 <= 1e-9 in x, z, u with identical support and identical stopping iteration; _f32: 2e-5.  GPU only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)                      # (complex spectra stay complex: both parts are compared)
    dt = np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64
    a, b = a.astype(dt), b.astype(dt)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _signal(N, Nf, rng):
    X = np.sort(rng.random(N) * (10.0 * N / 500)); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25.0 / Nf
    y = 2 * V ** 2 * np.cos(w[Nf // 10] * X) + 2 / (5 * V + 1) * np.cos(w[Nf // 3] * X - 0.3) + 0.1 * rng.standard_normal(N)
    return y, X, V, w


@pytest.fixture(scope="module")
def lpv_case(L):
    """N = 2^18, Nf = 128, Nv = 8 -> n = 2048 (16 row blocks, 136 tiles, most of them fixed point) and its device Gram."""
    rng = np.random.default_rng(31)
    N, Nf, Nv = 1 << 18, 128, 8
    y, X, V, w = _signal(N, Nf, rng)
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        G, b = p.get_gram()
    # penalties that leave a NON-TRIVIAL support (so that "identical support" can fail): from the data's own scale -- a group / an
    # element is active at the solution roughly when its correlation with y exceeds the penalty
    gb = np.linalg.norm(b.reshape(Nf, 2 * Nv), axis=1)
    xr = np.linalg.solve(G + np.eye(len(b)) / 0.05, b)
    lams = dict(group=float(np.quantile(gb, 0.7)), l1=float(np.quantile(np.abs(b), 0.7)), l0=float(np.quantile(np.abs(xr), 0.7) ** 2 / (2 * 0.05)))
    return dict(N=N, Nf=Nf, Nv=Nv, y=y, X=X, V=V, w=w, G=G, b=b, lams=lams)


def _device_run(L, case, prox, iters, tol, monkeypatch, mode, f32=False):
    if mode == "two":
        monkeypatch.setenv("LPVS_ITERATION", "two")
    else:
        monkeypatch.delenv("LPVS_ITERATION", raising=False)
    args = [case[k] for k in ("y", "X", "V", "w")]
    if f32:
        args = [a.astype(np.float32) for a in args]
    with L.Problem.lpv(*args, case["Nv"]) as p:
        p.set_prox(prox)
        p.admm_init(None, μ=0.05, tol=tol)
        info = p.matvec_info()
        it, nxz, conv = p.admm_run(iters)
        x, z, u = p.admm_get()
        G, b = p.get_gram() if f32 else (None, None)
    return dict(x=x, z=z, u=u, it=it, nxz=nxz, conv=conv, info=info, G=G, b=b)


def _proxes(L, oracle, case):
    Nf, Nv, lam = case["Nf"], case["Nv"], case["lams"]
    return {"group": (L.SlicedSeparableSum.frequency_groups(lam["group"], Nf, 2 * Nv), oracle.GroupL2(lam["group"], 2 * Nv)),
            "l1": (L.NormL1(lam["l1"]), oracle.NormL1(lam["l1"])),
            "l0": (L.NormL0(lam["l0"]), oracle.NormL0(lam["l0"]))}


@pytest.mark.parametrize("kind", ["group", "l1", "l0"])
@pytest.mark.parametrize("mode,kernel", [("one", "admm_iter_mixed_kernel"), ("two", "symv_tile_mixed_kernel")])
def test_mixed_storage_iteration_against_oracle(L, oracle, lpv_case, kind, mode, kernel, monkeypatch):
    """300 iterations, tol = 0: the iterates of the benchmarked kernel against oracle.admm_gram on the same G, b."""
    prox, oprox = _proxes(L, oracle, lpv_case)[kind]
    r = _device_run(L, lpv_case, prox, 300, 0.0, monkeypatch, mode)
    assert r["info"]["kernel"] == kernel, r["info"]
    assert r["info"].get("one_launch_iteration", False) == (mode == "one")
    ro = oracle.admm_gram(lpv_case["G"], lpv_case["b"], oprox, iters=300, tol=0.0, mu=0.05, history=True)
    assert r["it"] == 300 == ro["iters"]
    errs = {k: rel(r[k], ro[k]) for k in ("x", "z", "u")}
    print(f"{kind}/{mode}: rel-L2 vs oracle x {errs['x']:.2e} z {errs['z']:.2e} u {errs['u']:.2e}; nnz {np.count_nonzero(ro['z'])}")
    assert max(errs.values()) <= 1e-9, errs
    assert np.array_equal(r["z"] != 0, ro["z"] != 0)
    assert 0 < np.count_nonzero(ro["z"]) < ro["z"].size                      # a support that could differ
    assert abs(r["nxz"] - ro["nxz"][-1]) <= 1e-8 * max(ro["nxz"][-1], 1e-300)


@pytest.mark.parametrize("mode,kernel", [("one", "admm_iter_mixed_kernel"), ("two", "symv_tile_mixed_kernel")])
def test_mixed_storage_stopping_iteration_against_oracle(L, oracle, lpv_case, mode, kernel, monkeypatch):
    """tol > 0 (src/lasso.jl:164): the same stopping iteration as the oracle and the iterates of that iteration."""
    prox, oprox = _proxes(L, oracle, lpv_case)["l1"]
    full = oracle.admm_gram(lpv_case["G"], lpv_case["b"], oprox, iters=600, tol=0.0, mu=0.05, history=True)
    nx = full["nxz"]
    for k0 in (150, 401):                                                    # both parities of the stopping iteration
        # a tolerance strictly between two consecutive residual norms, away from both: the stop is unambiguous in either arithmetic
        k = next(k for k in range(k0, 598, 2) if nx[k] < 0.999 * nx[k - 1] and nx[:k].min() > np.sqrt(nx[k] * nx[k - 1]))
        tol = np.sqrt(nx[k] * nx[k - 1])                                      # first crossing is iteration k + 1 (1-based)
        ro = oracle.admm_gram(lpv_case["G"], lpv_case["b"], oprox, iters=600, tol=tol, mu=0.05)
        r = _device_run(L, lpv_case, prox, 600, tol, monkeypatch, mode)
        assert r["info"]["kernel"] == kernel
        assert r["conv"] and r["it"] == ro["iters"] == k + 1, (r["it"], ro["iters"], k + 1)
        for q in ("x", "z", "u"):
            assert rel(r[q], ro[q]) <= 1e-9, (q, rel(r[q], ro[q]))
        assert np.array_equal(r["z"] != 0, ro["z"] != 0)


def test_f32_one_launch_iteration_against_oracle(L, oracle, lpv_case, monkeypatch):
    """_f32 handles (single-precision copy of the inverse through the one-launch kernel): against the oracle on the widened inputs'
    Gram, 2e-5 (SURVEY 8(d) asks 1e-3 of a Float32 path)."""
    prox, oprox = _proxes(L, oracle, lpv_case)["group"]
    r = _device_run(L, lpv_case, prox, 300, 0.0, monkeypatch, "one", f32=True)
    assert r["info"]["kernel"] == "admm_iter_mixed_kernel" and r["info"]["one_launch_iteration"], r["info"]
    assert "f32" in r["info"]["storage"]
    ro = oracle.admm_gram(r["G"], r["b"], oprox, iters=300, tol=0.0, mu=0.05)   # Gram of the float inputs widened to double
    assert r["it"] == 300
    e = rel(r["z"], ro["z"])
    print(f"f32 one-launch vs oracle: rel-L2(z) {e:.2e}")
    assert e <= 2e-5, e
    assert np.array_equal(r["z"] != 0, ro["z"] != 0)


# ---------------------------------------------------------------------------------------------------------------- window batches
def _window_case(rng):
    """6 windows x 2^14 samples, Nf = 256 with a zero frequency (nreg = 511 -> np = 512: four row blocks, ten tiles per window -- the
    tile shape of cfg4), equidistant samples, rect window, mu = 1e-4 as BASELINE.json's config 4."""
    n, nwin, Nf = 1 << 14, 6, 256
    t = np.arange(n * nwin, dtype=np.float64)
    f = np.arange(Nf) / 512.0
    y = np.sin(2 * np.pi * f[33] * t) + 0.5 * np.sin(2 * np.pi * f[100] * t + 0.4) + 0.3 * rng.standard_normal(n * nwin)
    return n, nwin, Nf, t, f, y


def _engine(L, iters, tol):
    from lpvspectral_jl_amd import _lib
    return dict(estimator=_lib.EST_SPARSE, lam=0.0, prox=(_lib.PROX_L1, 0.2, 0), μ=1e-4, tol=tol, iters=iters, sign=_lib.LINEAR_QUADRATIC_AS_WRITTEN)


def test_window_batch_one_launch_against_oracle(L, oracle, monkeypatch):
    """Every window of the batch against oracle.admm_quadratic on that window's Q = A'WA, q = A'Wy (src/lasso.jl:118-121) read back
    from a single-window device handle, and against the oracle's own end-to-end ls_sparse_spectral(y, t, f, W)."""
    from lpvspectral_jl_amd import api
    rng = np.random.default_rng(17)
    n, nwin, Nf, t, f, y = _window_case(rng)
    monkeypatch.delenv("LPVS_ITERATION", raising=False)
    monkeypatch.delenv("LPVS_NT_LOADS", raising=False)
    iters = 400
    x, its = api.windows_estimate([y], t, f, n, 0, None, _engine(L, iters, 0.0))
    tm = api.windowpsd_last_timing()
    assert tm["one_launch_iteration"] and tm["gram_form"] == "ap-nufft", tm
    assert np.all(its == iters)
    W = np.ones(n)
    worst = 0.0
    for i in range(nwin):
        yi, ti = y[i * n:(i + 1) * n], t[i * n:(i + 1) * n]
        with L.Problem.fourier(yi, ti, f, W) as p:
            Q, q = p.get_gram()
        ro = oracle.admm_quadratic(Q, q, oracle.NormL1(0.2), iters=iters, tol=0.0, mu=1e-4)
        zo = oracle.fourier2complex(ro["z"], 1)
        e = rel(x[0, i], zo)
        worst = max(worst, e)
        assert e <= 1e-9, (i, e)
        assert np.array_equal(x[0, i] != 0, zo != 0)
        if i in (0, nwin - 1):                                               # the whole reference pipeline on the host (regressor, Gram, ADMM)
            xo = oracle.ls_sparse_spectral(yi, ti, f, W, proxg=oracle.NormL1(0.2), iters=iters, tol=0.0, mu=1e-4)[0]
            assert rel(x[0, i], xo) <= 1e-8, rel(x[0, i], xo)
            assert np.array_equal(x[0, i] != 0, xo != 0)
    print(f"window batch (one launch per iteration) vs oracle.admm_quadratic: worst rel-L2 {worst:.2e}")


def test_window_batch_stopping_iterations_against_oracle(L, oracle, monkeypatch):
    """tol > 0: every window stops at the oracle's iteration (each problem of the batch has its own flag, src/lasso.jl:164)."""
    from lpvspectral_jl_amd import api
    rng = np.random.default_rng(18)
    n, nwin, Nf, t, f, y = _window_case(rng)
    monkeypatch.delenv("LPVS_ITERATION", raising=False)
    tol = 2e-6
    x, its = api.windows_estimate([y], t, f, n, 0, None, _engine(L, 3000, tol))
    assert api.windowpsd_last_timing()["one_launch_iteration"]
    W = np.ones(n)
    for i in range(nwin):
        yi, ti = y[i * n:(i + 1) * n], t[i * n:(i + 1) * n]
        with L.Problem.fourier(yi, ti, f, W) as p:
            Q, q = p.get_gram()
        ro = oracle.admm_quadratic(Q, q, oracle.NormL1(0.2), iters=3000, tol=tol, mu=1e-4, history=True)
        k = ro["iters"]
        assert k < 3000
        # only windows whose stop is unambiguous (the norm is not within 1e-6 relative of the tolerance at the crossing) pin the count
        margin = min(abs(ro["nxz"][k - 1] - tol), abs(ro["nxz"][k - 2] - tol) if k >= 2 else 1.0) / tol
        if margin > 1e-6:
            assert its[0, i] == k, (i, its[0, i], k)
            assert rel(x[0, i], oracle.fourier2complex(ro["z"], 1)) <= 1e-9


def test_window_batch_nontemporal_loads_are_bit_identical(L, monkeypatch):
    """The NT template instance (what a 1024-window batch runs: its 764 MB of inverses exceed the Infinity Cache) on a 6-window batch:
    a load's cache policy must not change a bit."""
    from lpvspectral_jl_amd import api
    rng = np.random.default_rng(19)
    n, nwin, Nf, t, f, y = _window_case(rng)
    out = {}
    for nt in ("0", "1"):
        monkeypatch.setenv("LPVS_NT_LOADS", nt)
        for mode in ("one", "two"):
            if mode == "two":
                monkeypatch.setenv("LPVS_ITERATION", "two")
            else:
                monkeypatch.delenv("LPVS_ITERATION", raising=False)
            x, its = api.windows_estimate([y], t, f, n, 0, None, _engine(L, 300, 0.0))
            assert api.windowpsd_last_timing()["one_launch_iteration"] == (mode == "one")
            out[(nt, mode)] = x
    assert np.array_equal(out[("0", "one")], out[("1", "one")])
    assert np.array_equal(out[("0", "two")], out[("1", "two")])


@pytest.mark.parametrize("kind", ["ball", "group"])
def test_multi_signal_mixed_storage_against_oracle(L, oracle, lpv_case, kind):
    """The cfg5 kernel on the storage cfg5 streams: several channels sharing (X, V, w), the matrix-core tile product
    (symv_tile_mfma_ws_kernel, 4x4x4 MFMA) reading 36-bit fixed-point tiles below the diagonal and float-head tiles on it.  Every channel
    against oracle.admm_gram on the device Gram at equal iteration counts: rel-L2 <= 1e-9, identical support."""
    c = lpv_case
    rng = np.random.default_rng(5)
    ns = 3
    Y = np.stack([c["y"], c["y"][::-1].copy(), 0.5 * c["y"] + np.cos(c["w"][7] * c["X"]) * (1 + c["V"])], axis=1) + 0.01 * rng.standard_normal((c["N"], ns))
    prox, oprox = {"ball": (L.IndBallL0(40), oracle.IndBallL0(40)),
                   "group": (L.SlicedSeparableSum.frequency_groups(c["lams"]["group"], c["Nf"], 2 * c["Nv"]), oracle.GroupL2(c["lams"]["group"], 2 * c["Nv"]))}[kind]
    with L.Problem.lpv_multi(Y, c["X"], c["V"], c["w"], c["Nv"]) as p:
        G, _ = p.get_gram(); B = p.get_rhs()
        p.set_prox(prox)
        p.admm_init(None, μ=0.05, tol=0.0)
        info = p.matvec_info()
        assert info["kernel"] == "symv_tile_mfma_ws_kernel" and "36-bit fixed point" in info["storage"] and info["signals_per_pass"] == 8, info
        it, _, _ = p.admm_run(300)
        x, z, u = p.admm_get()
    assert it == 300 and rel(G, c["G"]) <= 1e-13
    for q in range(ns):
        ro = oracle.admm_gram(G, B[:, q], oprox, iters=300, tol=0.0, mu=0.05)
        errs = {k: rel(v[:, q], ro[k]) for k, v in (("x", x), ("z", z), ("u", u))}
        print(f"multi/{kind} channel {q}: rel-L2 vs oracle x {errs['x']:.2e} z {errs['z']:.2e} u {errs['u']:.2e}; nnz {np.count_nonzero(ro['z'])}")
        assert max(errs.values()) <= 1e-9, (q, errs)
        assert np.array_equal(z[:, q] != 0, ro["z"] != 0) and 0 < np.count_nonzero(ro["z"]) < ro["z"].size
