"""The one-launch ADMM iteration (csrc/admm_one_launch.hip: fixed-point accumulation of the tile partials, update in the next launch's
prologue; the default for single-signal handles with the mixed storage) against the two-launch iteration (LPVS_ITERATION=two):
same iterates to rounding, same stopping iteration, invariant under chunking, re-entry from a saved state, all fusable prox
operators.  GPU only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _signal(N, Nf, rng):
    X = np.sort(rng.random(N) * (10.0 * N / 500)); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25.0 / Nf
    y = 2 * V ** 2 * np.cos(w[Nf // 10] * X) + 2 / (5 * V + 1) * np.cos(w[Nf // 3] * X - 0.3) + 0.1 * rng.standard_normal(N)
    return y, X, V, w


def _proxes(L, Nf, Nv):
    return {"group": L.SlicedSeparableSum.frequency_groups(2.0, Nf, 2 * Nv), "l1": L.NormL1(0.5), "l0": L.NormL0(0.02)}


@pytest.fixture(scope="module")
def problem(L):
    rng = np.random.default_rng(21)
    N, Nf, Nv = 1 << 18, 128, 8                          # n = 2048: 16 row blocks, 136 tiles; enough samples for the mixed storage to hold
    return (N, Nf, Nv) + _signal(N, Nf, rng)


def _run(L, problem, prox, chunks, monkeypatch, mode, tol=0.0, state=None):
    N, Nf, Nv, y, X, V, w = problem
    if mode == "two":
        monkeypatch.setenv("LPVS_ITERATION", "two")
    else:
        monkeypatch.delenv("LPVS_ITERATION", raising=False)
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        p.set_prox(prox)
        p.admm_init(None, μ=0.05, tol=tol)
        if state is not None:                            # (x, z, u, iterations, offset vector of the x-update)
            p.admm_set_state(state[0], state[1], state[2], state[3], offset=state[4] if len(state) > 4 else None)
        info = p.matvec_info()
        assert info["kernel"] == ("symv_tile_mixed_kernel" if mode == "two" else "admm_iter_mixed_kernel"), info
        for c in chunks:
            it, nxz, conv = p.admm_run(c)
            if conv:
                break
        return (it, nxz, conv) + p.admm_get() + (p.admm_get_offset(),)


@pytest.mark.parametrize("kind", ["group", "l1", "l0"])
def test_one_launch_iteration_equals_two_launch_iteration(L, problem, kind, monkeypatch):
    prox = _proxes(L, problem[1], problem[2])[kind]
    a = _run(L, problem, prox, [300], monkeypatch, "two")
    b = _run(L, problem, prox, [300], monkeypatch, "one")
    assert a[0] == b[0] == 300
    for va, vb in zip(a[3:6], b[3:6]):                     # x, z, u
        assert np.abs(va - vb).max() <= 1e-12 * max(np.abs(va).max(), 1.0), np.abs(va - vb).max()
    assert np.array_equal(a[4] != 0, b[4] != 0)          # same support
    assert abs(a[1] - b[1]) <= 1e-10 * a[1]


def test_one_launch_iteration_does_not_depend_on_chunking(L, problem, monkeypatch):
    prox = _proxes(L, problem[1], problem[2])["group"]
    ref = _run(L, problem, prox, [240], monkeypatch, "one")
    for chunks in ([80, 80, 80], [1, 1, 2, 3, 233], [239, 1], [7] * 34 + [2]):
        r = _run(L, problem, prox, chunks, monkeypatch, "one")
        assert r[0] == ref[0] == 240 and r[1] == ref[1]
        for a, b in zip(r[3:6], ref[3:6]):
            assert np.array_equal(a, b)


def test_one_launch_iteration_stops_where_the_two_launch_one_does(L, problem, monkeypatch):
    """tol > 0: same stopping iteration (the convergence test is deferred by one launch in both schemes), the iterates of the converged
    iteration -- whatever the parity of the iteration (the alternate u buffer) and wherever in a chunk it falls."""
    prox = _proxes(L, problem[1], problem[2])["l1"]
    for tol in (3e-3, 2.5e-3, 1e-3):
        a = _run(L, problem, prox, [4000], monkeypatch, "two", tol=tol)
        assert a[2]
        for chunks in ([4000], [a[0] - 1, 50], [a[0], 50], [a[0] + 1, 50], [13] * 400):
            b = _run(L, problem, prox, chunks, monkeypatch, "one", tol=tol)
            assert b[2] and b[0] == a[0], (tol, chunks, a[0], b[0])
            for va, vb in zip(a[3:6], b[3:6]):
                assert np.abs(va - vb).max() <= 1e-12 * max(np.abs(va).max(), 1.0)


def test_one_launch_iteration_resumes_from_a_saved_state(L, problem, monkeypatch):
    prox = _proxes(L, problem[1], problem[2])["group"]
    full = _run(L, problem, prox, [200], monkeypatch, "one")
    half = _run(L, problem, prox, [120], monkeypatch, "one")
    # (x, z, u and the x-update's offset vector, re-formed after iteration 16 and every 512th: part of the state in between)
    rest = _run(L, problem, prox, [80], monkeypatch, "one", state=(half[3], half[4], half[5], 120, half[6]))
    assert rest[0] == 200
    for a, b in zip(rest[3:6], full[3:6]):
        assert np.abs(a - b).max() <= 1e-12 * max(np.abs(b).max(), 1.0)


def test_padded_and_stiff_problems_keep_the_quantum_tight(L, monkeypatch):
    """n not a multiple of 128 (pad rows carry a 1 on the diagonal of the inverse) and a small mu (1 / mu = 10^4): the bound behind the
    fixed-point quantum must not count the pad rows -- a quantum loose by 1 / mu^2 showed as 1e-9 in the iterates.  Single problem and a
    batch of windows, one-launch against two-launch iteration."""
    from lpvspectral_jl_amd import _lib, api
    rng = np.random.default_rng(5)
    N, Nf, Nv = 1 << 18, 127, 8                          # n = 2032 -> np = 2048
    y, X, V, w = _signal(N, Nf, rng)
    res = {}
    for mode in ("two", "one"):
        if mode == "two":
            monkeypatch.setenv("LPVS_ITERATION", "two")
        else:
            monkeypatch.delenv("LPVS_ITERATION", raising=False)
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            p.set_prox(L.NormL1(0.5))
            p.admm_init(None, μ=1e-3, tol=0.0)
            assert p.matvec_info()["kernel"] == ("symv_tile_mixed_kernel" if mode == "two" else "admm_iter_mixed_kernel")
            p.admm_run(300)
            res[mode] = p.admm_get()
    for a, b in zip(res["one"], res["two"]):
        assert np.abs(a - b).max() <= 1e-12 * max(np.abs(b).max(), 1.0), np.abs(a - b).max()
    n, nwin, Nfw = 1 << 14, 6, 256                       # nreg = 511 -> np = 512, four row blocks per window
    t = np.arange(n * nwin, dtype=np.float64)
    f = np.arange(Nfw) / 512.0
    yw = np.sin(2 * np.pi * f[33] * t) + 0.5 * np.sin(2 * np.pi * f[100] * t) + 0.3 * rng.standard_normal(n * nwin)
    eng = dict(estimator=_lib.EST_SPARSE, lam=0.0, prox=(_lib.PROX_L1, 0.2, 0), μ=1e-4, tol=0.0, iters=300, sign=_lib.LINEAR_QUADRATIC_AS_WRITTEN)
    out = {}
    for mode in ("two", "one"):
        if mode == "two":
            monkeypatch.setenv("LPVS_ITERATION", "two")
        else:
            monkeypatch.delenv("LPVS_ITERATION", raising=False)
        out[mode] = api.windows_estimate([yw], t, f, n, 0, None, eng)[0]
        assert api.windowpsd_last_timing()["one_launch_iteration"] == (mode == "one")
    assert np.abs(out["one"] - out["two"]).max() <= 1e-12 * np.abs(out["two"]).max(), np.abs(out["one"] - out["two"]).max() / np.abs(out["two"]).max()


def test_float32_handles_iterate_in_one_launch_too(L, problem, monkeypatch):
    """_f32 handles (single-precision copy of the inverse, double arithmetic): one-launch against two-launch iteration."""
    N, Nf, Nv, y, X, V, w = problem
    f32 = [a.astype(np.float32) for a in (y, X, V, w)]
    res = {}
    for mode in ("two", "one"):
        if mode == "two":
            monkeypatch.setenv("LPVS_ITERATION", "two")
        else:
            monkeypatch.delenv("LPVS_ITERATION", raising=False)
        with L.Problem.lpv(*f32, Nv) as p:
            assert p.f32
            p.set_prox(L.SlicedSeparableSum.frequency_groups(2.0, Nf, 2 * Nv))
            p.admm_init(None, μ=0.05, tol=0.0)
            info = p.matvec_info()
            assert info["kernel"] == ("symv_tile_f32_kernel" if mode == "two" else "admm_iter_mixed_kernel"), info
            it, nxz, conv = p.admm_run(300)
            res[mode] = (it, nxz) + p.admm_get()
    assert res["one"][0] == res["two"][0] == 300
    assert abs(res["one"][1] - res["two"][1]) <= 1e-9 * res["two"][1]
    for a, b in zip(res["one"][2:], res["two"][2:]):                 # (read back as float32)
        assert np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= 2e-7 * max(np.abs(b).max(), 1.0)


def test_more_than_64_row_blocks(L, monkeypatch):
    """np = 12288 (96 row blocks): block norms and maxima take several loads per lane (the NK = 6 instances of the kernel)."""
    rng = np.random.default_rng(8)
    N, Nf, Nv = 1 << 20, 768, 8
    y, X, V, w = _signal(N, Nf, rng)
    res = {}
    for mode in ("two", "one"):
        if mode == "two":
            monkeypatch.setenv("LPVS_ITERATION", "two")
        else:
            monkeypatch.delenv("LPVS_ITERATION", raising=False)
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            p.set_prox(L.SlicedSeparableSum.frequency_groups(2.0, Nf, 2 * Nv))
            p.admm_init(None, μ=0.05, tol=0.0)
            assert p.matvec_info()["kernel"] == ("symv_tile_mixed_kernel" if mode == "two" else "admm_iter_mixed_kernel")
            for c in (60, 1, 59):
                it, nxz, conv = p.admm_run(c)
            res[mode] = (it, nxz) + p.admm_get()
    assert res["one"][0] == res["two"][0] == 120
    for a, b in zip(res["one"][2:], res["two"][2:]):
        assert np.abs(a - b).max() <= 1e-12 * max(np.abs(b).max(), 1.0), np.abs(a - b).max()
    # this inverse (348 MB in the mixed storage) is larger than the Infinity Cache: the one-launch kernel streams it with non-temporal
    # loads by default; with plain loads (option nt_loads="off") the same bits
    with L.default_options(nt_loads="off"):
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            p.set_prox(L.SlicedSeparableSum.frequency_groups(2.0, Nf, 2 * Nv))
            p.admm_init(None, μ=0.05, tol=0.0)
            for c in (60, 1, 59):
                it, nxz, conv = p.admm_run(c)
            plain = (it, nxz) + p.admm_get()
    assert plain[0] == 120 and plain[1] == res["one"][1]
    for a, b in zip(plain[2:], res["one"][2:]):
        assert np.array_equal(a, b)


def test_non_finite_state_propagates_as_nan(L, problem, monkeypatch):
    """A NaN in the state handed to the iteration (x0) must come out as NaN iterates, as in the two-launch iteration and in the
    reference's own arithmetic -- not vanish in the conversion of a tile partial to the fixed-point accumulator."""
    N, Nf, Nv, y, X, V, w = problem
    x0 = np.zeros(2 * Nf * Nv); x0[37] = np.nan
    for mode in ("two", "one"):
        if mode == "two":
            monkeypatch.setenv("LPVS_ITERATION", "two")
        else:
            monkeypatch.delenv("LPVS_ITERATION", raising=False)
        with L.Problem.lpv(y, X, V, w, Nv) as p:
            p.set_prox(L.NormL1(0.5))
            p.admm_init(x0, μ=0.05, tol=0.0)
            assert p.matvec_info()["kernel"] == ("symv_tile_mixed_kernel" if mode == "two" else "admm_iter_mixed_kernel")
            p.admm_run(5)
            x, z, u = p.admm_get()
            assert np.isnan(x).all() and np.isnan(u).all(), (mode, np.isnan(x).sum(), np.isnan(u).sum())
