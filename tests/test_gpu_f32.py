"""Single-precision entry points (_f32): float I/O, double assembly, single-precision copy of M in the ADMM mat-vec.
SURVEY.md section 8(d) tolerance for the fp32 device path: rel-L2(z) <= 1e-3 and identical support against the fp64
oracle on problems whose true amplitudes are well above the threshold; the measured values are asserted much tighter."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a, dtype=np.complex128) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def test_f32_regressors_match_oracle(L, oracle):
    rng = np.random.default_rng(3)
    N, Nf, Nv = 700, 9, 3
    t = np.sort(rng.random(N) * 40).astype(np.float32)
    f = (np.arange(Nf) / 16.0).astype(np.float32)          # zero frequency first
    A, zf = L.get_fourier_regressor(t, f)
    assert A.dtype == np.float32 and zf == 1 and A.shape == (N, 2 * Nf - 1)
    Ao, zo = oracle.get_fourier_regressor(t.astype(np.float64), f.astype(np.float64))
    assert np.abs(A - Ao).max() <= 1e-7                     # one float rounding of values <= 1/sqrt(2Nf)
    X = np.sort(rng.random(N) * 10).astype(np.float32); V = rng.random(N).astype(np.float32)
    w = (2 * np.pi * (np.arange(Nf) + 1.0) / 4).astype(np.float32)
    Phi = L.lpv_regressor(X, V, w, Nv)
    assert Phi.dtype == np.float32
    Po = oracle.lpv_regressor(X.astype(np.float64), V.astype(np.float64), w.astype(np.float64), Nv)
    assert np.abs(Phi - Po).max() <= 1e-7


def test_f32_ls_spectral_and_sparse_fourier(L, oracle):
    rng = np.random.default_rng(11)
    N = 3000
    t = np.sort(rng.random(N) * N).astype(np.float32)
    f = (np.arange(1, 65) / 128.0).astype(np.float32)
    y = (2 * np.sin(2 * np.pi * f[5] * t + 0.3) + np.sin(2 * np.pi * f[20] * t + 1.1) + 0.1 * rng.standard_normal(N)).astype(np.float32)
    y64, t64, f64 = (a.astype(np.float64) for a in (y, t, f))
    x, _ = L.ls_spectral(y, t, f, λ=1e-3)
    assert x.dtype == np.complex64
    xo, _ = oracle.ls_spectral(y64, t64, f64, lam=1e-3)
    assert rel(x, xo) <= 1e-5
    import io
    p, _ = L.ls_sparse_spectral(y, t, f, λ=20.0, iters=400, tol=0.0, printerval=1000, out=io.StringIO())
    assert p.dtype == np.complex64
    po, _, _ = oracle.ls_sparse_spectral(y64, t64, f64, lam=20.0, iters=400, tol=0.0)
    assert rel(p, po) <= 1e-5                               # n = 128 < 2048: M stays double, only the I/O is single
    assert np.array_equal(np.abs(p) > 0, np.abs(po) > 0)


def test_f32_lpv_group_lasso_streams_single_precision_matrix(L, oracle):
    """n = 2048 takes the tile-packed path, so M is streamed in single precision: measured rel-L2 4e-6."""
    rng = np.random.default_rng(21)
    N, Nf, Nv = 5000, 128, 8
    X = np.sort(rng.random(N) * 10 * N / 500).astype(np.float32)
    V = np.linspace(0, 1, N).astype(np.float32)
    w = (2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf / 4).astype(np.float32)
    y = (2 * V ** 2 * np.cos(w[12] * X) + 2 / (5 * V + 1) * np.cos(w[60] * X) + 0.1 * rng.standard_normal(N)).astype(np.float32)
    import io
    se = L.ls_sparse_spectral_lpv(y, X, V, w, Nv, λ=3.0, iters=150, tol=0.0, printerval=1000, out=io.StringIO())
    assert se.x.dtype == np.complex64
    y64, X64, V64, w64 = (a.astype(np.float64) for a in (y, X, V, w))
    # the checker is the fp64 CPU oracle on the (exactly widened) float inputs, not another device path
    Phi = oracle.lpv_regressor(X64, V64, w64, Nv)
    Go, bo = oracle.gram(Phi, y64)
    ro = oracle.admm_gram(Go, bo, oracle.GroupL2(3.0, 2 * Nv), iters=150, tol=0.0, mu=0.05)
    xo = oracle.lpv_unpermute(ro["z"], Nf, Nv)
    r = rel(se.x, xo)
    assert r <= 2e-5, r                                     # measured 4.1e-6 against the f64 device path; SURVEY tolerance is 1e-3
    assert np.array_equal(np.abs(se.x) > 0, np.abs(xo) > 0)
    ref = L.ls_sparse_spectral_lpv(y64, X64, V64, w64, Nv, λ=3.0, iters=150, tol=0.0, printerval=1000, out=io.StringIO())
    assert rel(ref.x, xo) <= 1e-9                           # and the f64 device path on the same inputs
    with L.Problem.lpv(y, X, V, w, Nv) as p:                # the handle reports single-precision bytes per mat-vec
        p.set_prox(L.SlicedSeparableSum.frequency_groups(3.0, Nf, 2 * Nv))
        p.admm_init(None, μ=0.05, tol=0.0)
        us, nbytes = p.time_matvec(20)
        x, z, u = p.admm_get()
    assert nbytes == 4 * (2048 * (2048 + 128) // 2) and x.dtype == np.float32


def test_f32_float_grid_takes_the_structured_gram(L, oracle):
    """A frequency grid stored in floats is a progression only up to float rounding (residual phases |eps x| far above the
    double-precision admission bound).  _f32 handles admit it: the grid is snapped to the exact progression its floats were
    rounded from -- a phase change of at most 2^-24 |w x|, which is what a Float32 run of the reference commits in
    fl32(w x) anyway.  Checked against the fp64 oracle evaluated ON THE SNAPPED GRID (float I/O tolerance), and against
    the un-snapped grid with the tolerance the snapping implies."""
    import io
    rng = np.random.default_rng(41)
    N, Nf, Nv = 3000, 16, 4
    X = np.sort(rng.random(N) * 5000).astype(np.float32)              # |eps x| up to 157 * 6e-8 * 5000 = 0.05 rad: snap regime
    V = np.linspace(0, 1, N).astype(np.float32)
    w = (2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf).astype(np.float32)
    y = (2 * V ** 2 * np.cos(w[3].astype(np.float64) * X) + np.cos(w[11].astype(np.float64) * X + 0.4) + 0.1 * rng.standard_normal(N)).astype(np.float32)
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        assert p.timing()["gram_form"] in ("ap", "ap-nufft") and p.f32
    with L.Problem.lpv(y.astype(np.float64), X.astype(np.float64), V.astype(np.float64), w.astype(np.float64), Nv) as p:
        assert p.timing()["gram_form"] == "krs"                      # the same grid through the _f64 entry point: not admitted
    se = L.ls_sparse_spectral_lpv(y, X, V, w, Nv, λ=3.0, iters=200, tol=0.0, printerval=1000, out=io.StringIO())
    y64, X64, V64, w64 = (a.astype(np.float64) for a in (y, X, V, w))
    wl = w64.astype(np.longdouble)
    snapped = np.asarray(wl[0] + np.arange(Nf, dtype=np.longdouble) * (wl[-1] - wl[0]) / (Nf - 1), dtype=np.float64)
    assert 1e-6 < np.abs(snapped - w64).max() * X64.max() < 0.2
    def run(wg):
        Phi = oracle.lpv_regressor(X64, V64, wg, Nv)
        Go, bo = oracle.gram(Phi, y64)
        ro = oracle.admm_gram(Go, bo, oracle.GroupL2(3.0, 2 * Nv), iters=200, tol=0.0, mu=0.05)
        return oracle.lpv_unpermute(ro["z"], Nf, Nv)
    xs, xu = run(snapped), run(w64)
    assert rel(se.x, xs) <= 2e-5, rel(se.x, xs)                       # float I/O only
    assert np.array_equal(np.abs(se.x) > 0, np.abs(xs) > 0)
    print(f"f32 float grid: rel-L2 vs oracle on the snapped grid {rel(se.x, xs):.2e}; the snapping itself moves the oracle by {rel(xs, xu):.2e}")


def test_f32_multi_signal_and_window_engine_entry_points(L, oracle):
    """lpvs_problem_create_lpv_multi_f32 and lpvs_windows_estimate_f32: float in / out, double arithmetic, against the fp64
    oracle on the widened inputs."""
    import io
    rng = np.random.default_rng(51)
    N, Nf, Nv, ns = 1500, 12, 4, 3
    X = np.sort(rng.random(N) * 10).astype(np.float32); V = np.linspace(0, 1, N).astype(np.float32)
    w = (2 * np.pi * (np.arange(Nf) + 1.0) * 2).astype(np.float32)
    Y = np.stack([np.cos(w[(3 * q + 1) % Nf].astype(np.float64) * X) * (1 + q * V) + 0.05 * rng.standard_normal(N) for q in range(ns)], axis=1).astype(np.float32)
    ses = L.ls_sparse_spectral_lpv_multi(Y, X, V, w, Nv, λ=2.0, iters=150, tol=0.0, printerval=1000, out=io.StringIO())
    X64, V64, w64 = (a.astype(np.float64) for a in (X, V, w))
    Phi = oracle.lpv_regressor(X64, V64, w64, Nv)
    for q in range(ns):
        Go, bo = oracle.gram(Phi, Y[:, q].astype(np.float64))
        ro = oracle.admm_gram(Go, bo, oracle.GroupL2(2.0, 2 * Nv), iters=150, tol=0.0, mu=0.05)
        xo = oracle.lpv_unpermute(ro["z"], Nf, Nv)
        assert ses[q].x.dtype == np.complex64 and rel(ses[q].x, xo) <= 2e-5 and np.array_equal(np.abs(ses[q].x) > 0, np.abs(xo) > 0)
    # windows: csd of two float records with the dense and the sparse estimator
    Lh = 4000
    t = np.cumsum(0.5 + rng.random(Lh)).astype(np.float32)
    f = (np.arange(1, 33) / 80.0).astype(np.float32)
    t64, f64 = t.astype(np.float64), f.astype(np.float64)
    y = (np.sin(2 * np.pi * f64[8] * t64) + 0.1 * rng.standard_normal(Lh)).astype(np.float32)
    u = (0.7 * np.sin(2 * np.pi * f64[8] * t64 + 0.5) + 0.1 * rng.standard_normal(Lh)).astype(np.float32)
    S, _ = L.ls_windowcsd(y, u, t, f, nw=8, noverlap=0, λ=1e-3)
    So, _ = oracle.ls_windowcsd(y.astype(np.float64), u.astype(np.float64), t64, f64, nw=8, noverlap=0, lam=1e-3)
    assert rel(S, So) <= 1e-4, rel(S, So)                          # float outputs of the engine, double accumulation over the windows
    Ss, _ = L.ls_windowpsd(y, t, f, nw=8, noverlap=0, estimator=L.ls_sparse_spectral, λ=0.5, μ=0.05, tol=1e-9, iters=2000)
    Sso, _ = oracle.ls_windowpsd(y.astype(np.float64), t64, f64, nw=8, noverlap=0,
                                 estimator=lambda yy, tt, ff, W, **k: oracle.ls_sparse_spectral(yy, tt, ff, W, **k), lam=0.5, mu=0.05, tol=1e-9, iters=2000)
    assert rel(Ss, Sso) <= 1e-4 and int(np.argmax(Ss)) == 8


def test_f32_remaining_entry_points(L):
    """lpvs_problem_create_lpv_rows_f32, lpvs_problem_solve_ridge_f32, lpvs_admm_set_state_f32, lpvs_windows_estimate_multi_f32,
    lpvs_windowcsd_f32: float in / out around the same double arithmetic as their _f64 twins on the widened inputs."""
    rng = np.random.default_rng(52)
    N, Nf, Nv = 1500, 12, 4
    X = np.sort(rng.random(N) * 10).astype(np.float32); V = np.linspace(0, 1, N).astype(np.float32)
    w = (2 * np.pi * (np.arange(Nf) + 1.0) * 2).astype(np.float32)
    y = (np.cos(w[3].astype(np.float64) * X) * (1 + V) + 0.05 * rng.standard_normal(N)).astype(np.float32)
    X64, V64, w64, y64 = (a.astype(np.float64) for a in (X, V, w, y))
    prox = L.SlicedSeparableSum.frequency_groups(2.0, Nf, 2 * Nv)
    # row-sharded constructor with the whole record as its one shard == the plain constructor
    with L.Problem.lpv_rows(y, X, V, w, Nv, L.lpv_ranges(X64, V64)) as pr, L.Problem.lpv(y, X, V, w, Nv) as pl, \
            L.Problem.lpv(y64, X64, V64, w64, Nv) as pd:
        assert pr.f32 and pl.f32 and not pd.f32
        for p in (pr, pl, pd):
            p.set_prox(prox)
            p.admm_init(None, μ=0.05, tol=0.0)
            p.admm_run(60)
        xr, zr, ur = pr.admm_get(); xl, zl, ul = pl.admm_get(); xd, zd, ud = pd.admm_get()
        assert xr.dtype == np.float32 and np.array_equal(zr, zl)
        assert rel(zr, zd) <= 2e-5 and np.array_equal(zr != 0, zd != 0)
        # re-entry from float iterates: 60 + 40 iterations == 100 iterations up to the float rounding of the saved state
        pl.admm_init(None, μ=0.05, tol=0.0)
        pl.admm_set_state(xl, zl, ul, iters=60)
        it, _, _ = pl.admm_run(40)
        pd.admm_run(40)
        assert it == 100
        assert rel(pl.admm_get()[1], pd.admm_get()[1]) <= 5e-5
        # dense estimator on the float handle
        a32, a64 = pl.solve_ridge(1e-3), pd.solve_ridge(1e-3)
        assert a32.dtype == np.float32 and rel(a32, a64) <= 2e-5
    # windows on several devices driven by one process (here: one device), float records
    Lh = 4000
    t = np.cumsum(0.5 + rng.random(Lh)).astype(np.float32)
    f = (np.arange(1, 33) / 80.0).astype(np.float32)
    t64, f64 = t.astype(np.float64), f.astype(np.float64)
    ys = (np.sin(2 * np.pi * f64[8] * t64) + 0.1 * rng.standard_normal(Lh)).astype(np.float32)
    us = (0.7 * np.sin(2 * np.pi * f64[8] * t64 + 0.5) + 0.1 * rng.standard_normal(Lh)).astype(np.float32)
    eng = dict(estimator=1, lam=0.0, prox=(1, 0.5, 0), μ=0.05, tol=1e-9, iters=500, sign=-1)
    W = L.hanning(500).astype(np.float32)
    xm, itm = L.windows_estimate_multi([ys, us], t, f, 500, 0, W, eng, ngpus=1)
    xs, its = L.windows_estimate([ys, us], t, f, 500, 0, W, eng)
    xd, itd = L.windows_estimate([ys.astype(np.float64), us.astype(np.float64)], t64, f64, 500, 0, W.astype(np.float64), eng)
    assert xm.dtype == np.complex64 and np.array_equal(xm, xs) and np.array_equal(itm, its)
    assert rel(xm, xd) <= 2e-5
    Syu, Syy, Suu, xy, xu = L.windowcsd_batched(ys, us, t, f, 500, 0, W, eng)
    assert Syy.dtype == np.float32 and np.array_equal(xy, xs[0]) and np.array_equal(xu, xs[1])


def test_f32_window_calls_take_the_chunked_plan(L):
    """the single-device _f32 window entry points widen freqs into a device buffer, which used to keep them off
    the engine's chunked plan (cache-sized chunks, parts in flight), and the float-grid admission -- a thread-local switch -- did not
    travel to the parts' worker threads.  40 windows, float grid: the default plan, small chunks with three parts in flight and the
    uncut plan agree bit for bit, and the structured Gram is taken on the worker threads too."""
    from lpvspectral_jl_amd import api
    rng = np.random.default_rng(61)
    n, nwin, Nf = 1 << 10, 40, 64
    t = np.arange(n * nwin, dtype=np.float32)
    f = (np.arange(Nf) / 128.0).astype(np.float32)          # a float grid: admitted as the progression it was rounded from
    y = (np.sin(2 * np.pi * f[9].astype(np.float64) * t.astype(np.float64)) + 0.3 * rng.standard_normal(n * nwin)).astype(np.float32)
    eng = dict(estimator=1, lam=0.0, prox=(1, 0.3, 0), μ=1e-3, tol=0.0, iters=100, sign=-1)
    out = {}
    for name, opts in (("uncut", dict(window_chunk_mb="uncut", windows_in_flight=1)), ("default", dict()), ("small", dict(window_chunk_mb=1, windows_in_flight=3))):
        with L.default_options(**opts):
            x, its = L.windows_estimate([y], t, f, n, 0, None, eng)
        tm = api.windowpsd_last_timing()
        assert x.dtype == np.complex64 and tm["gram_form"] in ("ap", "ap-nufft"), (name, tm)
        out[name] = x
    assert np.array_equal(out["default"], out["uncut"]) and np.array_equal(out["small"], out["uncut"])


def test_f32_record_with_f64_time_stamps_keeps_t_exact(L):
    """Eltype promotion as Julia does it (src/lsfft.jl:121 -> src/lasso.jl:111 -> src/lsfft.jl:26: the regressor is evaluated in the
    eltype of t and freqs): a Float32 record with Float64 time stamps above 2^24 is a Float64 problem -- identical to the widened
    record, and different from what rounding t to 24 bits would give (time stamps collapsing onto duplicates)."""
    rng = np.random.default_rng(71)
    Lh, nw = 1 << 25, 8
    t = np.arange(Lh, dtype=np.float64)                               # sample indices up to 2^25: odd ones are not floats
    f = np.arange(1, 25) / 64.0
    y = (np.sin(2 * np.pi * f[5] * t + 0.3) + 0.2 * rng.standard_normal(Lh)).astype(np.float32)
    kw = dict(nw=nw, noverlap=0, estimator=L.ls_sparse_spectral, λ=0.1, μ=1e-4, iters=200, tol=0.0)
    S32, _ = L.ls_windowpsd(y, t, f, **kw)
    S64, _ = L.ls_windowpsd(y.astype(np.float64), t, f, **kw)
    assert S32.dtype == np.float64 and np.array_equal(S32, S64)
    Sr, _ = L.ls_windowpsd(y, t.astype(np.float32), f.astype(np.float32), **kw)      # an all-Float32 call: the _f32 entry points
    assert rel(Sr, S64) > 1e-6                                        # (that IS a different problem: t rounded to 24 bits)
    # the dense estimator and the cross-spectral driver promote the same way
    P32, _ = L.ls_windowpsd(y, t, f, nw=nw, noverlap=0, λ=1e-6)
    P64, _ = L.ls_windowpsd(y.astype(np.float64), t, f, nw=nw, noverlap=0, λ=1e-6)
    assert np.array_equal(P32, P64)
    C32, _ = L.ls_windowcsd(y, y, t, f, nw=nw, noverlap=0, λ=1e-6)
    C64, _ = L.ls_windowcsd(y.astype(np.float64), y.astype(np.float64), t, f, nw=nw, noverlap=0, λ=1e-6)
    assert np.array_equal(C32, C64)
    # default_freqs of Float32 time stamps is a Float32 grid (rfftfreq(n, fs::Float32)); of Float64 ones a Float64 grid
    assert L.default_freqs(t[:1000].astype(np.float32)).dtype == np.float32 and L.default_freqs(t[:1000]).dtype == np.float64
