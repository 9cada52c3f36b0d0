"""The batched-window engine behind ls_windowpsd / ls_windowcsd / ls_cohere (src/lsfft.jl:112-193) against the CPU oracle's
restatement of those drivers: ns signals share every window's Gram and factorisation.  GPU only.

Tolerances: dense estimator (normal equations on both sides) rel <= 1e-8; sparse estimator vs the oracle's faithful CG form
rel <= 1e-6 (CG's own tolerance), identical support and stopping iterations vs the single-window device path."""
import numpy as np
import pytest

from _guards import precondition_not_met

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def two_signals(Lh, seed, zero):
    rng = np.random.default_rng(seed)
    t = np.cumsum(0.5 + rng.random(Lh))                       # non-equidistant
    f = np.arange(0 if zero else 1, 40) / 100.0
    common = 1.5 * np.sin(2 * np.pi * 0.11 * t)
    y = common + 0.7 * np.cos(2 * np.pi * 0.29 * t + 0.4) + 0.1 * rng.standard_normal(Lh) + (0.5 if zero else 0)
    u = 0.8 * common + 0.5 * np.sin(2 * np.pi * 0.21 * t + 1.0) + 0.1 * rng.standard_normal(Lh)
    return y, u, t, f


def test_known_answers_through_the_engine(L):
    """test/runtests.jl:203-208 on the batched path (default estimator = ls_spectral, the weighted 4-argument method)."""
    t = np.arange(1000) * 0.1
    y = np.sin(2 * np.pi * t)
    x, freqs = L.ls_windowcsd(y, y, t, noverlap=0)
    a = np.abs(x)
    assert abs(a.max() - 2.0 * len(freqs)) < 1e-4 and a.argmax() + 1 == 11
    c, _ = L.ls_cohere(y, y, t)
    assert np.all(c == 1)
    S, fr = L.ls_windowpsd(y, t, noverlap=0)
    assert S.argmax() + 1 == 13 and len(fr) == 63
    S16, _ = L.ls_windowpsd(y, t, nw=16, noverlap=0)
    assert np.abs(S16).argmax() + 1 == 7
    rng = np.random.default_rng(0)
    c, _ = L.ls_cohere(y, y + 0.5 * rng.standard_normal(1000), t, nw=8, noverlap=-1)
    assert abs(c.max() - 1.0) < 0.15 and abs(int(c.argmax()) + 1 - 14) <= 1 and c.mean() < 0.25     # :210-213


@pytest.mark.parametrize("noverlap,zero", [(0, True), (100, False), (-1, False)])
def test_csd_cohere_psd_dense_estimator_vs_oracle(L, oracle, noverlap, zero):
    y, u, t, f = two_signals(4000, 8, zero)
    kw = dict(nw=8, noverlap=noverlap)
    S, _ = L.ls_windowcsd(y, u, t, f, window_func=L.hanning, λ=1e-3, **kw)
    So, _ = oracle.ls_windowcsd(y, u, t, f, window_func=oracle.hanning, lam=1e-3, **kw)
    assert rel(S, So) <= 1e-8, rel(S, So)
    Sseq, _ = L.ls_windowcsd(y, u, t, f, window_func=L.hanning, λ=1e-3, batched=False, **kw)
    assert rel(S, Sseq) <= 1e-10
    c, _ = L.ls_cohere(y, u, t, f, λ=1e-3, **kw)
    co, _ = oracle.ls_cohere(y, u, t, f, lam=1e-3, **kw)
    assert rel(c, co) <= 1e-7 and 0 <= c.min() and c.max() <= 1 + 1e-12
    assert abs(f[int(np.argmax(c))] - 0.11) <= 0.011          # the shared component
    P, _ = L.ls_windowpsd(y, t, f, window_func=L.hanning, λ=1e-3, **kw)
    Po, _ = oracle.ls_windowpsd(y, t, f, window_func=oracle.hanning, lam=1e-3, **kw)
    assert rel(P, Po) <= 1e-8


@pytest.mark.parametrize("noverlap,zero", [(0, True), (100, False)])
def test_csd_cohere_sparse_estimator_vs_oracle(L, oracle, noverlap, zero):
    y, u, t, f = two_signals(4000, 9, zero)
    kw = dict(nw=8, noverlap=noverlap)
    est = dict(λ=0.5, μ=0.05, tol=1e-9, iters=3000)
    oest = dict(lam=0.5, mu=0.05, tol=1e-9, iters=3000)
    osp = lambda yy, tt, ff, W, **k: oracle.ls_sparse_spectral(yy, tt, ff, W, **k)
    S, _ = L.ls_windowcsd(y, u, t, f, window_func=L.hanning, estimator=L.ls_sparse_spectral, **est, **kw)
    So, _ = oracle.ls_windowcsd(y, u, t, f, window_func=oracle.hanning, estimator=osp, **oest, **kw)
    assert rel(S, So) <= 1e-6, rel(S, So)
    assert np.array_equal(S != 0, So != 0)
    c, _ = L.ls_cohere(y, u, t, f, estimator=L.ls_sparse_spectral, **est, **kw)
    with np.errstate(invalid="ignore", divide="ignore"):
        co, _ = oracle.ls_cohere(y, u, t, f, estimator=osp, **oest, **kw)
    assert np.array_equal(np.isnan(c), np.isnan(co))            # 0/0 where neither signal has the frequency, as in the reference
    ok = ~np.isnan(co)
    assert ok.sum() >= 1 and rel(c[ok], co[ok]) <= 1e-6
    # sequential device path (the reference's loop calling the 4-argument estimator twice per window)
    Sseq, _ = L.ls_windowcsd(y, u, t, f, window_func=L.hanning, estimator=L.ls_sparse_spectral, batched=False, printerval=100000, **est, **kw)
    assert rel(S, Sseq) <= 1e-9                                 # engine: 6-byte copy of the inverses; handles of this size: doubles


def test_engine_shared_gram_equals_separate_runs_and_shards(L, monkeypatch):
    """ns = 3 signals through one engine call == three single-signal calls, bit for bit (same kernels, same order); disjoint
    window ranges reproduce the whole; accumulators are the in-order sums of the per-window products.  (Bit for bit needs the
    same storage of the packed inverses on both sides: single-signal batches default to the mixed storage, batches with several
    signals per window to uniform 6-byte elements -- LPVS_M_STORAGE=split makes both uniform; test_mixed_vs_split_windows below
    holds the default against it.)"""
    monkeypatch.setenv("LPVS_M_STORAGE", "split")
    rng = np.random.default_rng(12)
    Lh, n, noverlap = 6000, 750, 250
    t = np.cumsum(0.5 + rng.random(Lh))
    f = np.arange(0, 48) / 120.0
    Y = [np.sin(2 * np.pi * f[5 + 7 * q] * t + q) + 0.2 * rng.standard_normal(Lh) for q in range(3)]
    W = L.hanning(n)
    eng_s = dict(estimator=1, lam=0.0, prox=(1, 0.4, 0), μ=0.05, tol=1e-8, iters=2000, sign=-1)
    eng_d = dict(estimator=2, lam=1e-4, prox=(1, 0.0, 0), μ=0.05, tol=0.0, iters=0, sign=1)
    for eng in (eng_s, eng_d):
        x3, it3 = L.windows_estimate(Y, t, f, n, noverlap, W, eng)
        k = x3.shape[1]
        assert x3.shape == (3, k, len(f)) and k == (Lh - n) // (n - noverlap) + 1
        for q in range(3):
            x1, it1 = L.windows_estimate([Y[q]], t, f, n, noverlap, W, eng)
            assert np.array_equal(x1[0], x3[q]) and np.array_equal(it1[0], it3[q])
        if eng is eng_s:
            assert len(set(it3.ravel())) > 1 and it3.max() < 2000            # every problem stops at its own iteration
        Syu, Syy, Suu, xy, xu = L.windowcsd_batched(Y[0], Y[1], t, f, n, noverlap, W, eng)
        assert np.array_equal(xy, x3[0]) and np.array_equal(xu, x3[1])
        acc = np.zeros(len(f), dtype=complex); ayy = np.zeros(len(f)); auu = np.zeros(len(f))
        for i in range(k):
            a, b = xy[i], xu[i]
            acc = acc + ((a.real * b.real + a.imag * b.imag) + 1j * (a.imag * b.real - a.real * b.imag))
            ayy += a.real * a.real + a.imag * a.imag
            auu += b.real * b.real + b.imag * b.imag
        assert np.array_equal(Syu, acc) and np.array_equal(Syy, ayy) and np.array_equal(Suu, auu)
        parts = [L.windowcsd_batched(Y[0], Y[1], t, f, n, noverlap, W, eng, win_lo=lo, win_hi=hi) for lo, hi in ((0, 3), (3, 4), (4, k))]
        assert np.array_equal(np.vstack([p_[3] for p_ in parts]), xy) and np.array_equal(np.vstack([p_[4] for p_ in parts]), xu)
    tm = L.windowpsd_last_timing()
    assert tm["windows"] == k - 4 and tm["passes"] >= 1


def test_engine_sparse_matches_single_window_handles(L, monkeypatch):
    """One window of the sparse engine (two right-hand sides) against the single-problem handle path with the same
    Quadratic(Q, +q) convention: identical stopping iteration, coefficients to summation order."""
    y, u, t, f = two_signals(3000, 3, True)
    n, W = 1000, L.hanning(1000)
    eng = dict(estimator=1, lam=0.0, prox=(1, 0.5, 0), μ=0.05, tol=1e-9, iters=3000, sign=-1)
    x, its = L.windows_estimate([y, u], t, f, n, 0, W, eng)
    monkeypatch.setenv("LPVS_M_STORAGE", "f64")                      # doubles in the engine too: only the summation order differs
    x8, its8 = L.windows_estimate([y, u], t, f, n, 0, W, eng)
    monkeypatch.delenv("LPVS_M_STORAGE")
    assert np.array_equal(its8, its)
    for q, sig in enumerate((y, u)):
        for i in range(3):
            with L.Problem.fourier(sig[i * n:(i + 1) * n], t[i * n:(i + 1) * n], f, W) as p:
                p.set_prox(L.NormL1(0.5))
                p.admm_init(None, μ=0.05, tol=1e-9, linear_sign=-1)
                it, _, conv = p.admm_run(3000)
                xi = p.params(0)
            assert conv and it == its[q, i]
            assert rel(x[q, i], xi) <= 1e-9 and rel(x8[q, i], xi) <= 1e-12


def test_multi_device_driver_shards_reproduce_single_device(L, monkeypatch):
    """lpvs_windows_estimate_multi_f64 (one host process, a thread per device shard).  On the 1-GPU box the shards share the
    device (rehearsal switch): 3 ragged shards of 7 windows == the single-device engine, bit for bit; and the single-rank
    RCCL all-gather (forced) exercises the run-time binding of librccl and returns the same bytes."""
    rng = np.random.default_rng(21)
    Lh, n, noverlap = 5600, 800, 0
    t = np.cumsum(0.5 + rng.random(Lh))
    f = np.arange(1, 40) / 100.0
    Y = [np.sin(2 * np.pi * f[9] * t) + 0.2 * rng.standard_normal(Lh), np.cos(2 * np.pi * f[9] * t + 0.3) + 0.2 * rng.standard_normal(Lh)]
    W = L.hanning(n)
    for eng in (dict(estimator=1, lam=0.0, prox=(1, 0.4, 0), μ=0.05, tol=1e-8, iters=1500, sign=-1),
                dict(estimator=2, lam=1e-4, prox=(1, 0.0, 0), μ=0.05, tol=0.0, iters=0, sign=1)):
        x1, it1 = L.windows_estimate(Y, t, f, n, noverlap, W, eng)
        assert x1.shape[1] == 7
        monkeypatch.setenv("LPVS_MULTI_ALLOW_SHARED_DEVICE", "1")
        x3, it3 = L.windows_estimate_multi(Y, t, f, n, noverlap, W, eng, devices=[0, 0, 0])
        monkeypatch.delenv("LPVS_MULTI_ALLOW_SHARED_DEVICE")
        assert np.array_equal(x3, x1) and np.array_equal(it3, it1)
        xa, ita = L.windows_estimate_multi(Y, t, f, n, noverlap, W, eng, ngpus=1)
        assert np.array_equal(xa, x1) and np.array_equal(ita, it1)
        monkeypatch.setenv("LPVS_MULTI_FORCE_RCCL", "1")
        xr, itr = L.windows_estimate_multi(Y, t, f, n, noverlap, W, eng, ngpus=1)
        monkeypatch.delenv("LPVS_MULTI_FORCE_RCCL")
        assert np.array_equal(xr, x1) and np.array_equal(itr, it1)
    with pytest.raises(ValueError):
        L.windows_estimate_multi(Y, t, f, n, noverlap, W, eng, devices=[0, 0])       # duplicates need the rehearsal switch
    S1, _ = L.ls_windowcsd(Y[0], Y[1], t, f, nw=7, noverlap=0, window_func=L.hanning, λ=1e-4)
    S2, _ = L.ls_windowcsd(Y[0], Y[1], t, f, nw=7, noverlap=0, window_func=L.hanning, λ=1e-4, ngpus=0)   # all visible devices
    assert np.array_equal(S1, S2)


def test_mixed_vs_split_windows(L, monkeypatch):
    """The default (mixed) storage of single-signal window batches against uniform 6-byte elements and doubles: same stopping
    iterations, coefficients within the parity bound."""
    rng = np.random.default_rng(12)
    Lh, n, noverlap = 6000, 750, 250
    t = np.cumsum(0.5 + rng.random(Lh))
    f = np.arange(0, 48) / 120.0
    y = np.sin(2 * np.pi * f[5] * t) + 0.2 * rng.standard_normal(Lh)
    W = L.hanning(n)
    eng = dict(estimator=1, lam=0.0, prox=(1, 0.4, 0), μ=0.05, tol=1e-8, iters=2000, sign=-1)
    out = {}
    for st in ("mixed", "split", "f64"):
        monkeypatch.setenv("LPVS_M_STORAGE", st)
        out[st] = L.windows_estimate([y], t, f, n, noverlap, W, eng)
    for st in ("mixed", "split"):
        assert np.array_equal(out[st][1], out["f64"][1])
        assert rel(out[st][0], out["f64"][0]) <= 1e-9, (st, rel(out[st][0], out["f64"][0]))


def test_window_batches_read_32_bits_with_the_stale_nibble_product(L, monkeypatch):
    """Round 5: where a window batch iterates in one launch, that launch reads 32 of the 36 bits of the fixed-point tiles and the product of
    the 4-bit planes rides, up to 32 iterations stale, in every window's offset vector (the launches after which a refresh is due multiply
    the planes too: admm_iter_mixed_kernel<..., BATCH, NIBR>).  An irregular grid (the inverses are NOT nearly diagonal), three row blocks per
    window: the default against 36-bit reads (storage "mixed" by name) and doubles; LPVS_NIB_PERIOD=0 switches it off; bit-reproducible."""
    from lpvspectral_jl_amd import api
    rng = np.random.default_rng(44)
    n, nwin, Nf = 1500, 8, 160
    t = np.cumsum(0.5 + rng.random(n * nwin))
    f = (np.arange(Nf) + 1.0) / (2.6 * Nf)
    y = np.sin(2 * np.pi * f[11] * t) + 0.6 * np.cos(2 * np.pi * f[90] * t) + 0.3 * rng.standard_normal(n * nwin)
    eng = dict(estimator=1, lam=0.0, prox=(1, 1.5, 0), μ=0.05, tol=0.0, iters=600, sign=-1)
    monkeypatch.delenv("LPVS_ITERATION", raising=False)
    out, flags = {}, {}
    for name, st, env in (("default", None, {}), ("again", None, {}), ("mixed32", "mixed32", {}), ("mixed", "mixed", {}), ("f64", "f64", {}), ("period 0", None, {"LPVS_NIB_PERIOD": "0"})):
        if st is None:
            monkeypatch.delenv("LPVS_M_STORAGE", raising=False)
        else:
            monkeypatch.setenv("LPVS_M_STORAGE", st)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out[name] = api.windows_estimate([y], t, f, n, 0, None, eng)[0]
        flags[name] = api.windowpsd_last_timing()
        for k in env:
            monkeypatch.delenv(k)
    if not flags["mixed"]["one_launch_iteration"]:
        precondition_not_met("this batch does not iterate in one launch: " + str(flags["mixed"]))
    assert flags["default"]["reads_32_bits"] and flags["mixed32"]["reads_32_bits"]
    assert not flags["mixed"]["reads_32_bits"] and not flags["period 0"]["reads_32_bits"] and not flags["f64"]["one_launch_iteration"]
    assert np.array_equal(out["default"], out["again"]) and np.array_equal(out["default"], out["mixed32"]) and np.array_equal(out["mixed"], out["period 0"])
    assert 0 < np.count_nonzero(out["f64"]) < out["f64"].size
    e32, e36, d = rel(out["default"], out["f64"]), rel(out["mixed"], out["f64"]), rel(out["default"], out["mixed"])
    print(f"8 windows, np = 384, irregular grid, 600 iterations: 32-bit reads + stale nibble product vs doubles {e32:.2e}, 36-bit reads vs doubles {e36:.2e}, between them {d:.2e}")
    assert e32 <= 1e-9 and e36 <= 1e-9 and d <= 1e-10
    assert np.array_equal(out["default"] != 0, out["f64"] != 0)


@pytest.mark.parametrize("zero", [True, False])
def test_indball_and_init_through_the_engine_vs_oracle(L, oracle, zero):
    """The two estimator options that used to fall back to the sequential loop now run in the batch: proxg = IndBallL0(r)
    (README.md:79-83: one workgroup per window selects its r largest) and init = true (src/lasso.jl:112: x0 = fourier_solve(A, y,
    zerofreq, lam), the UNWEIGHTED ridge solution whatever the window function).  Per window against oracle.admm_quadratic on the
    window's own Q, q (read back from a single-window handle) started from the oracle's own fourier_solve, and against the
    sequential device loop."""
    from lpvspectral_jl_amd import api
    y, u, t, f = two_signals(4000, 21, zero)
    n, noverlap, k = 500, 0, 8
    W = L.hanning(n)
    nreg = 2 * len(f) - (1 if zero else 0)
    for name, proxg, oproxg, init in (("ball", L.IndBallL0(6), oracle.IndBallL0(6), False), ("ball+init", L.IndBallL0(6), oracle.IndBallL0(6), True),
                                      ("l1+init", L.NormL1(0.5), oracle.NormL1(0.5), True)):
        kw = dict(λ=0.7, proxg=proxg, μ=0.05, tol=0.0, iters=400, init=init)
        eng = api._engine_args(L.ls_sparse_spectral, kw, nreg)
        assert eng is not None and eng["estimator"] == (3 if init else 1), eng
        x, its = api.windows_estimate([y], t, f, n, noverlap, W, eng)
        assert x.shape == (1, k, len(f)) and np.all(its == 400)
        for i in range(k):
            yi, ti = y[i * n:(i + 1) * n], t[i * n:(i + 1) * n]
            with L.Problem.fourier(yi, ti, f, W) as p:
                Q, q = p.get_gram()
            x0 = None
            if init:
                A, zf = oracle.get_fourier_regressor(ti, f)
                p0 = oracle.fourier_solve(A, yi, zf, 0.7)                     # UNWEIGHTED, as written
                x0 = np.concatenate([p0.real, p0.imag[1:] if zf else p0.imag])
            ro = oracle.admm_quadratic(Q, q, oproxg, x0=x0, iters=400, tol=0.0, mu=0.05)
            zo = oracle.fourier2complex(ro["z"], 1 if zero else None)
            assert rel(x[0, i], zo) <= 1e-8, (name, i, rel(x[0, i], zo))
            assert np.array_equal(x[0, i] != 0, zo != 0), (name, i)
            if "ball" in name:
                assert np.count_nonzero(ro["z"]) == 6
        # the drivers take the engine for these keywords now, and agree with the reference's sequential loop on the device
        S, _ = L.ls_windowpsd(y, t, f, nw=k, noverlap=0, window_func=L.hanning, estimator=L.ls_sparse_spectral, **kw)
        assert api.windowpsd_last_timing()["windows"] == k
        Sseq, _ = L.ls_windowpsd(y, t, f, nw=k, noverlap=0, window_func=L.hanning, estimator=L.ls_sparse_spectral, batched=False, printerval=100000, **kw)
        assert rel(S, Sseq) <= 1e-8, (name, rel(S, Sseq))


@pytest.mark.parametrize("knobs", [{"LPVS_WINDOW_CHUNK_MB": "3", "LPVS_WINDOWS_IN_FLIGHT": "2"}, {"LPVS_WINDOW_CHUNK_MB": "0", "LPVS_WINDOWS_IN_FLIGHT": "3"},
                                   {"LPVS_WINDOW_CHUNK_MB": "5", "LPVS_WINDOWS_IN_FLIGHT": "1"}, {}])
def test_chunked_engine_is_bit_identical_to_the_uncut_one(L, knobs, monkeypatch):
    """The engine in cache-sized chunks with several parts of a chunk in flight (windows_engine_chunked): coefficients, iteration counts
    and every accumulation over windows (S of ls_windowpsd, the csd sums) equal the uncut call bit for bit -- here with chunks of a few
    windows, ragged last chunks, two or three parts, the sparse estimator on one and on two signals."""
    rng = np.random.default_rng(12)
    n, nwin, Nf = 1 << 10, 83, 96                              # (0.092 MB of packed inverse per window: chunks of 32 / 54 windows)
    t = np.arange(nwin * n, dtype=np.float64)
    f = np.arange(1, Nf + 1) / 250.0
    y = np.sin(2 * np.pi * f[20] * t) * (1 + 0.3 * np.sin(2 * np.pi * t / (7 * n))) + 0.3 * rng.standard_normal(nwin * n)
    u = 0.7 * np.sin(2 * np.pi * f[20] * t + 0.5) + 0.4 * np.cos(2 * np.pi * f[55] * t) + 0.3 * rng.standard_normal(nwin * n)
    kw = dict(λ=0.3, μ=1e-3, tol=0.0, iters=150)
    def run():
        x, S, its = L.windowpsd_sparse_batched(y, t, f, n, 0, None, **kw)
        eng = dict(estimator=L._lib.EST_SPARSE, lam=0.0, prox=L.NormL1(0.3).device_params(2 * Nf), μ=1e-3, tol=0.0, iters=150, sign=1)
        acc = L.windowcsd_batched(y, u, t, f, n, 0, None, eng)
        return x, S, its, acc
    monkeypatch.setenv("LPVS_WINDOW_CHUNK_MB", "0"); monkeypatch.setenv("LPVS_WINDOWS_IN_FLIGHT", "1")
    x0, S0, its0, acc0 = run()
    monkeypatch.delenv("LPVS_WINDOW_CHUNK_MB"); monkeypatch.delenv("LPVS_WINDOWS_IN_FLIGHT")
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    x1, S1, its1, acc1 = run()
    assert np.array_equal(x1, x0) and np.array_equal(S1, S0) and np.array_equal(its1, its0)
    for a, b in zip(acc1, acc0):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_window_state_entry_point(L, oracle, monkeypatch):
    """lpvs_windows_estimate_state_f64 (round 6): the raw ADMM state x, z, u of every problem of a window batch -- two signals sharing every window's
    factorisation, a window SUB-range, ragged chunks with two parts in flight -- is (a) the same run as lpvs_windows_estimate_f64 (fourier2complex(z)
    equals its coefficients bit for bit, same iteration counts), (b) the single-window handle's state of the same problem to summation order, with
    u += x - z holding between them, and (c) refused for the dense estimator and for device output pointers."""
    import ctypes as C
    from lpvspectral_jl_amd import _lib, api
    rng = np.random.default_rng(21)
    n, nwin, Nf = 1 << 10, 41, 96
    t = np.arange(nwin * n, dtype=np.float64)
    f = np.arange(0, Nf) / 250.0                                   # zero frequency first: Nreg = 2 Nf - 1
    y = np.sin(2 * np.pi * f[20] * t) + 0.3 * rng.standard_normal(nwin * n) + 0.2
    v = 0.7 * np.cos(2 * np.pi * f[55] * t) + 0.3 * rng.standard_normal(nwin * n)
    eng = dict(estimator=_lib.EST_SPARSE, lam=0.0, prox=(_lib.PROX_L1, 0.3, 0), μ=1e-3, tol=0.0, iters=150, sign=_lib.LINEAR_QUADRATIC_AS_WRITTEN)
    monkeypatch.setenv("LPVS_WINDOW_CHUNK_MB", "3"); monkeypatch.setenv("LPVS_WINDOWS_IN_FLIGHT", "2")
    lo, hi = 5, 38
    xc, its = api.windows_estimate([y, v], t, f, n, 0, None, eng, win_lo=lo, win_hi=hi)
    x, z, u, its_s = api.windows_estimate_state([y, v], t, f, n, 0, None, eng, win_lo=lo, win_hi=hi)
    assert x.shape == z.shape == u.shape == (2, hi - lo, 2 * Nf - 1) and np.array_equal(its, its_s) and np.all(its == 150)
    for q in range(2):
        for i in range(hi - lo):
            assert np.array_equal(oracle.fourier2complex(z[q, i], 1), xc[q, i]), (q, i)
    for q, sig in enumerate((y, v)):
        for i in (0, 16, hi - lo - 1):                               # first window of the range, a chunk boundary's neighbourhood, the last
            w = lo + i
            with L.Problem.fourier(sig[w * n:(w + 1) * n], t[w * n:(w + 1) * n], f, np.ones(n)) as p:
                p.set_prox(L.NormL1(0.3))
                p.admm_init(None, μ=1e-3, tol=0.0, linear_sign=-1)
                p.admm_run(150)
                xs, zs, us = p.admm_get()
            assert rel(x[q, i], xs) <= 1e-9 and rel(z[q, i], zs) <= 1e-9 and rel(u[q, i], us) <= 1e-9, (q, i)
            assert np.array_equal(z[q, i] != 0, zs != 0) and np.count_nonzero(u[q, i]) > 0
    # (c) misuse
    with pytest.raises(ValueError, match="sparse estimators"):
        api.windows_estimate_state([y], t, f, n, 0, None, dict(eng, estimator=_lib.EST_DENSE))
    import torch
    dz = torch.zeros(nwin * (2 * Nf - 1), dtype=torch.float64, device="cuda")
    yk, ty, fy = (np.ascontiguousarray(a) for a in (y, t, f))
    rc = _lib.lib().lpvs_windows_estimate_state_f64(_lib.out_ptr(yk), 1, _lib.out_ptr(ty), len(yk), n, 0, None, _lib.out_ptr(fy), Nf, _lib.EST_SPARSE, 0.0,
                                                    _lib.PROX_L1, 0.3, 0, 1e-3, 0.0, 10, -1, 0, nwin, 0, None, C.c_void_p(dz.data_ptr()), None, None)
    assert rc != 0 and b"HOST arrays" in _lib.lib().lpvs_last_error()


@pytest.mark.parametrize("grid,window,mu,prox", [("irregular", "rect", 0.05, "l1"), ("irregular", "hanning", 1e-4, "l0"), ("irregular", "hanning", 1.0, "group"),
                                                 ("equidistant", "hanning", 0.05, "group"), ("equidistant", "rect", 1.0, "l1")])
def test_window_batches_off_family_against_oracle(L, oracle, grid, window, mu, prox):
    """The window batches' default numerics (mixed storage, one launch per iteration, 32-bit reads + stale nibble product) OFF cfg4's input family
    (BASELINE's equidistant record with mu = 1e-4 has nearly diagonal inverses: tools/cfg4_tile_magnitudes.py), held to the ORACLE -- not to another HIP
    path -- through the raw state: irregular sampling (dense inverses), a window function, mu from 1e-4 to 1, all three fusable prox operators, 600
    iterations (past launch 256, where the refresh period has ramped to 32); x, z AND u of every window against oracle.admm_gram on the window's
    device Gram (Quadratic(Q, +q) as written: b = -q), rel-L2 <= 1e-9, identical support.  Which iteration ran is asserted."""
    from lpvspectral_jl_amd import _lib, api
    rng = np.random.default_rng(7)
    n, nwin, Nf = 3000, 6, 256                                      # Nreg = 512: four row blocks, ten tiles per window -- cfg4's tile shape
    t = np.cumsum(0.5 + rng.random(n * nwin)) if grid == "irregular" else np.arange(n * nwin, dtype=np.float64)
    f = (np.arange(Nf) + 1.0) / (2.6 * Nf)
    y = np.sin(2 * np.pi * f[11] * t) + 0.6 * np.cos(2 * np.pi * f[90] * t + 0.3) + 0.3 * rng.standard_normal(n * nwin)
    W = L.hanning(n) if window == "hanning" else None
    Wh = np.asarray(W) if W is not None else np.ones(n)
    with L.Problem.fourier(y[:n], t[:n], f, Wh) as p0:
        Q0, q0 = p0.get_gram()
    # a penalty that leaves a non-trivial support: from the first window's own correlations
    lam = {"l1": float(np.quantile(np.abs(q0), 0.8)), "group": float(np.quantile(np.abs(q0), 0.8)) * 1.5,
           "l0": float(np.quantile(np.abs(np.linalg.solve(Q0 + np.eye(len(q0)) / mu, q0)), 0.8)) ** 2 / (2 * mu)}[prox]
    kind, glen, oprox = {"l1": (_lib.PROX_L1, 0, oracle.NormL1(lam)), "l0": (_lib.PROX_L0, 0, oracle.NormL0(lam)),
                         "group": (_lib.PROX_GROUP_L2, 4, oracle.GroupL2(lam, 4))}[prox]
    eng = dict(estimator=_lib.EST_SPARSE, lam=0.0, prox=(kind, lam, glen), μ=mu, tol=0.0, iters=600, sign=_lib.LINEAR_QUADRATIC_AS_WRITTEN)
    x, z, u, its = api.windows_estimate_state([y], t, f, n, 0, W, eng)
    tm = api.windowpsd_last_timing()
    assert np.all(its == 600) and x.shape == (1, nwin, 2 * Nf)
    worst = 0.0
    for i in range(nwin):
        with L.Problem.fourier(y[i * n:(i + 1) * n], t[i * n:(i + 1) * n], f, Wh) as p:
            Q, q = p.get_gram()
        ro = oracle.admm_gram(Q, -q, oprox, iters=600, tol=0.0, mu=mu)
        e = {k: rel(v[0, i], ro[k]) for k, v in (("x", x), ("z", z), ("u", u))}
        worst = max(worst, *e.values())
        assert max(e.values()) <= 1e-9, (i, e, tm)
        assert np.array_equal(z[0, i] != 0, ro["z"] != 0) and 0 < np.count_nonzero(ro["z"]) < ro["z"].size, i
    print(f"window batch, {grid} grid, {window} window, mu = {mu:g}, {prox}: x, z, u of {nwin} windows vs oracle.admm_gram worst rel-L2 {worst:.2e}; "
          f"one launch per iteration {tm['one_launch_iteration']}, 32-bit reads {tm['reads_32_bits']}")
    if grid == "equidistant":
        assert tm["one_launch_iteration"] and tm["reads_32_bits"], tm        # (nearly diagonal inverses: every tile fixed point)
