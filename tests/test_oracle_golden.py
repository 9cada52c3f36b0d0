"""Pins the CPU oracle against the reference's own known answers (test/runtests.jl) and
cross-checks the C restatement against independent numpy statements.  CPU only."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def _t():
    g = GOLD["default_freqs"]["t"]
    return np.arange(1000) * g["step"] + g["start"]


@pytest.mark.parametrize("case", GOLD["windows2"])
def test_windows2(oracle, case):
    y = np.arange(1, case["L"] + 1)
    W = oracle.Windows2(y, y, case["n"], case["noverlap"])
    assert len(W) == case["count"]
    ws = list(W)
    assert np.array_equal(ws[0][0], np.arange(case["first"][0], case["first"][1] + 1))
    if "second" in case:
        assert np.array_equal(ws[1][0], np.arange(case["second"][0], case["second"][1] + 1))
        assert np.array_equal(ws[1][1], ws[1][0])
    res = oracle.mapwindows(lambda yt: -yt[0], W)
    assert np.array_equal(res, -np.arange(1, case["L"] + 1))


def test_default_freqs_and_check_freq(oracle):
    g = GOLD["default_freqs"]
    f = oracle.default_freqs(_t())
    assert f[0] == g["first"] and f[-1] == g["last"] and len(f) == g["length"]
    assert oracle.check_freq(f) == GOLD["check_freq"]["ok_returns"]
    with pytest.raises(ValueError):
        oracle.check_freq(GOLD["check_freq"]["throws_ArgumentError_for"])


def test_regressor_size_and_numpy_crosscheck(oracle):
    t = _t()
    f = oracle.default_freqs(t)
    A, z = oracle.get_fourier_regressor(t, f)
    assert list(A.shape) == GOLD["regressor_size"]["size"] and z == 1
    A2, _ = oracle.get_fourier_regressor_np(t, f)
    assert np.abs(A - A2).max() < 1e-15
    f2 = f[1:]
    A3, z3 = oracle.get_fourier_regressor(t, f2)
    A4, _ = oracle.get_fourier_regressor_np(t, f2)
    assert z3 is None and A3.shape == (1000, 1000) and np.abs(A3 - A4).max() < 1e-15


def test_ls_spectral_known_answer(oracle):
    g = GOLD["ls_spectral_sine"]
    t = _t()
    y = np.sin(2 * np.pi * t)
    x, f = oracle.ls_spectral(y, t)
    p = np.abs(x) ** 2
    assert abs(p.max() - g["findmax_abs2"][0]) < g["atol"] and p.argmax() + 1 == g["findmax_abs2"][1]
    x, _ = oracle.ls_spectral(y, t, f, np.ones(len(y)))
    p = np.abs(x) ** 2
    assert abs(p.max() - g["findmax_abs2"][0]) < g["atol"] and p.argmax() + 1 == g["findmax_abs2"][1]


def test_tls_spectral_known_answer(oracle):
    g = GOLD["tls_spectral_sine"]
    t = _t()
    y = np.sin(2 * np.pi * t)
    x, f = oracle.tls_spectral(y, t)
    p = np.abs(x) ** 2
    assert len(f) == 500
    assert abs(p.max() - g["findmax_abs2"][0]) < g["atol"] and p.argmax() + 1 == g["findmax_abs2"][1]


@pytest.mark.parametrize("case", GOLD["ls_windowpsd"])
def test_ls_windowpsd_known_answer(oracle, case):
    t = _t()
    y = np.sin(2 * np.pi * t)
    S, _ = oracle.ls_windowpsd(y, t, nw=case["nw"], noverlap=case["noverlap"])
    assert np.abs(S).argmax() + 1 == case["argmax"]


def test_ls_windowcsd_and_cohere_known_answers(oracle):
    """test/runtests.jl:203-213: csd peak (2 length(freqs), 11); cohere(y, y) == 1 everywhere; with noise the mean coherence
    stays below 0.25 and the maximum sits next to the signal's frequency (the reference's own draw is Julia's RNG)."""
    g = GOLD["ls_windowcsd"]
    t = _t()
    y = np.sin(2 * np.pi * t)
    x, fr = oracle.ls_windowcsd(y, y, t, nw=g["nw"], noverlap=g["noverlap"])
    a = np.abs(x)
    assert abs(a.max() - 2.0 * len(fr)) < g["atol"] and a.argmax() + 1 == g["findmax_abs"][1]
    c, _ = oracle.ls_cohere(y, y, t)
    assert np.all(c == GOLD["ls_cohere_self"]["all_equal"])
    rng = np.random.default_rng(0)
    c, _ = oracle.ls_cohere(y, y + 0.5 * rng.standard_normal(len(y)), t, nw=8, noverlap=-1)
    assert abs(c.max() - 1.0) < 0.15 and abs(int(c.argmax()) + 1 - 14) <= 1 and c.mean() < 0.25
    W3 = oracle.Windows3(np.arange(1, 101), np.arange(1, 101), np.arange(1, 101), 10, 1)          # test/runtests.jl:48-62
    assert len(W3) == 11 and np.array_equal(list(W3)[1][2], np.arange(10, 20))


def _lpv_signal(N, seed):
    rng = np.random.default_rng(seed)
    X = np.sort(10 * rng.random(N))
    V = np.linspace(0, 1, N)
    fd = [lambda v: 2 * v ** 2, lambda v: 2 / (5 * v + 1), lambda v: 3 * np.exp(-10 * (v - 0.5) ** 2)]
    w = 2 * np.pi * np.array([2.0, 10.0, 20.0])
    dep = np.stack([fd[i](V) for i in range(3)], 1)
    Y = (dep * np.cos(w[None, :] * X[:, None] - 0.5 * dep)).sum(1) + 0.1 * rng.standard_normal(N)
    return Y, X, V


def test_lpv_regressor_matches_literal_transcription(oracle):
    Y, X, V = _lpv_signal(60, 3)
    w = 2 * np.pi * np.arange(2, 12, 2.0)
    Phi = oracle.lpv_regressor(X, V, w, 4)
    Phi2, inds = oracle.lpv_regressor_np(X, V, w, 4)
    assert np.abs(Phi - Phi2).max() < 1e-15
    # inds layout of src/lasso.jl:47 (verified table: Nf=3, Nv=2 -> [1,4,7,10, 2,5,8,11, 3,6,9,12])
    i32 = np.arange(12).reshape((3, 4), order="F").T.ravel(order="F") + 1
    assert list(i32) == [1, 4, 7, 10, 2, 5, 8, 11, 3, 6, 9, 12]


def test_lpv_top3_known_answer(oracle):
    g = GOLD["lpv_top3"]
    Y, X, V = _lpv_signal(g["N"], 0)
    w_test = 2 * np.pi * np.array(g["w_test_hz"], dtype=float)
    x = oracle.ls_spectral_lpv(Y, X, V, w_test, g["Nv"], lam=g["lambda"])
    top = set(np.argsort(-oracle.psd(x, len(w_test)))[:3] + 1)
    assert top == set(g["top3_psd_indices"])
    p, r = oracle.ls_sparse_spectral_lpv(Y, X, V, w_test, g["Nv"], lam=5, tol=1e-8, iters=2000)
    assert set(np.argsort(-oracle.psd(p, len(w_test)))[:3] + 1) == set(g["top3_psd_indices"])


def test_faithful_cg_form_equals_gram_form(oracle):
    """The reference's x-update (warm-started CG, reltol sqrt(eps)) and the exact Gram-form solve walk
    the same trajectory: equal iteration counts, z equal to ~1e-9 (tall and fat cases)."""
    Y, X, V = _lpv_signal(300, 1)
    w_test = 2 * np.pi * np.arange(2, 26, 2.0)
    for Nv in (4, 20):  # tall (96 < 300) and fat (480 > 300)
        Phi = oracle.lpv_regressor(X, V, w_test, Nv)
        r1 = oracle.admm_ls(Phi, Y, oracle.GroupL2(5, 2 * Nv), iters=400, tol=1e-9, history=True)
        G, b = oracle.gram(Phi, Y)
        r2 = oracle.admm_gram(G, b, oracle.GroupL2(5, 2 * Nv), iters=400, tol=1e-9, history=True)
        assert r1["iters"] == r2["iters"]
        assert np.linalg.norm(r1["z"] - r2["z"]) <= 1e-8 * max(np.linalg.norm(r2["z"]), 1e-30)
        assert np.array_equal(r1["z"] != 0, r2["z"] != 0)


def test_quadratic_as_written_sign(oracle):
    """src/lasso.jl:119-121 passes q=+A'Wy to Quadratic: coefficients come out negated relative to the
    least-squares path; abs2 (the PSD) is unaffected."""
    rng = np.random.default_rng(5)
    t = np.sort(rng.random(200)) * 20
    f = np.arange(1, 21) / 10.0
    y = np.sin(2 * np.pi * 0.7 * t) + 0.05 * rng.standard_normal(200)
    W = np.ones(200)
    xu, _, _ = oracle.ls_sparse_spectral(y, t, f, lam=0.05, iters=500, tol=0, mu=0.05)
    xw, _, _ = oracle.ls_sparse_spectral(y, t, f, W, lam=0.05, iters=500, tol=0, mu=0.05)
    assert np.linalg.norm(xu + xw) < 1e-6 * np.linalg.norm(xu)


def test_prox_operators(oracle):
    v = np.array([3.0, -0.2, 0.5, -4.0, 0.0, 1.0])
    assert np.allclose(oracle.prox(oracle.NormL1(2.0), v, 0.25), [2.5, 0, 0, -3.5, 0, 0.5])
    assert np.allclose(oracle.prox(oracle.NormL0(2.0), v, 0.25), [3.0, 0, 0, -4.0, 0, 0])  # thr = 1: strict >
    assert np.allclose(oracle.prox(oracle.IndBallL0(2), v, 0.25), [3.0, 0, 0, -4.0, 0, 0])
    z = oracle.prox(oracle.GroupL2(2.0, 3), v, 0.25)
    n1, n2 = np.linalg.norm(v[:3]), np.linalg.norm(v[3:])
    assert np.allclose(z[:3], (1 - 0.5 / n1) * v[:3]) and np.allclose(z[3:], (1 - 0.5 / n2) * v[3:])
    assert np.all(oracle.prox(oracle.GroupL2(100.0, 3), v, 0.25) == 0)


def test_shared_factor_form_equals_the_single_runs(oracle):
    """oracle.admm_gram_multi (several right-hand sides, ONE Cholesky factor, snapshots: what a multi-channel device handle is held to at
    a long horizon) is admm_gram per column, bit for bit, for every prox operator and at every snapshot."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((300, 64)); G = A.T @ A; B = A.T @ rng.standard_normal((300, 3))
    for g in (oracle.IndBallL0(5), oracle.GroupL2(2.0, 8), oracle.NormL1(1.0), oracle.NormL0(0.5)):
        out = oracle.admm_gram_multi(G, B, g, [7, 40, 90], mu=0.05)
        for c in range(3):
            for cnt in (7, 40, 90):
                r = oracle.admm_gram(G, B[:, c], g, iters=cnt, tol=0.0, mu=0.05)
                assert all(np.array_equal(out[cnt][k][:, c], r[name]) for k, name in enumerate("xzu")), (g.kind, c, cnt)


def test_admm_mu_assert(oracle):
    with pytest.raises(AssertionError):
        oracle.admm_gram(np.eye(3), np.ones(3), oracle.NormL1(1.0), mu=1.5)


def test_oracle_under_address_and_ub_sanitizer():
    """SURVEY section 5 (race detection / sanitizers): the C restatement, built with -fsanitize=address,undefined
    (oracle/Makefile), runs a tour of its entry points -- regressors, prox operators, both ADMM forms, windows -- in a
    subprocess; any heap overflow / UB report fails the run.  (GPU sanitizers are not available on the pool.)"""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    odir = os.path.join(root, "oracle")
    subprocess.check_call(["make", "-s", "-C", odir, "liblpvs_oracle_asan.so"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        from oracle import oracle as o
        rng = np.random.default_rng(0)
        t = np.sort(rng.random(300)) * 300; f = np.arange(0, 17) / 40.0
        y = np.sin(2 * np.pi * f[5] * t) + 0.1 * rng.standard_normal(300)
        A, zf = o.get_fourier_regressor(t, f)
        G, b = o.gram(A, y, np.ones(300))
        for g in (o.NormL1(0.5), o.NormL0(0.5), o.IndBallL0(3), o.GroupL2(0.5, 3)):
            r1 = o.admm_gram(G, b, g, iters=40, tol=1e-9)
            r2 = o.admm_ls(A, y, g, iters=40, tol=1e-9)
            assert np.linalg.norm(r1["z"] - r2["z"]) <= 1e-6 * max(np.linalg.norm(r1["z"]), 1.0)
        o.admm_quadratic(G, b, o.NormL1(0.5), iters=20)
        X = np.sort(rng.random(200)) * 10; V = rng.random(200)
        Phi = o.lpv_regressor(X, V, 2 * np.pi * np.arange(1, 5.0), 3)
        Phi2 = o.lpv_regressor(X, V, 2 * np.pi * np.arange(1, 5.0), 3, True, True, False)
        o.ls_sparse_spectral_lpv(y[:200], X, V, 2 * np.pi * np.arange(1, 5.0), 3, lam=2.0, iters=30)
        W = o.Windows2(np.arange(1.0, 101), np.arange(1.0, 101), 10, 1)
        assert np.array_equal(o.mapwindows(lambda yt: -yt[0], W), -np.arange(1.0, 101))
        o.ls_cohere(y, y + 0.1, t, f, nw=3)
        print("asan tour ok")
    """ % root)
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               LPVS_ORACLE_SO=os.path.join(odir, "liblpvs_oracle_asan.so"), OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "asan tour ok" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]
