"""lpvs_problem_set_option / lpvs_set_default_option (include/lpvspectral.h LPVS_OPT_*): the storage of the packed inverse, the
iteration scheme and the Gram form as API options instead of environment variables -- explicit options win over the environment,
handles carry their own, the batched-window entry points take the calling thread's defaults.  GPU only."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def _signal(N, Nf, rng):
    X = np.sort(rng.random(N) * (10.0 * N / 500)); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25.0 / Nf
    y = 2 * V ** 2 * np.cos(w[Nf // 10] * X) + 2 / (5 * V + 1) * np.cos(w[Nf // 3] * X - 0.3) + 0.1 * rng.standard_normal(N)
    return y, X, V, w


@pytest.fixture(scope="module")
def sig():
    return _signal(1 << 18, 128, np.random.default_rng(3))     # n = 2048: the smallest size with a packed inverse


def test_handle_options_choose_storage_and_iteration(L, sig, monkeypatch):
    y, X, V, w = sig
    monkeypatch.delenv("LPVS_M_STORAGE", raising=False); monkeypatch.delenv("LPVS_ITERATION", raising=False)
    prox = L.SlicedSeparableSum.frequency_groups(2.0, 128, 16)
    out = {}
    with L.Problem.lpv(y, X, V, w, 8) as p:
        p.set_prox(prox)
        for storage, iteration, kernel in ((None, None, "admm_iter_mixed_kernel"), (None, "two", "symv_tile_mixed_kernel"),
                                           ("split", None, "symv_tile_split_kernel"), ("f64", None, "symv_tile_kernel<double>"),
                                           ("mixed", "one", "admm_iter_mixed_kernel")):
            p.set_option("storage", storage); p.set_option("iteration", iteration)
            assert p.get_option("storage") == storage and p.get_option("iteration") == iteration
            p.admm_init(None, μ=0.05, tol=0.0)
            assert p.matvec_info()["kernel"] == kernel, (storage, iteration, p.matvec_info())
            p.admm_run(200)
            out[(storage, iteration)] = p.admm_get()[1]
        # an explicit option wins over the environment; without one the environment still works (experiments, old scripts)
        monkeypatch.setenv("LPVS_M_STORAGE", "f64")
        p.set_option("storage", "split"); p.set_option("iteration", None)
        p.admm_init(None, μ=0.05, tol=0.0)
        assert p.matvec_info()["kernel"] == "symv_tile_split_kernel"
        p.set_option("storage", None)
        assert p.get_option("storage") == "f64"
        p.admm_init(None, μ=0.05, tol=0.0)
        assert p.matvec_info()["kernel"] == "symv_tile_kernel<double>"
        with pytest.raises(RuntimeError):                                   # constructor-time choices are not handle options
            p.set_option("gram_form", "krs")
        with pytest.raises(ValueError):
            p.set_option("storage", "bf16")
    ref = out[("f64", None)]
    for k, z in out.items():
        assert rel(z, ref) <= 1e-9, (k, rel(z, ref))
        assert np.array_equal(z != 0, ref != 0)


def test_storage_switches_on_one_handle_repack_and_rebind(L, sig, monkeypatch):
    """split -> mixed and split -> default on ONE handle with the same mu (so the factorisation is cached and
    only the packed copy changes).  The stale (Mp_mode, Mp_valid) of the split packing used to be taken for a demotion that had just
    happened: the buffer was repacked in the mixed layout and then decoded by the split kernel."""
    y, X, V, w = sig
    monkeypatch.delenv("LPVS_M_STORAGE", raising=False); monkeypatch.delenv("LPVS_ITERATION", raising=False)
    prox = L.SlicedSeparableSum.frequency_groups(2.0, 128, 16)
    seq = (("f64", "symv_tile_kernel<double>"), ("split", "symv_tile_split_kernel"), ("mixed", "admm_iter_mixed_kernel"),
           ("split", "symv_tile_split_kernel"), (None, "admm_iter_mixed_kernel"), ("split", "symv_tile_split_kernel"), ("mixed", "admm_iter_mixed_kernel"),
           ("f64", "symv_tile_kernel<double>"), (None, "admm_iter_mixed_kernel"),
           ("mixed32", "admm_iter_mixed_kernel"), ("mixed", "admm_iter_mixed_kernel"), ("mixed32", "admm_iter_mixed_kernel"))   # (round 5: only the bits READ change -- the packed copy stays)
    zs = []
    with L.Problem.lpv(y, X, V, w, 8) as p:
        p.set_prox(prox)
        for storage, kernel in seq:
            p.set_option("storage", storage)
            p.admm_init(None, μ=0.05, tol=0.0)
            assert p.matvec_info()["kernel"] == kernel, (storage, p.matvec_info())
            # (32-bit reads + the stale nibble product: the default of this corrected handle, and "mixed32" by name; "mixed" = all 36 bits)
            assert ("32-bit fixed point reads" in p.matvec_info()["storage"]) == (storage in (None, "mixed32")), (storage, p.matvec_info())
            p.admm_run(200)
            zs.append((storage, p.admm_get()[1]))
    ref = zs[0][1]
    for storage, z in zs[1:]:
        assert rel(z, ref) <= 1e-9, (storage, rel(z, ref))
        assert np.array_equal(z != 0, ref != 0), storage


def test_default_options_reach_constructors_estimators_and_the_window_engine(L, sig, monkeypatch):
    from lpvspectral_jl_amd import api
    y, X, V, w = sig
    for v in ("LPVS_M_STORAGE", "LPVS_ITERATION", "LPVS_GRAM_FORM", "LPVS_NUDFT"):
        monkeypatch.delenv(v, raising=False)
    N = 20000
    with L.default_options(gram_form="krs"):
        assert L.get_default_option("gram_form") == "krs"
        with L.Problem.lpv(y[:N], X[:N], V[:N], w[:16], 4) as p:
            assert p.timing()["gram_form"] == "krs"
    assert L.get_default_option("gram_form") is None
    with L.Problem.lpv(y[:N], X[:N], V[:N], w[:16], 4) as p:
        assert p.timing()["gram_form"] in ("ap", "ap-nufft")
    with L.default_options(slot_sums="direct"):
        with L.Problem.lpv(y[:N], X[:N], V[:N], w[:16], 4) as p:
            assert p.timing()["gram_form"] == "ap"
    # estimator keywords (extensions): same result to the storage's 1e-10
    kw = dict(λ=2.0, iters=150, tol=0.0, printerval=1000, out=io.StringIO())
    a = L.ls_sparse_spectral_lpv(y, X, V, w, 8, **kw)
    b = L.ls_sparse_spectral_lpv(y, X, V, w, 8, storage="f64", iteration="two", **kw)
    assert L.get_default_option("storage") is None                        # restored
    assert rel(a.x, b.x) <= 1e-9 and np.array_equal(a.x != 0, b.x != 0)
    # the batched-window engine has no handle: it takes the calling thread's defaults
    n, nwin, Nf = 1 << 14, 4, 256
    t = np.arange(n * nwin, dtype=np.float64); f = np.arange(Nf) / 512.0
    yw = np.sin(2 * np.pi * f[33] * t) + 0.3 * np.random.default_rng(4).standard_normal(n * nwin)
    kww = dict(nw=nwin, noverlap=0, estimator=L.ls_sparse_spectral, λ=0.2, μ=1e-4, iters=100, tol=0.0)
    S1, _ = L.ls_windowpsd(yw, t, f, **kww)
    assert api.windowpsd_last_timing()["one_launch_iteration"]
    S2, _ = L.ls_windowpsd(yw, t, f, iteration="two", **kww)
    assert not api.windowpsd_last_timing()["one_launch_iteration"]
    S3, _ = L.ls_windowpsd(yw, t, f, storage="f64", ngpus=0, **kww)      # the multi-device driver: options travel to its worker threads
    assert rel(S2, S1) <= 1e-9 and rel(S3, S1) <= 1e-9


def test_window_plan_options_replace_the_environment_knobs(L, monkeypatch):
    """LPVS_OPT_WINDOW_CHUNK_MB / LPVS_OPT_WINDOWS_IN_FLIGHT / LPVS_OPT_RESERVE_CUS as default options (the environment variables of
    the same names stay as the fallback): integer values, the uncut plan by name, an explicit option wins over the environment, and the
    engine's results do not depend on the plan, bit for bit."""
    for v in ("LPVS_WINDOW_CHUNK_MB", "LPVS_WINDOWS_IN_FLIGHT", "LPVS_RESERVE_CUS"):
        monkeypatch.delenv(v, raising=False)
    rng = np.random.default_rng(12)
    n, nwin, Nf = 1 << 10, 83, 96
    t = np.arange(nwin * n, dtype=np.float64)
    f = np.arange(1, Nf + 1) / 250.0
    y = np.sin(2 * np.pi * f[20] * t) + 0.3 * rng.standard_normal(nwin * n)
    kw = dict(λ=0.3, μ=1e-3, tol=0.0, iters=150)
    with L.default_options(window_chunk_mb="uncut", windows_in_flight=1):
        assert L.get_default_option("window_chunk_mb") == "uncut" and L.get_default_option("windows_in_flight") == 1
        x0, S0, its0 = L.windowpsd_sparse_batched(y, t, f, n, 0, None, **kw)
    assert L.get_default_option("window_chunk_mb") is None
    for opts in (dict(window_chunk_mb=3, windows_in_flight=2), dict(window_chunk_mb=5, windows_in_flight=3), dict()):
        with L.default_options(**opts):
            x1, S1, its1 = L.windowpsd_sparse_batched(y, t, f, n, 0, None, **kw)
        assert np.array_equal(x1, x0) and np.array_equal(S1, S0) and np.array_equal(its1, its0), opts
    monkeypatch.setenv("LPVS_WINDOW_CHUNK_MB", "0"); monkeypatch.setenv("LPVS_WINDOWS_IN_FLIGHT", "1")     # the fallback still works ...
    x2, S2, _ = L.windowpsd_sparse_batched(y, t, f, n, 0, None, **kw)
    with L.default_options(window_chunk_mb=3, windows_in_flight=2):                                         # ... and loses against an option
        x3, S3, _ = L.windowpsd_sparse_batched(y, t, f, n, 0, None, **kw)
    assert np.array_equal(x2, x0) and np.array_equal(x3, x0)
    with pytest.raises(ValueError):
        L.set_default_option("windows_in_flight", 9)
    # the factorisation's CU reservation: a device-level knob, same inverse either way
    N = 1 << 16
    sy, sX, sV, sw = _signal(N, 128, np.random.default_rng(5))
    Ms = []
    for r in ("none", 16, None):
        with L.default_options(reserve_cus=r):
            with L.Problem.lpv(sy, sX, sV, sw, 8) as p:
                with pytest.raises(RuntimeError):
                    p.set_option("reserve_cus", 4)                         # not a handle option
                Ms.append(p.get_inverse(20.0))
    assert rel(Ms[1], Ms[0]) <= 1e-12 and rel(Ms[2], Ms[0]) <= 1e-12
