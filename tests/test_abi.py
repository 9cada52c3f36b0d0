"""CPU checks of the drop-in boundary: the shared library loads, exports every symbol that
include/lpvspectral.h declares (and nothing the header lacks), the host-only entry points work,
and compute entry points fail loudly (never silently fall back) when no GPU is visible."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lpvspectral.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lpvs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(L):
    from lpvspectral_jl_amd import _lib
    names = _declared()
    assert len(names) >= 25
    handle = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/lpvspectral.h but not exported"
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r"\bT (lpvs_[a-z0-9_]+)", exported)))
    assert exported == names, "library exports and header declarations differ"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table out of date"


def test_version_and_error_string(L):
    from lpvspectral_jl_amd._lib import lib
    assert lib().lpvs_version() == 100
    assert isinstance(lib().lpvs_last_error(), bytes)


def test_window_bookkeeping_is_bit_exact(L):
    y = np.arange(1, 101)
    W = L.Windows2(y, y, 10, 0)
    assert len(W) == 10 and np.array_equal(next(iter(W))[0], np.arange(1, 11))
    assert np.array_equal(L.mapwindows(lambda yt: -yt[0], W), -np.arange(1, 101))
    W = L.Windows2(y, y, 10, 1)
    assert len(W) == 11 and np.array_equal(W.collect()[1][0], np.arange(10, 20))
    assert np.array_equal(L.mapwindows(lambda yt: -yt[0], W), -np.arange(1, 101))
    W3 = L.Windows3(y, y, y, 10, 1)
    assert len(W3) == 11 and all(np.array_equal(p, np.arange(10, 20)) for p in W3.collect()[1])
    assert len(L.Windows2(y, y)) == (100 - 12) // (12 - 6) + 1          # defaults n=L>>3, noverlap=n>>1
    assert len(L.Windows2(np.arange(5), np.arange(5), 10, 0)) == 0      # L < n -> no windows
    with pytest.raises(L.DomainError):
        L.Windows2(y, y, 10, 10)
    with pytest.raises(AssertionError):
        L.Windows2(y, y[:-1], 10, 0)


def test_window_offsets_match_oracle(L, oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        Lh = int(rng.integers(1, 5000)); n = int(rng.integers(1, 300)); nov = int(rng.integers(-1, n))
        a = L.Windows2(np.zeros(Lh), np.zeros(Lh), n, nov)
        b = oracle.Windows2(np.zeros(Lh), np.zeros(Lh), n, nov)
        assert np.array_equal(a.offsets, b.offsets)


def test_merge_matches_oracle(L, oracle):
    rng = np.random.default_rng(1)
    y = rng.standard_normal(257)
    for n, nov in [(32, 0), (32, 16), (50, 49), (40, 7)]:
        a = L.Windows2(y, y, n, nov); b = oracle.Windows2(y, y, n, nov)
        yf = rng.standard_normal((len(a), n))
        assert np.array_equal(L.merge(yf, a), oracle.merge(yf, b))


def test_check_freq_and_default_freqs(L):
    t = np.arange(1000) * 0.1
    f = L.default_freqs(t)
    assert f[0] == 0 and f[-1] == 5 and len(f) == 501
    assert L.check_freq(f) == 1 and L.check_freq(f[1:]) is None
    with pytest.raises(ValueError):
        L.check_freq([1, 0, 2])


def test_prox_mirrors(L):
    g = L.SlicedSeparableSum([L.NormL2(5.0)] * 3, [(range(1, 5),), (range(5, 9),), (range(9, 13),)])
    assert g.device_params(12) == (4, 5.0, 4)
    with pytest.raises(NotImplementedError):
        L.SlicedSeparableSum([L.NormL2(5.0)] * 2, [(range(1, 5),), (range(6, 10),)])
    assert L.NormL1(0.01).device_params(8) == (1, 0.01, 0)
    assert L.IndBallL0(32).device_params(8) == (3, 32.0, 0)


def test_no_silent_cpu_fallback(L):
    from lpvspectral_jl_amd._lib import lib
    if lib().lpvs_device_count() > 0:
        pytest.skip("GPU present")
    t = np.arange(64.0)
    with pytest.raises(L.DeviceError):
        L.get_fourier_regressor(t, np.arange(1, 5) / 10)
    with pytest.raises(L.DeviceError):
        L.ls_sparse_spectral(np.sin(t), t, np.arange(1, 5) / 10, iters=3)
    with pytest.raises(L.DeviceError):
        L.ls_sparse_spectral_lpv(np.sin(t), t, t / 64, np.arange(1, 5.0), 2, iters=3)
    with pytest.raises(L.DeviceError):
        L.ADMM(np.zeros(3), L.Quadratic(np.eye(3), np.ones(3)), L.NormL1(1.0), iters=2)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "lpvspectral.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for fn in fs:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle" not in txt.lower().replace("(the oracle", ""), f"{fn} mentions the oracle"


def test_default_options_roundtrip_without_a_gpu(L):
    """lpvs_set_default_option / lpvs_get_default_option are host-only bookkeeping (thread-local)."""
    import threading
    assert L.get_default_option("storage") is None
    with L.default_options(storage="f64", iteration="two", nt_loads="on"):
        assert (L.get_default_option("storage"), L.get_default_option("iteration"), L.get_default_option("nt_loads")) == ("f64", "two", "on")
        seen = []
        th = threading.Thread(target=lambda: seen.append(L.get_default_option("storage")))
        th.start(); th.join()
        assert seen == [None]                                               # another thread has its own defaults
    assert L.get_default_option("storage") is None and L.get_default_option("iteration") is None
    with pytest.raises(ValueError):
        L.set_default_option("storage", "bf16")
    with pytest.raises(KeyError):
        L.set_default_option("precision", "f64")
    from lpvspectral_jl_amd._lib import lib
    assert lib().lpvs_set_default_option(99, 1) == -1 and lib().lpvs_set_default_option(1, 7) == -1
