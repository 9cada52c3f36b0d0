"""Host mirror of the reference's estimator API on top of the HIP library.

Same names, argument order, keyword names (``λ``, ``μ``, ``proxg``, ``iters``, ``tol``,
``printerval``, ``cb``, ``init``, ``normalize``, ``coulomb``, ``nw``, ``noverlap``,
``window_func``, ``estimator``) and error behaviour as ``src/lsfft.jl`` / ``src/lasso.jl`` of
LPVSpectral.jl, so the parity tests read like ``test/runtests.jl``.  All arithmetic happens in
``liblpvspectral.so``; inputs may be numpy arrays (host) or float64 torch CUDA tensors (resident
in HBM).  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import logging
import sys
from dataclasses import dataclass
from typing import Any, Callable, Optional

import numpy as np

from . import _lib
from ._lib import as_f32, is_f32, as_f64, check, lib, out_ptr
from .prox import IndBallL0, LeastSquares, NormL0, NormL1, NormL2, Quadratic, SlicedSeparableSum
from .windows import Windows2, Windows3, rect

log = logging.getLogger("lpvspectral")


# --------------------------------------------------------------------------- result type
@dataclass
class SpectralExt:
    """src/LPVSpectral.jl:59-70."""
    Y: Any
    X: Any
    V: Any
    w: Any
    Nv: int
    λ: float
    coulomb: bool
    normalize: bool
    x: Any
    Σ: Any


def abs2(x):
    """Julia's ``abs2``: re² + im² (not hypot²), so that e.g. ``ls_cohere(y,y) == 1`` holds exactly."""
    x = np.asarray(x)
    return x.real * x.real + x.imag * x.imag if np.iscomplexobj(x) else x * x


def _mul_conj(a, b):
    """``a .* conj.(b)`` evaluated as Julia does (four real products, no fused multiply-add): numpy's own complex multiply may
    contract to FMAs on some CPUs, and then ``ls_cohere(y, y) == 1`` (test/runtests.jl:207-208) holds only approximately."""
    a, b = np.asarray(a), np.asarray(b)
    ar, ai, br, bi = a.real, a.imag, b.real, b.imag
    return (ar * br + ai * bi) + 1j * (ai * br - ar * bi)


def reshape_params(x, Nf):
    """src/utilities.jl:77: params as an [Nω × Nv] matrix."""
    return np.reshape(np.asarray(x), (int(Nf), -1), order="F")


def psd(se: SpectralExt):
    """src/lsfft.jl:214-217."""
    rp = reshape_params(np.array(se.x, copy=True), len(np.ravel(se.w)))
    return abs2(rp.sum(axis=1, keepdims=True))


# --------------------------------------------------------------------------- small host helpers
def default_freqs(t_or_n, fs=None, n=None):
    """src/lsfft.jl:3-9: ``default_freqs(n::Int, fs=1)``, ``default_freqs(t)``, ``default_freqs(t, n)``.
    rfftfreq(n, fs) = (0:n÷2)·fs/n (host side, no FFTW)."""
    if np.isscalar(t_or_n):
        nn = int(t_or_n)
        fs = 1.0 if fs is None else float(fs)
    else:
        t32 = is_f32(t_or_n)
        t = _host(t_or_n)
        if n is not None:
            t = t[: int(n)]
        nn = len(t)
        if fs is None:
            fs = 1.0 / np.mean(np.diff(t))
        if t32:   # Float32 time stamps give a Float32 grid (rfftfreq(n, fs::Float32)): multiplier fs / n rounded to Float32, k * multiplier
            m = np.float32(np.float32(fs) / np.float32(nn))
            return (np.arange(nn // 2 + 1) * np.float64(m)).astype(np.float32)
    return (np.arange(nn // 2 + 1) * float(fs)) / nn


def _all_f32(*arrays):
    """Julia's eltype promotion for the entry points that take several arrays: Float32 only when every array is."""
    return all(is_f32(a) for a in arrays)


def _host_vec(a):
    """A host vector in its own float eltype (float32 stays float32: eltype promotion is decided later), anything else float64."""
    if is_f32(a) and not _lib.is_device_array(a):
        return np.ravel(np.asarray(a.detach().numpy() if hasattr(a, "detach") else a, dtype=np.float32))
    return np.ravel(_host(a))


def _host(a):
    if _lib.is_device_array(a):
        return a.detach().cpu().numpy().astype(np.float64, copy=False)
    if hasattr(a, "detach") and hasattr(a, "numpy"):
        a = a.detach().numpy()
    return np.asarray(a, dtype=np.float64)


def check_freq(f):
    """src/lsfft.jl:20-24 -> ``None`` or 1; ValueError (ArgumentError) if zero is not first."""
    z = C.c_int64(0)
    if is_f32(f):
        kf, pf, Nf = as_f32(f)
        check(lib().lpvs_check_freq_f32(pf, Nf, C.byref(z)))
    else:
        fa = np.ascontiguousarray(_host(f))
        check(lib().lpvs_check_freq_f64(out_ptr(fa), len(fa), C.byref(z)))
    return None if z.value == 0 else int(z.value)


def get_fourier_regressor(t, f):
    """src/lsfft.jl:26-49 -> ``(A, zerofreq)`` with A an N×Nreg column-major numpy array."""
    f32 = _all_f32(t, f)                              # t and f share the eltype T of src/lsfft.jl:26 (mixed: promoted to Float64 here)
    conv = as_f32 if f32 else as_f64
    kt, pt, N = conv(t)
    kf, pf, Nf = conv(f)
    zf = check_freq(f)
    nreg = 2 * Nf - (1 if zf else 0)
    A = np.zeros((N, nreg), order="F", dtype=np.float32 if f32 else np.float64)
    z = C.c_int64(0)
    fn = lib().lpvs_fourier_regressor_f32 if f32 else lib().lpvs_fourier_regressor_f64
    check(fn(pt, N, pf, Nf, out_ptr(A), C.byref(z)))
    return A, zf


def basis_activation_func(V, Nv, normalize=True, coulomb=False):
    """src/utilities.jl:23-36.  The reference returns a closure K(v); this mirror returns the table of
    K evaluated at every V[n] (N × Nv, or N × 2Nv with coulomb), which is how the closure is used
    (src/lasso.jl:42-44)."""
    kv, pv, N = as_f64(V)
    nb = 2 * Nv if coulomb else Nv
    K = np.zeros((N, nb), order="F")
    check(lib().lpvs_basis_activation_f64(pv, N, int(Nv), int(bool(normalize)), int(bool(coulomb)), out_ptr(K)))
    return K


def lpv_regressor(X, V, w, Nv, normalize=True, coulomb=False, permuted=True):
    """Materialised Φ of src/lasso.jl:35-50 (tests / small problems; the solve never forms it)."""
    f32 = _all_f32(X, V, w)                           # the phases w .* X promote (src/lasso.jl:39): Float32 only for an all-Float32 call
    conv = as_f32 if f32 else as_f64
    kx, px, N = conv(X)
    kv, pv, _ = conv(V)
    kw, pw, Nf = conv(_host_vec(w) if not _lib.is_device_array(w) else w)
    nb = 2 * Nv if coulomb else Nv
    Phi = np.zeros((N, 2 * Nf * nb), order="F", dtype=np.float32 if f32 else np.float64)
    fn = lib().lpvs_lpv_regressor_f32 if f32 else lib().lpvs_lpv_regressor_f64
    check(fn(px, pv, N, pw, Nf, int(Nv), int(bool(normalize)), int(bool(coulomb)), int(bool(permuted)), out_ptr(Phi)))
    return Phi


def fourier2complex(x, zerofreq):
    """src/utilities.jl:62-73 (host formatting helper)."""
    x = _host(x)
    n = len(x) // 2
    if zerofreq is None:
        return x[:n] + 1j * x[n:]
    out = np.empty(n + 1, dtype=np.complex128)
    out[0] = x[0]
    out[1:] = x[1:n + 1] + 1j * x[n + 1:]
    return out


def _host_qr_ridge(A, y, lam):
    """``[A; λI] \\ [y; 0]`` by LAPACK on the host -- the reference's own route (``real_complex_bs`` / ``fourier_solve``,
    src/utilities.jl:49-60) for the cases the device normal equations reject as singular to working precision."""
    n = A.shape[1]
    return np.linalg.lstsq(np.vstack([A, lam * np.eye(n)]), np.concatenate([np.asarray(y, dtype=np.float64), np.zeros(n)]), rcond=None)[0]


# --------------------------------------------------------------------------- options (include/lpvspectral.h LPVS_OPT_*)
def set_default_option(name, value=None):
    """Thread-local default for handles created / batched-window calls made afterwards: ``set_default_option("storage", "f64")``;
    ``None`` returns to the library's own choice (which the environment variable of the same name may override)."""
    oid, vid = _lib.option_ids(name, value)
    check(lib().lpvs_set_default_option(oid, vid))


def get_default_option(name):
    oid, vals = _lib.OPTIONS[name]
    v = C.c_int32(0)
    check(lib().lpvs_get_default_option(oid, C.byref(v)))
    named = {i: k for k, i in vals.items()}
    if name in _lib._INT_OPTIONS:                                    # integer-valued options: the number itself (None = no explicit default)
        return named.get(int(v.value), int(v.value) if v.value != 0 else None)
    return named.get(int(v.value))


class default_options:
    """``with default_options(storage="f64", iteration="two"): ...`` -- the defaults inside the block, the previous ones after it."""

    def __init__(self, **opts):
        self.opts = {k: v for k, v in opts.items() if v is not None}

    def __enter__(self):
        self.prev = {k: get_default_option(k) for k in self.opts}
        for k, v in self.opts.items():
            set_default_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_default_option(k, v)


def _with_options(fn):
    """Estimator / driver decorator: option keywords apply, as thread defaults, to everything the call creates."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with default_options(**_pop_options(kwargs)):
            return fn(*args, **kwargs)
    return wrapper


def _pop_options(kwargs):
    """The option keywords of an estimator call (``storage=``, ``iteration=``, ``gram_form=``, ``nt_loads=``, ``slot_sums=``:
    extensions, the reference has none of them) taken out of ``kwargs``."""
    return {k: kwargs.pop(k) for k in list(kwargs) if k in _lib.OPTIONS}


# --------------------------------------------------------------------------- device problem handle
class Problem:
    """Owner of one ``lpvs_problem`` handle (regressor + Gram resident on one MI355X)."""

    ns = 1   # signals sharing the regressor (lpv_multi)
    f32 = False   # created from float32 inputs: float I/O, single-precision copy of M in the ADMM mat-vec

    def __init__(self, handle, kind):
        self._h = handle
        self.kind = kind
        n = C.c_int64(0)
        check(lib().lpvs_problem_size(self._h, C.byref(n)))
        self.n = int(n.value)

    # constructors -----------------------------------------------------------------------
    @classmethod
    def fourier(cls, y, t, f, W=None, device=0):
        # eltype as the reference promotes it: the 3-argument method evaluates the regressor in T = eltype(y) (``T.(t), T.(f)``,
        # src/lasso.jl:91); the weighted 4-argument method in the eltype of t and f themselves (src/lasso.jl:111 -> src/lsfft.jl:26),
        # so a Float32 record with Float64 time stamps is a Float64 problem there -- t is never rounded to 24 bits
        f32 = is_f32(y) if W is None else _all_f32(y, t, f)
        conv = as_f32 if f32 else as_f64
        ky, py, N = conv(y)
        kt, pt, Nt = conv(t)
        kf, pf, Nf = conv(f)
        kw, pw, Nw = conv(W)
        assert N == Nt, "y and t has to be the same length"
        assert W is None or Nw == N, "W has to be the same length as y"
        h = C.c_void_p()
        fn = lib().lpvs_problem_create_fourier_f32 if f32 else lib().lpvs_problem_create_fourier_f64
        check(fn(py, pt, N, pf, Nf, pw, int(device), C.byref(h)))
        p = cls(h, "fourier")
        p.Nf, p.f32 = Nf, f32
        return p

    @classmethod
    def lpv(cls, y, X, V, w, Nv, normalize=True, coulomb=False, device=0):
        # Y, X, V::AbstractVector{S} (src/lasso.jl:27); the phases w .* X are formed in the promoted type of w and X (:39), so the
        # Float32 entry points serve only an all-Float32 call -- anything else is widened exactly and runs as a Float64 problem
        f32 = _all_f32(y, X, V, w)
        conv = as_f32 if f32 else as_f64
        ky, py, N = conv(y)
        kx, px, Nx = conv(X)
        kv, pv, Nvv = conv(V)
        kw, pw, Nf = conv(w)
        assert N == Nx == Nvv, "y, X and V has to be the same length"
        h = C.c_void_p()
        fn = lib().lpvs_problem_create_lpv_f32 if f32 else lib().lpvs_problem_create_lpv_f64
        check(fn(py, px, pv, N, pw, Nf, int(Nv), int(bool(normalize)), int(bool(coulomb)), int(device), C.byref(h)))
        p = cls(h, "lpv")
        p.Nf, p.nb, p.f32 = Nf, (2 * Nv if coulomb else Nv), f32
        return p

    @classmethod
    def lpv_multi(cls, Y, X, V, w, Nv, normalize=True, coulomb=False, device=0):
        """``Y`` is N x ns (one column per signal / channel) sharing ``X, V, w``: one Gram, ns right-hand sides."""
        if _all_f32(Y, X, V, w) and not _lib.is_device_array(Y):   # an all-Float32 call: the _f32 entry point (float I/O, double arithmetic)
            Yh = np.asfortranarray(np.asarray(Y, dtype=np.float32))
            N, ns = Yh.shape
            kx, px, Nx = as_f32(np.asarray(X, dtype=np.float32))
            kv, pv, Nvv = as_f32(np.asarray(V, dtype=np.float32))
            kw, pw, Nf = as_f32(np.asarray(w, dtype=np.float32))
            assert N == Nx == Nvv, "Y, X and V has to have the same number of samples"
            h = C.c_void_p()
            check(lib().lpvs_problem_create_lpv_multi_f32(out_ptr(Yh), int(ns), px, pv, N, pw, Nf, int(Nv), int(bool(normalize)), int(bool(coulomb)),
                                                          int(device), C.byref(h)))
            p = cls(h, "lpv")
            p.Nf, p.nb, p.ns, p.f32 = Nf, (2 * Nv if coulomb else Nv), int(ns), True
            return p
        if _lib.is_device_array(Y):
            assert Y.dim() == 2
            ns, N = Y.shape[1], Y.shape[0]
            Yc = Y.t().contiguous()                 # column-major N x ns == row-major [ns][N]
            ky, py = Yc, C.c_void_p(Yc.data_ptr())
        else:
            Yh = np.asfortranarray(_host(Y))
            N, ns = Yh.shape
            ky, py = Yh, out_ptr(Yh)
        kx, px, Nx = as_f64(X)
        kv, pv, Nvv = as_f64(V)
        kw, pw, Nf = as_f64(w)
        assert N == Nx == Nvv, "Y, X and V has to have the same number of samples"
        h = C.c_void_p()
        check(lib().lpvs_problem_create_lpv_multi_f64(py, int(ns), px, pv, N, pw, Nf, int(Nv), int(bool(normalize)), int(bool(coulomb)), int(device), C.byref(h)))
        p = cls(h, "lpv")
        p.Nf, p.nb, p.ns = Nf, (2 * Nv if coulomb else Nv), int(ns)
        return p

    @classmethod
    def lpv_rows(cls, y, X, V, w, Nv, ranges, normalize=True, coulomb=False, device=0):
        """Partial problem over a ROW SHARD ``(y, X, V)`` of one signal: ``ranges`` =
        ``[min V, max V, max|V|, max|X|]`` over ALL rows.  Its Gram / rhs are partial sums; exchange them with
        :meth:`device_gram` + an all-reduce, then :meth:`gram_modified`."""
        f32 = _all_f32(y, X, V, w) and not _lib.is_device_array(y)       # an all-Float32 call: the _f32 entry point
        conv = as_f32 if f32 else as_f64
        ky, py, N = conv(np.asarray(y, dtype=np.float32) if f32 else y)
        kx, px, Nx = conv(np.asarray(X, dtype=np.float32) if f32 else X)
        kv, pv, Nvv = conv(np.asarray(V, dtype=np.float32) if f32 else V)
        kw, pw, Nf = conv(np.asarray(w, dtype=np.float32) if f32 else w)
        assert N == Nx == Nvv, "y, X and V has to be the same length"
        r4 = np.ascontiguousarray(np.asarray(ranges, dtype=np.float64).ravel())
        assert r4.size == 4
        h = C.c_void_p()
        fn = lib().lpvs_problem_create_lpv_rows_f32 if f32 else lib().lpvs_problem_create_lpv_rows_f64
        check(fn(py, 1, px, pv, N, pw, Nf, int(Nv), int(bool(normalize)), int(bool(coulomb)), out_ptr(r4), int(device), C.byref(h)))
        p = cls(h, "lpv")
        p.Nf, p.nb = Nf, (2 * Nv if coulomb else Nv)
        p.f32 = f32
        return p

    @classmethod
    def dense(cls, A, y, W=None, device=0):
        if _lib.is_device_array(A):
            raise TypeError("dense(): pass A as a host (numpy) column-major matrix or a transposed-contiguous device tensor via gram()")
        Ah = np.asfortranarray(_host(A))
        m, n = Ah.shape
        ky, py, N = as_f64(y)
        kw, pw, Nw = as_f64(W)
        assert N == m
        h = C.c_void_p()
        check(lib().lpvs_problem_create_dense_f64(out_ptr(Ah), py, m, n, pw, int(device), C.byref(h)))
        return cls(h, "dense")

    @classmethod
    def gram(cls, G, b, device=0):
        kg, pg, n2 = as_f64(G)
        kb, pb, n = as_f64(b)
        assert n2 == n * n, "G must be n x n"
        h = C.c_void_p()
        check(lib().lpvs_problem_create_gram_f64(pg, pb, n, int(device), C.byref(h)))
        return cls(h, "gram")

    # housekeeping ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib().lpvs_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # options --------------------------------------------------------------------------------
    def set_option(self, name, value=None):
        """``storage`` = "mixed" | "split" | "f64", ``iteration`` = "one" | "two", ``nt_loads`` = "on" | "off" (``None``: default);
        takes effect at the next :meth:`admm_init` / :meth:`admm_run`."""
        oid, vid = _lib.option_ids(name, value)
        check(lib().lpvs_problem_set_option(self._h, oid, vid))

    def get_option(self, name):
        """The value in effect for this handle (explicit, thread default or environment), ``None`` = the library's own choice."""
        oid, vals = _lib.OPTIONS[name]
        v = C.c_int32(0)
        check(lib().lpvs_problem_get_option(self._h, oid, C.byref(v)))
        return {i: k for k, i in vals.items()}.get(int(v.value))

    # accessors ------------------------------------------------------------------------------
    def get_gram(self):
        G = np.zeros((self.n, self.n), order="F")
        b = np.zeros(self.n)
        check(lib().lpvs_problem_get_gram_f64(self._h, out_ptr(G), out_ptr(b)))
        return G, b

    def get_rhs(self):
        b = np.zeros(self.n if self.ns == 1 else (self.n, self.ns), order="F")
        check(lib().lpvs_problem_get_rhs_f64(self._h, out_ptr(b)))
        return b

    def device_gram(self):
        """Torch views (no copy) of the handle's device Gram ``[np, np]`` and right-hand sides ``[ns, np]`` for an
        in-place exchange step; call :meth:`gram_modified` afterwards."""
        import torch
        G, b, npad = C.c_void_p(), C.c_void_p(), C.c_int64(0)
        check(lib().lpvs_problem_device_gram_f64(self._h, C.byref(G), C.byref(b), C.byref(npad)))
        n_p = int(npad.value)

        class _View:   # __cuda_array_interface__ is how torch (CUDA and ROCm builds) adopts foreign device memory
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None}

        dev = torch.device("cuda", torch.cuda.current_device())
        Gt = torch.as_tensor(_View(G.value, (n_p, n_p)), device=dev)
        bt = torch.as_tensor(_View(b.value, (self.ns, n_p)), device=dev)
        return Gt, bt

    def gram_modified(self):
        check(lib().lpvs_problem_gram_modified(self._h))

    def get_inverse(self, shift):
        M = np.zeros((self.n, self.n), order="F")
        check(lib().lpvs_problem_get_inverse_f64(self._h, float(shift), out_ptr(M)))
        return M

    def solve_ridge(self, ridge):
        x = np.zeros(self.n, dtype=np.float32 if self.f32 else np.float64)
        fn = lib().lpvs_problem_solve_ridge_f32 if self.f32 else lib().lpvs_problem_solve_ridge_f64
        check(fn(self._h, float(ridge), out_ptr(x)))
        return x

    def set_prox(self, proxg):
        if not hasattr(proxg, "device_params"):
            raise NotImplementedError(f"proxg of type {type(proxg).__name__} has no device kernel "
                                      "(supported: NormL1, NormL0, IndBallL0, SlicedSeparableSum(NormL2))")
        kind, param, glen = proxg.device_params(self.n)
        check(lib().lpvs_problem_set_prox(self._h, kind, float(param), int(glen)))

    def admm_init(self, x0=None, μ=0.05, tol=1e-5, linear_sign=_lib.LINEAR_LEAST_SQUARES):
        k0, p0, n0 = (as_f32 if self.f32 else as_f64)(x0)
        assert x0 is None or n0 == self.n, "x0 has the wrong length"
        fn = lib().lpvs_admm_init_f32 if self.f32 else lib().lpvs_admm_init_f64
        check(fn(self._h, p0, float(μ), float(tol), int(linear_sign)))

    def admm_run(self, max_iters):
        it, nxz, conv = C.c_int64(0), C.c_double(0), C.c_int32(0)
        check(lib().lpvs_admm_run(self._h, int(max_iters), C.byref(it), C.byref(nxz), C.byref(conv)))
        return int(it.value), float(nxz.value), bool(conv.value)

    def admm_set_state(self, x, z, u, iters=0, offset=None):
        """Resume: install iterates saved by :meth:`admm_get` (after :meth:`admm_init` with the same parameters).  ``offset`` = the vector
        :meth:`admm_get_offset` returned with them (handles of n >= 2048): with it the run continues bit for bit, without it the library
        re-forms the vector from ``x`` (the same to second order)."""
        arrs = [np.asfortranarray(np.asarray(a, dtype=np.float32 if self.f32 else np.float64)) for a in (x, z, u)]
        assert all(a.size == self.n * self.ns for a in arrs), "x, z, u have the wrong length"
        fn = lib().lpvs_admm_set_state_f32 if self.f32 else lib().lpvs_admm_set_state_f64
        check(fn(self._h, out_ptr(arrs[0]), out_ptr(arrs[1]), out_ptr(arrs[2]), int(iters)))
        if offset is not None:
            off = np.asfortranarray(np.asarray(offset, dtype=np.float64))
            check(lib().lpvs_admm_set_offset_f64(self._h, out_ptr(off), int(off.size)))   # (a wrong length is the library's LPVS_EARGUMENT -> ValueError)

    def _offset_len(self):
        k = C.c_int64(0)
        check(lib().lpvs_admm_offset_len(self._h, C.byref(k)))
        return int(k.value)

    def admm_get_offset(self):
        """The offset vector of the x-update currently in effect (doubles), or ``None`` for handles without one (n < 2048).  Opaque:
        n x ns values, or 2 n for a handle that iterates on 32-bit reads of its fixed-point tiles (the vector with and without the nibble
        term of the last refresh; the first n are the offset the x-update adds)."""
        k = self._offset_len()
        if k == 0:
            return None
        xb = np.zeros(k if k != self.n * self.ns or self.ns == 1 else (self.n, self.ns), order="F")
        check(lib().lpvs_admm_get_offset_f64(self._h, out_ptr(xb), int(xb.size)))
        return xb

    def admm_get(self):
        shape = self.n if self.ns == 1 else (self.n, self.ns)
        x, z, u = (np.zeros(shape, order="F", dtype=np.float32 if self.f32 else np.float64) for _ in range(3))
        fn = lib().lpvs_admm_get_f32 if self.f32 else lib().lpvs_admm_get_f64
        check(fn(self._h, out_ptr(x), out_ptr(z), out_ptr(u)))
        return x, z, u

    def time_matvec(self, reps=200):
        """(microseconds per launch, bytes of M streamed per launch) of the ADMM mat-vec kernel, HIP events on the handle's stream."""
        us, nbytes = C.c_double(0), C.c_double(0)
        check(lib().lpvs_admm_time_matvec(self._h, int(reps), C.byref(us), C.byref(nbytes)))
        return float(us.value), float(nbytes.value)

    def matvec_info(self):
        """Kernel / storage of the inverse the ADMM mat-vec of this handle streams (after :meth:`admm_init`)."""
        k = C.c_int32(0)
        check(lib().lpvs_admm_matvec_kind(self._h, C.byref(k)))
        np_ = -(-self.n // 128) * 128
        one_launch = bool(int(k.value) & 16)
        fix32 = bool(int(k.value) & 32)            # the fixed-point tiles keep 32 significant bits (handles whose x-update is corrected)
        fixs = ("36-bit fixed point with per-row steps of which the iteration reads the 32 leading bits (4.03 B; the 4-bit planes meet a right-hand side "
                "every 32 iterations -- more often in the first 256 -- and ride in the offset vector: 32-bit fixed point reads)" if fix32 else
                "36-bit fixed point with per-row steps (4.53 B)")
        fixb = 66048 if fix32 else 74240
        k = C.c_int32(int(k.value) & 15)
        if one_launch and int(k.value) == 0:                               # small problems (np < 2048): full matrix, one launch per iteration
            return dict(kernel="admm_small_iter_kernel", one_launch_iteration=True,
                        storage="full symmetric f64; every workgroup redoes the update (prox, dual step, norm) and multiplies its four rows",
                        bytes_formula="8 B x np^2 (np = %d)" % np_)
        if one_launch and int(k.value) == 2:                               # _f32 handles: single-precision copy of the inverse
            return dict(kernel="admm_iter_mixed_kernel", one_launch_iteration=True,
                        storage="tile-packed lower triangle, f32 (4 B); tile partials added into x by 64-bit fixed-point atomics, prox / dual "
                                "update in the next launch's prologue",
                        bytes_formula="4 B x np(np+128)/2 (np = %d)" % np_)
        if one_launch:                                                     # mixed storage, single signal, fusable prox
            return dict(kernel="admm_iter_mixed_kernel", one_launch_iteration=True,
                        storage="tile-packed lower triangle, mixed: float head + 16-bit tail (6 B, 40 significant bits) for the diagonal tiles, "
                                + fixs + " for tiles of small entries; tile partials added into x by 64-bit "
                                "fixed-point atomics, prox / dual update in the next launch's prologue",
                        bytes_formula="98304 B per 6-byte tile, %d B per fixed-point tile (np = %d; 8-byte form %.1f MB)" % (fixb, np_, 8e-6 * np_ * (np_ + 128) / 2))
        if self.ns > 1 and int(k.value) == 4:                              # fixed-point tiles below the diagonal, float-head tiles on it
            q4 = self.ns <= 8 and os.environ.get("LPVS_MULTI_MFMA") != "16"
            nb = np_ // 128
            return dict(kernel="symv_tile_mfma_ws_kernel", storage="tile-packed lower triangle, mixed: 36-bit fixed point with per-row steps (4.53 B) for the "
                        "tiles below the diagonal, float head + 16-bit tail (6 B) for the diagonal tiles",
                        mfma="v_mfma_f64_4x4x4_4b_f64 (8 signal columns per pass, none padded)" if q4 else "v_mfma_f64_16x16x4_f64 (16 signal columns per pass)",
                        signals_per_pass=8 if q4 else 16,
                        bytes_formula="74240 B x %d tiles below the diagonal + 98304 B x %d diagonal tiles (np = %d), streamed once per %d signals"
                                      % (nb * (nb - 1) // 2, nb, np_, 8 if q4 else 16))
        if self.ns > 1 and int(k.value) in (1, 3):
            elt = 8 if int(k.value) == 1 else 6
            q4 = self.ns <= 8 and os.environ.get("LPVS_MULTI_MFMA") != "16"
            return dict(kernel="symv_tile_mfma_ws_kernel", storage="tile-packed lower triangle, %s" % ("f64 (8 B)" if elt == 8 else "float head + 16-bit tail (6 B, 40 significant bits)"),
                        mfma="v_mfma_f64_4x4x4_4b_f64 (8 signal columns per pass, none padded)" if q4 else "v_mfma_f64_16x16x4_f64 (16 signal columns per pass)",
                        signals_per_pass=8 if q4 else 16,
                        bytes_formula="%d B x np(np+128)/2 (np = %d), streamed once per %d signals" % (elt, np_, 8 if q4 else 16))
        return {0: dict(kernel="symv_kernel", storage="full symmetric f64", bytes_formula="8 B x np^2"),
                1: dict(kernel="symv_tile_kernel<double>", storage="tile-packed lower triangle, f64 (8 B)",
                        bytes_formula="8 B x np(np+128)/2 (np = %d)" % np_),
                2: dict(kernel="symv_tile_f32_kernel" if self.ns == 1 else "symv_tile_kernel<float>", storage="tile-packed lower triangle, f32 (4 B)",
                        bytes_formula="4 B x np(np+128)/2 (np = %d)" % np_),
                4: dict(kernel="symv_tile_mixed_kernel", storage="tile-packed lower triangle, mixed: float head + 16-bit tail (6 B, 40 significant bits) "
                                                                 "for the diagonal tiles, " + fixs + " for tiles of small entries",
                        bytes_formula="98304 B per 6-byte tile, %d B per fixed-point tile (np = %d; 8-byte form %.1f MB)" % (fixb, np_, 8e-6 * np_ * (np_ + 128) / 2)),
                3: dict(kernel="symv_tile_split_kernel", storage="tile-packed lower triangle, float head + 16-bit tail (6 B, 40 significant bits)",
                        bytes_formula="6 B x np(np+128)/2 (np = %d; the 8-byte form would be %.1f MB)" % (np_, 8e-6 * np_ * (np_ + 128) / 2))}[int(k.value)]

    def admm_status(self, signal=0):
        it, nxz, conv = C.c_int64(0), C.c_double(0), C.c_int32(0)
        check(lib().lpvs_admm_status(self._h, int(signal), C.byref(it), C.byref(nxz), C.byref(conv)))
        return int(it.value), float(nxz.value), bool(conv.value)

    def params(self, which=0):
        m = self.Nf * self.nb if self.kind == "lpv" else self.Nf
        shape = m if self.ns == 1 else (m, self.ns)
        dt = np.float32 if self.f32 else np.float64
        re, im = np.zeros(shape, order="F", dtype=dt), np.zeros(shape, order="F", dtype=dt)
        fn = lib().lpvs_problem_get_params_f32 if self.f32 else lib().lpvs_problem_get_params_f64
        check(fn(self._h, int(which), out_ptr(re), out_ptr(im)))
        return (re + 1j * im).astype(np.complex64 if self.f32 else np.complex128)

    def pack(self, coef):
        m = self.Nf * self.nb if self.kind == "lpv" else self.Nf
        kc, pc, _ = as_f64(coef)
        re, im = np.zeros(m), np.zeros(m)
        check(lib().lpvs_problem_pack_params_f64(self._h, pc, out_ptr(re), out_ptr(im)))
        return re + 1j * im

    def timing(self):
        t = np.zeros(13)
        check(lib().lpvs_problem_get_timing(self._h, out_ptr(t), 13))
        return dict(basis_ms=t[0], gram_ms=t[1], reduce_rhs_ms=t[2], factor_ms=t[3], admm_ms=t[4],
                    gram_issued_flops=t[5], gram_flops=t[6], admm_iters=t[7],
                    gram_form=("given", "kr", "krs", "panel", "ap", "ap-nufft")[int(t[8])],
                    xcorr_ms=t[9], xcorr_count=int(t[10]),      # the x-update corrections inside admm_ms
                    nibble_refreshes=int(t[11]), nibble_refresh_us=t[12])   # the stale nibble product's refreshes inside admm_ms; one of them, stand-alone


# --------------------------------------------------------------------------- ADMM driver
def _admm_on_problem(prob: Problem, x0, proxg, linear_sign, iters=10000, tol=1e-5, printerval=100, cb=None, μ=0.05,
                     out=sys.stdout):
    """src/lasso.jl:136-171 with the iterations on device.  Control returns to the host every
    ``printerval`` iterations, so the progress lines (:159,:165), ``cb(x,z)`` (:160-162) and
    KeyboardInterrupt (:59-63) behave as in the reference."""
    assert 0 <= μ <= 1, "μ should be ≤ 1"                       # src/lasso.jl:143
    prob.set_prox(proxg)
    prob.admm_init(x0, μ=μ, tol=tol, linear_sign=linear_sign)
    done, conv, nxz = 0, False, 0.0
    printerval = int(printerval) if printerval and printerval > 0 else iters
    while done < iters and not conv:
        chunk = min(printerval - done % printerval, iters - done)
        done, nxz, conv = prob.admm_run(chunk)
        if done % printerval == 0:
            print("%d ||x-z||₂ %.10f" % (done, nxz), file=out)      # src/lasso.jl:159
            if cb is not None:                                      # src/lasso.jl:160-162
                x, z, _ = prob.admm_get()
                cb(x, z)
        if conv:
            print("%d ||x-z||₂ %.10f" % (done, nxz), file=out)      # src/lasso.jl:165
            log.info("||x-z||₂ ≤ tol")                              # src/lasso.jl:166
            break
    x, z, _ = prob.admm_get()
    return x, z


def ADMM(x, proxf, proxg, iters=10000, tol=1e-5, printerval=100, cb=None, μ=0.05, device=0):
    """``ADMM(x, proxf, proxg; iters, tol, printerval, cb, μ)`` (src/lasso.jl:136-171) -> ``(x, z)``.

    ``proxf`` is a :class:`LeastSquares` (dense A, b) or :class:`Quadratic` (Q, q) mirror object; its
    Gram is formed / uploaded once and every iteration runs on the GPU."""
    if isinstance(proxf, LeastSquares):
        prob = Problem.dense(proxf.A, proxf.b, device=device)
    elif isinstance(proxf, Quadratic):
        prob = Problem.gram(proxf.Q, proxf.q, device=device)
    else:
        raise NotImplementedError(f"proxf of type {type(proxf).__name__} has no device path")
    with prob:
        return _admm_on_problem(prob, x, proxg, proxf.linear_sign, iters, tol, printerval, cb, μ)


# --------------------------------------------------------------------------- estimators
def ls_spectral(y, t, f=None, W=None, λ=1e-10, verbose=False, device=0):
    """``ls_spectral(y,t,f=default_freqs(t); λ=1e-10)`` (src/lsfft.jl:62-67) and the weighted
    ``ls_spectral(y,t,f,W; λ)`` (:74-80) -> ``(x, f)``.

    The weighted form is the reference's own formula ``(A'WA + λI) \\ A'Wy`` on the device Gram.  The
    unweighted form ``[A; λI] \\ [y; 0]`` is solved from normal equations in the better-conditioned shape:
    primal ``(A'A+λ²I)⁻¹A'y`` for tall systems, dual ``A'(AA'+λ²I)⁻¹y`` for fat ones (the reference's default
    frequency grid on an even-length record has one more column than rows) -- identical minimiser, no SVD."""
    f = default_freqs(t) if f is None else f
    if W is None:
        f32 = _all_f32(y, t, f)                                      # get_fourier_regressor(t, f) in their own eltype, src/lsfft.jl:63
        conv = as_f32 if f32 else as_f64
        ky, py, N = conv(y)
        kt, pt, Nt = conv(t)
        kf, pf, Nf = conv(f)
        assert N == Nt, "y and t has to be the same length"
        dt = np.float32 if f32 else np.float64
        re, im = np.zeros(Nf, dtype=dt), np.zeros(Nf, dtype=dt)
        fn = lib().lpvs_ls_spectral_f32 if f32 else lib().lpvs_ls_spectral_f64
        check(fn(py, pt, N, pf, Nf, float(λ), int(device), out_ptr(re), out_ptr(im)))
        if verbose:                                                  # src/lsfft.jl:65: cond(A'A), from the device Gram
            with Problem.fourier(y, t, f, None, device=device) as prob:
                G, _ = prob.get_gram()
            log.info("Condition number: %s\n", round(float(np.linalg.cond(G)), 2))
        return (re + 1j * im).astype(np.complex64 if f32 else np.complex128), _host(f)
    with Problem.fourier(y, t, f, W, device=device) as prob:
        x = prob.solve_ridge(λ)
        if verbose:
            G, _ = prob.get_gram()
            log.info("Condition number: %s\n", round(float(np.linalg.cond(G)), 2))
        params = prob.pack(x)
    return params, _host(f)


def tls_spectral(y, t, f=None, device=0):
    """``tls_spectral(y,t,f=default_freqs(t)[1:end-1])`` (src/lsfft.jl:85-99) -> ``(x, f)``: total least squares.

    The reference takes the right singular vector of the smallest singular value of ``[A y]``; it is the eigenvector of
    the smallest eigenvalue of ``[A y]'[A y] = [[A'A, A'y], [y'A, y'y]]``.  The O(N n²) part -- ``A'A`` and ``A'y`` -- is
    the device Gram of the Fourier problem; the (n+1)×(n+1) symmetric eigenproblem goes to the host LAPACK, as the
    reference's own ``LAPACK.gesvd!`` does."""
    f = default_freqs(t)[:-1] if f is None else f
    yh = _host(y).astype(np.float64)
    with Problem.fourier(y, t, f, None, device=device) as prob:
        G, b = prob.get_gram()
        n = prob.n
        H = np.empty((n + 1, n + 1))
        H[:n, :n] = G
        H[:n, n] = b
        H[n, :n] = b
        H[n, n] = float(np.dot(yh, yh))
        _, V = np.linalg.eigh(H)
        v = V[:, 0]                                                  # smallest eigenvalue first
        params = prob.pack(-v[:n] / v[n])                            # x = -V21 / V22, fourier2complex
    return params, _host(f)


def ls_sparse_spectral(y, t, f=None, W=None, init=False, λ=1.0, proxg=None, device=0, **kwargs):
    """``ls_sparse_spectral(y,t,f; init, λ, proxg=NormL1(λ), kwargs...)`` (src/lasso.jl:85-102) and the
    weighted 4-argument method ``(y,t,f,W; ...)`` (:105-126) -> ``(params, f)``.

    The weighted method reproduces the reference as written: ``Quadratic(Q, q=+A'Wy)``
    (src/lasso.jl:119-121), i.e. the linear term enters with the opposite sign of least squares."""
    f = default_freqs(t) if f is None else f
    proxg = NormL1(λ) if proxg is None else proxg
    with default_options(**_pop_options(kwargs)), Problem.fourier(y, t, f, W, device=device) as prob:
        x0 = None
        if init:  # fourier_solve(A,y,zerofreq,λ), src/lasso.jl:92,112 -- UNWEIGHTED in both methods (:112 ignores W)
            zf = check_freq(f)
            if W is None:
                p = prob.pack(prob.solve_ridge(λ * λ))               # the handle's own Gram is the unweighted one
            else:
                p = ls_spectral(y, t, f, λ=λ, device=device)[0]
            p = np.asarray(p, dtype=np.complex128)
            x0 = np.concatenate([p.real, p.imag[1:] if zf else p.imag])  # src/lasso.jl:93-97
        sign = _lib.LINEAR_LEAST_SQUARES if W is None else _lib.LINEAR_QUADRATIC_AS_WRITTEN
        _admm_on_problem(prob, x0, proxg, sign, **kwargs)
        params = prob.params(0)                                      # fourier2complex(z, zerofreq)
    return params, _host(f)


def ls_sparse_spectral_lpv(y, X, V, w, Nv, λ=1, coulomb=False, normalize=True, device=0, proxg=None, **kwargs):
    """``ls_sparse_spectral_lpv(Y,X,V,w,Nv; λ, coulomb, normalize, kwargs...)`` (src/lasso.jl:27-70)
    -> :class:`SpectralExt` with ``Σ=None``.

    ``proxg=None`` gives the reference's frequency-grouped lasso (src/lasso.jl:53-55); passing another
    prox object (e.g. ``IndBallL0(32)``) is an extension the reference does not have."""
    if coulomb:
        raise NotImplementedError("coulomb=true is ill-defined in the reference's sparse LPV path "
                                  "(half of x is never written by prox!, SURVEY.md §2 ‡); use ls_spectral_lpv")
    w = _host_vec(w) if not _lib.is_device_array(w) else w
    Nf = len(w)
    Nv = int(Nv)
    with default_options(**_pop_options(kwargs)), Problem.lpv(y, X, V, w, Nv, normalize, False, device=device) as prob:
        g = SlicedSeparableSum.frequency_groups(λ, Nf, 2 * Nv) if proxg is None else proxg
        try:
            _admm_on_problem(prob, None, g, _lib.LINEAR_LEAST_SQUARES, **kwargs)
            params = prob.params(0)
        except KeyboardInterrupt:                                    # src/lasso.jl:59-63
            log.info("Aborting")
            params = prob.params(1)                                  # z = copy(x)
    return SpectralExt(y, X, V, w, Nv, λ, coulomb, normalize, params, None)


def lpv_ranges(X, V):
    """``[min V, max V, max|V|, max|X|]`` of a (shard of a) signal, reduced on the device."""
    kx, px, N = as_f64(X)
    kv, pv, Nv_ = as_f64(V)
    assert N == Nv_
    out = np.zeros(4)
    check(lib().lpvs_lpv_ranges_f64(px, pv, N, out_ptr(out)))
    return out


def ls_sparse_spectral_lpv_rowsharded(y, X, V, w, Nv, λ=1, normalize=True, device=0, proxg=None, dist=None, **kwargs):
    """One signal whose SAMPLE ROWS are sharded over the ranks of ``dist`` (``torch.distributed``, initialised;
    backend nccl = RCCL over xGMI): every rank passes its rows ``(y, X, V)``.  SURVEY.md §8(e)(2): each rank forms the
    partial Gram / rhs of its rows on its GPU, ONE all-reduce sums them (n_p² + n_p f64 per rank), and the ADMM
    (src/lasso.jl:136-171) then runs replicated, so every rank returns the same :class:`SpectralExt` as
    :func:`ls_sparse_spectral_lpv` on the whole signal (up to the summation order of the Gram)."""
    from . import sharding
    w = _host_vec(w) if not _lib.is_device_array(w) else w
    Nf, Nv = len(w), int(Nv)
    ranges = sharding.allreduce_ranges(lpv_ranges(X, V), dist)
    with Problem.lpv_rows(y, X, V, w, Nv, ranges, normalize, False, device=device) as prob:
        G, b = prob.device_gram()
        sharding.allreduce_sum_(G, dist)
        sharding.allreduce_sum_(b, dist)
        prob.gram_modified()
        g = SlicedSeparableSum.frequency_groups(λ, Nf, 2 * Nv) if proxg is None else proxg
        _admm_on_problem(prob, None, g, _lib.LINEAR_LEAST_SQUARES, **kwargs)
        params = prob.params(0)
    return SpectralExt(y, X, V, w, Nv, λ, False, normalize, params, None)


def ls_sparse_spectral_lpv_multi(Y, X, V, w, Nv, λ=1, normalize=True, device=0, proxg=None, **kwargs):
    """Batched multichannel form of :func:`ls_sparse_spectral_lpv` (extension; BASELINE.json config 5): the columns
    of ``Y`` (N x ns) are independent signals sharing ``X, V, w``.  One Gram / factorisation, every ADMM kernel
    advances all signals; each signal stops at its own ``‖x−z‖₂ < tol``.  Returns a list of :class:`SpectralExt`
    whose entries equal the single-signal results."""
    w = _host_vec(w) if not _lib.is_device_array(w) else w
    Nf, Nv = len(w), int(Nv)
    with default_options(**_pop_options(kwargs)), Problem.lpv_multi(Y, X, V, w, Nv, normalize, False, device=device) as prob:
        g = SlicedSeparableSum.frequency_groups(λ, Nf, 2 * Nv) if proxg is None else proxg
        _admm_on_problem(prob, None, g, _lib.LINEAR_LEAST_SQUARES, **kwargs)
        P = prob.params(0)
        ns = prob.ns
    P = P.reshape(-1, ns, order="F")
    Yh = Y if not _lib.is_device_array(Y) else None
    return [SpectralExt(None if Yh is None else np.asarray(Yh)[:, q], X, V, w, Nv, λ, False, normalize, P[:, q].copy(), None)
            for q in range(ns)]


def lpv_batch_multi(Y, X, V, w, Nv, proxg=None, λ=1, normalize=True, μ=0.05, tol=1e-5, iters=10000, ngpus=0, devices=None):
    """``lpvs_lpv_batch_multi_f64``: the channels (columns of ``Y``, N x ns) split into contiguous ranges over ``ngpus`` devices driven
    by this one process (a host thread per device: one Gram / factorisation per device, its channels advanced together).  Returns
    ``(params [Nf*Nv x ns complex], iters [ns])``; no progress lines / callbacks (use :func:`ls_sparse_spectral_lpv_multi` per device
    for those)."""
    assert 0 <= μ <= 1, "μ should be ≤ 1"                             # src/lasso.jl:143
    w = _host_vec(w).astype(np.float64)
    Yh = np.asfortranarray(_host(Y))
    N, ns = Yh.shape
    Xh, Vh = np.ascontiguousarray(_host(X)), np.ascontiguousarray(_host(V))
    assert N == len(Xh) == len(Vh), "Y, X and V has to have the same number of samples"
    Nf, Nv = len(w), int(Nv)
    g = SlicedSeparableSum.frequency_groups(λ, Nf, 2 * Nv) if proxg is None else proxg
    kind, param, glen = g.device_params(2 * Nf * Nv)
    dv = None if devices is None else np.ascontiguousarray(np.asarray(devices, dtype=np.int32))
    if dv is not None:
        ngpus = len(dv)
    m = Nf * Nv
    re, im = np.zeros((m, ns), order="F"), np.zeros((m, ns), order="F")
    its = np.zeros(ns, dtype=np.int64)
    check(lib().lpvs_lpv_batch_multi_f64(out_ptr(Yh), ns, out_ptr(Xh), out_ptr(Vh), N, out_ptr(w), Nf, Nv, int(bool(normalize)), int(kind), float(param),
                                         int(glen), float(μ), float(tol), int(iters), None if dv is None else out_ptr(dv), int(ngpus),
                                         out_ptr(re), out_ptr(im), out_ptr(its)))
    return re + 1j * im, its


def lpv_signals_multi(Y, X, V, w, Nv, proxg=None, λ=1, normalize=True, μ=0.05, tol=1e-5, iters=10000, ngpus=0, devices=None, in_flight=2):
    """``lpvs_lpv_signals_multi_f64``: the loop ``[ls_sparse_spectral_lpv(Y[:, q], X[:, q], V[:, q], w, Nv; ...) for q]`` inside the
    library -- every signal its own samples of X and V (all three N x nsig), contiguous signal ranges over ``ngpus`` devices,
    ``in_flight`` solves at a time per device (each on its own handle and stream).  Returns ``(params [Nf*Nv x nsig complex],
    iters [nsig])``; the results do not depend on ``ngpus`` / ``in_flight``."""
    assert 0 <= μ <= 1, "μ should be ≤ 1"                             # src/lasso.jl:143
    w = _host_vec(w).astype(np.float64)
    dev_in = all(_lib.is_device_array(a) for a in (Y, X, V))
    if dev_in:                                                        # resident inputs (column-major N x nsig on the device) are used in place
        import torch
        if not all(a.dtype == torch.float64 for a in (Y, X, V)):     # the _f64 entry point reads 8-byte elements: never reinterpret
            raise TypeError("lpv_signals_multi: device inputs must be float64 tensors (got %s)" % ", ".join(str(a.dtype) for a in (Y, X, V)))
        if not (Y.device == X.device == V.device):
            raise ValueError("lpv_signals_multi: Y, X and V must live on the same device (got %s, %s, %s)" % (Y.device, X.device, V.device))
        if Y.dim() != 2:
            raise ValueError("lpv_signals_multi: device inputs are N x nsig matrices")
        Yh, Xh, Vh = Y, X, V
        N, nsig = int(Y.shape[0]), int(Y.shape[1])
        assert tuple(X.shape) == tuple(V.shape) == (N, nsig), "Y, X and V has to have the same number of samples"
        assert all(a.stride(0) == 1 and a.stride(1) == N for a in (Y, X, V)), "device inputs must be column-major (N x nsig, e.g. A.T.contiguous().T)"
    else:
        Yh, Xh, Vh = (np.asfortranarray(_host(a), dtype=np.float64) for a in (Y, X, V))
        N, nsig = Yh.shape
        assert Xh.shape == Vh.shape == (N, nsig), "Y, X and V has to have the same number of samples"
    Nf, Nv = len(w), int(Nv)
    g = SlicedSeparableSum.frequency_groups(λ, Nf, 2 * Nv) if proxg is None else proxg
    kind, param, glen = g.device_params(2 * Nf * Nv)
    dv = None if devices is None else np.ascontiguousarray(np.asarray(devices, dtype=np.int32))
    if dv is not None:
        ngpus = len(dv)
    m = Nf * Nv
    re, im = np.zeros((m, nsig), order="F"), np.zeros((m, nsig), order="F")
    its = np.zeros(nsig, dtype=np.int64)
    ptr = (lambda a: C.c_void_p(a.data_ptr())) if dev_in else out_ptr
    if dev_in:
        import torch
        torch.cuda.current_stream(Y.device).synchronize()             # the library works on its own streams
    check(lib().lpvs_lpv_signals_multi_f64(ptr(Yh), ptr(Xh), ptr(Vh), N, nsig, out_ptr(w), Nf, Nv, int(bool(normalize)), int(kind), float(param),
                                           int(glen), float(μ), float(tol), int(iters), None if dv is None else out_ptr(dv), int(ngpus), int(in_flight),
                                           out_ptr(re), out_ptr(im), out_ptr(its)))
    return re + 1j * im, its


def ls_spectral_lpv(Y, X, V, w, Nv, λ=1e-8, coulomb=False, normalize=True, device=0, covariance=True):
    """``ls_spectral_lpv(Y,X,V,w,Nv; λ, coulomb, normalize)`` (src/lsfft.jl:239-259) -> :class:`SpectralExt`.

    Ridge solve ``[Ar; λI] \\ [Y; 0]`` in normal-equation form on the device Gram (built in the permuted column
    order; ridge and solution are permutation-invariant and are un-permuted by the same packing as the sparse path).
    The residual statistics come from the same Gram, with the constant signal as a second right-hand side:
    ``‖e‖² = x'Gx − 2b'x + Y'Y``, ``Σe = (Φ'1)'x − ΣY`` ⇒ ``var(e)``; ``Σ = var(e)·(AA'AA + λI)⁻¹`` (:252-254, λ not
    squared, as written) in the reference's ``[re; im]`` parameter order; the fva warning of :255-256 is issued."""
    w = _host_vec(w) if not _lib.is_device_array(w) else w
    Nv = int(Nv)
    Yh = _host(Y)
    def ridge(prob):
        try:
            return prob.solve_ridge(λ * λ)
        except _lib.NumericError as e:
            # (G + λ²I) is singular to working precision (e.g. more unknowns than samples with the default λ = 1e-8): the
            # normal equations cannot reproduce the reference there, its QR of [Ar; λI] can.  Same route, host LAPACK, on
            # the regressor the device kernel materialises (src/utilities.jl:49-54).
            log.info("ls_spectral_lpv: %s; solving [A; λI] \\ [Y; 0] by QR on the host", e)
            Phi = lpv_regressor(X, V, w, Nv, normalize, coulomb, permuted=True)
            return _host_qr_ridge(Phi, Yh, λ)

    if not covariance:
        with Problem.lpv(Y, X, V, w, Nv, normalize, coulomb, device=device) as prob:
            params = prob.pack(ridge(prob))
        return SpectralExt(Y, X, V, w, Nv, λ, coulomb, normalize, params, None)
    Y2 = np.stack([Yh, np.ones_like(Yh)], axis=1)
    with Problem.lpv_multi(Y2, X, V, w, Nv, normalize, coulomb, device=device) as prob:
        x = ridge(prob)                                  # first right-hand side = Y
        params = prob.pack(x)
        G, _ = prob.get_gram()
        B = prob.get_rhs()
        try:
            Minv = prob.get_inverse(λ)
        except _lib.NumericError:
            Minv = np.linalg.inv(G + λ * np.eye(G.shape[0]))   # inv(AA'AA + λI) as the reference evaluates it (:253)
        Nf, nb = prob.Nf, prob.nb
    N = len(Yh)
    e2 = float(x @ (G @ x) - 2.0 * (B[:, 0] @ x) + Yh @ Yh)         # ‖AA·x − Y‖²
    esum = float(B[:, 1] @ x - Yh.sum())
    var_e = (e2 - esum * esum / N) / (N - 1)                         # var(e), corrected (Statistics.var)
    var_y = float(np.var(Yh, ddof=1))
    # un-permute: reference order u = c·Nf·nb + j·Nf + f  <-  device order p = f·2nb + c·nb + j
    f_, c_, j_ = np.meshgrid(np.arange(Nf), np.arange(2), np.arange(nb), indexing="ij")
    perm = np.empty(2 * Nf * nb, dtype=np.int64)
    perm[(c_ * Nf * nb + j_ * Nf + f_).ravel()] = (f_ * 2 * nb + c_ * nb + j_).ravel()
    Sigma = var_e * Minv[np.ix_(perm, perm)]
    fva = 1.0 - var_e / var_y                                        # :255
    if fva < 0.9:
        import warnings
        warnings.warn(f"Fraction of variance explained = {fva}")    # :256
    return SpectralExt(Y, X, V, w, Nv, λ, coulomb, normalize, params, Sigma)


def windowpsd_sparse_batched(y, t, freqs, n, noverlap=-1, W=None, proxg=None, λ=1.0, μ=0.05, tol=1e-5, iters=10000,
                             win_lo=0, win_hi=None, device=0, as_written=True):
    """All windows ``[win_lo, win_hi)`` of ``Windows2(y,t,n,noverlap)`` solved together on one GPU with the weighted
    ``ls_sparse_spectral(y,t,f,W)`` estimator (src/lasso.jl:105-126): one batch of panels / Gram matrices /
    factorisations / ADMM iterations per launch.  Returns ``(x [nwin x Nf complex], S_partial [Nf], iters [nwin])``
    where ``S_partial = Σ_i |x_i|²`` in window order (not yet divided by k²)."""
    ky, py, Ly = as_f64(y)
    kt, pt, Lt = as_f64(t)
    kf, pf, Nf = as_f64(freqs)
    kw, pw, nW = as_f64(W)
    assert Ly == Lt, "y and t has to be the same length"
    assert W is None or nW == n, "W must have one weight per window sample"
    k = C.c_int64(0)
    check(lib().lpvs_window_count(Ly, int(n), int(noverlap), C.byref(k)))
    win_hi = int(k.value) if win_hi is None else int(win_hi)
    nwin = win_hi - int(win_lo)
    proxg = NormL1(λ) if proxg is None else proxg
    kind, param, glen = proxg.device_params(2 * Nf)
    xre, xim = np.zeros((max(nwin, 1), Nf)), np.zeros((max(nwin, 1), Nf))
    S, its = np.zeros(Nf), np.zeros(max(nwin, 1), dtype=np.int64)
    sign = _lib.LINEAR_QUADRATIC_AS_WRITTEN if as_written else _lib.LINEAR_LEAST_SQUARES
    check(lib().lpvs_windowpsd_sparse_f64(py, pt, Ly, int(n), int(noverlap), pw, pf, Nf, int(kind), float(param), int(glen),
                                          float(μ), float(tol), int(iters), int(sign), int(win_lo), win_hi, int(device),
                                          out_ptr(xre), out_ptr(xim), out_ptr(S), out_ptr(its)))
    return (xre + 1j * xim)[:nwin], S, its[:nwin]


def windowpsd_last_timing():
    """Phase times (HIP events) of this thread's last batched-window call."""
    o = np.zeros(12)
    check(lib().lpvs_windowpsd_last_timing(out_ptr(o), 12))
    d = dict(gram_rhs_ms=o[0], inverse_ms=o[1], solve_ms=o[2], windows=int(o[3]), passes=int(o[6]), structured_gram=bool(o[7]),
             gram_form=("dense", "ap", "ap-nufft")[int(o[7])], one_launch_iteration=bool(o[9]), reads_32_bits=int(o[9]) == 2,
             rccl_gather_ranks=int(o[10]), multi_devices=int(o[11]))
    if o[4] > 0:
        d["matvec_us_per_iteration"] = float(o[4]); d["matvec_windows"] = int(o[5]); d["matvec_bytes_per_launch"] = float(o[8])

    return d


def _engine_args(estimator, kwargs, nreg):
    """Map the reference's estimator + kwargs onto the engine's arguments, or None when that combination has no batched
    form (user-supplied estimator, per-iteration callback, unknown keywords -> the reference's sequential loop)."""
    kw = dict(kwargs)
    kw.pop("device", None)
    if estimator is ls_spectral:                                       # 4-argument weighted method, src/lsfft.jl:74-80
        if kw.pop("verbose", False) or set(kw) - {"λ"}:
            return None
        return dict(estimator=_lib.EST_DENSE, lam=float(kw.get("λ", 1e-10)), prox=(_lib.PROX_L1, 0.0, 0), μ=0.05, tol=0.0, iters=0,
                    sign=_lib.LINEAR_LEAST_SQUARES)
    if estimator is ls_sparse_spectral:                                # 4-argument weighted method, src/lasso.jl:105-126
        if kw.get("cb") is not None or set(kw) - {"λ", "proxg", "μ", "tol", "iters", "printerval", "cb", "init", "out"}:
            return None                                                # a per-iteration callback needs the host loop
        pg = kw.get("proxg")
        pg = NormL1(kw.get("λ", 1.0)) if pg is None else pg
        if not hasattr(pg, "device_params"):
            return None
        μ = kw.get("μ", 0.05)
        assert 0 <= μ <= 1, "μ should be ≤ 1"                          # src/lasso.jl:143
        init = bool(kw.get("init", False))                             # x0 = fourier_solve(A, y, zerofreq, λ): one batched ridge solve (:112)
        return dict(estimator=_lib.EST_SPARSE_INIT if init else _lib.EST_SPARSE, lam=float(kw.get("λ", 1.0)) if init else 0.0,
                    prox=pg.device_params(nreg), μ=float(μ), tol=float(kw.get("tol", 1e-5)),
                    iters=int(kw.get("iters", 10000)), sign=_lib.LINEAR_QUADRATIC_AS_WRITTEN)
    return None


def windows_estimate(Y, t, freqs, n, noverlap, W, eng, win_lo=0, win_hi=None, device=0):
    """The batched-window engine (``lpvs_windows_estimate_f64``): ``Y`` is a list of signals sharing ``t``; returns
    ``x[ns][nwin][Nf]`` complex and the iteration counts ``[ns][nwin]``."""
    # Float32 entry points only when y, t and freqs are ALL Float32: the window drivers call the 4-argument estimator, whose regressor
    # is evaluated in the eltype of t and freqs (src/lsfft.jl:121 -> src/lasso.jl:111); Float32 y with Float64 t stays exact in t
    if all(is_f32(y) and not _lib.is_device_array(y) for y in Y) and _all_f32(t, freqs):      # lpvs_windows_estimate_f32
        keep = np.ascontiguousarray(np.stack([np.asarray(y, dtype=np.float32) for y in Y]))
        ns, Ly = keep.shape
        th = np.ascontiguousarray(np.asarray(_host(t), dtype=np.float32)); fh = np.ascontiguousarray(np.asarray(_host(freqs), dtype=np.float32))
        Wh = None if W is None else np.ascontiguousarray(np.asarray(_host(W), dtype=np.float32))
        assert Ly == len(th), "y and t has to be the same length"
        assert Wh is None or len(Wh) == n, "W must have one weight per window sample"
        k = C.c_int64(0)
        check(lib().lpvs_window_count(Ly, int(n), int(noverlap), C.byref(k)))
        win_hi = int(k.value) if win_hi is None else int(win_hi)
        nwin, Nf = win_hi - int(win_lo), len(fh)
        xre, xim = np.zeros((ns, max(nwin, 1), Nf), dtype=np.float32), np.zeros((ns, max(nwin, 1), Nf), dtype=np.float32)
        its = np.zeros((ns, max(nwin, 1)), dtype=np.int64)
        kind, param, glen = eng["prox"]
        check(lib().lpvs_windows_estimate_f32(out_ptr(keep), ns, out_ptr(th), Ly, int(n), int(noverlap), None if Wh is None else out_ptr(Wh), out_ptr(fh), Nf,
                                              int(eng["estimator"]), float(eng["lam"]), int(kind), float(param), int(glen), float(eng["μ"]), float(eng["tol"]),
                                              int(eng["iters"]), int(eng["sign"]), int(win_lo), win_hi, int(device), out_ptr(xre), out_ptr(xim), out_ptr(its)))
        return (xre + 1j * xim).astype(np.complex64)[:, :nwin], its[:, :nwin]
    Ys = [as_f64(y) for y in Y]
    ns, Ly = len(Ys), Ys[0][2]
    assert all(e[2] == Ly for e in Ys), "signals must have the same length"
    kt, pt, Lt = as_f64(t)
    kf, pf, Nf = as_f64(freqs)
    kw, pw, nW = as_f64(W)
    assert Ly == Lt, "y and t has to be the same length"
    assert W is None or nW == n, "W must have one weight per window sample"
    k = C.c_int64(0)
    check(lib().lpvs_window_count(Ly, int(n), int(noverlap), C.byref(k)))
    win_hi = int(k.value) if win_hi is None else int(win_hi)
    nwin = win_hi - int(win_lo)
    dev_in = all(_lib.is_device_array(y) for y in Y)
    if ns == 1:
        keep, pY = Ys[0][0], Ys[0][1]
    elif dev_in:
        import torch
        keep = torch.stack([y.reshape(-1) for y in Y]).contiguous()   # [ns][L] row-major == L x ns column-major
        pY = C.c_void_p(keep.data_ptr())
    else:
        keep = np.ascontiguousarray(np.stack([_host(y) for y in Y]))
        pY = out_ptr(keep)
    xre, xim = np.zeros((ns, max(nwin, 1), Nf)), np.zeros((ns, max(nwin, 1), Nf))
    its = np.zeros((ns, max(nwin, 1)), dtype=np.int64)
    kind, param, glen = eng["prox"]
    check(lib().lpvs_windows_estimate_f64(pY, ns, pt, Ly, int(n), int(noverlap), pw, pf, Nf, int(eng["estimator"]), float(eng["lam"]),
                                          int(kind), float(param), int(glen), float(eng["μ"]), float(eng["tol"]), int(eng["iters"]),
                                          int(eng["sign"]), int(win_lo), win_hi, int(device), out_ptr(xre), out_ptr(xim), out_ptr(its)))
    return (xre + 1j * xim)[:, :nwin], its[:, :nwin]


def windows_estimate_state(Y, t, freqs, n, noverlap, W, eng, win_lo=0, win_hi=None, device=0):
    """``lpvs_windows_estimate_state_f64``: the engine of :func:`windows_estimate`, returning the raw ADMM state ``x, z, u`` of every
    problem (each ``[ns][nwin][Nreg]`` in the reference's ordering ``[re; im]``: the ``(x, z)`` of src/lasso.jl:170 and the dual variable) and
    the iteration counts ``[ns][nwin]``.  Sparse estimators, Float64."""
    Ys = [as_f64(y) for y in Y]
    ns, Ly = len(Ys), Ys[0][2]
    assert all(e[2] == Ly for e in Ys), "signals must have the same length"
    kt, pt, Lt = as_f64(t)
    kf, pf, Nf = as_f64(freqs)
    kw, pw, nW = as_f64(W)
    assert Ly == Lt, "y and t has to be the same length"
    assert W is None or nW == n, "W must have one weight per window sample"
    k = C.c_int64(0)
    check(lib().lpvs_window_count(Ly, int(n), int(noverlap), C.byref(k)))
    win_hi = int(k.value) if win_hi is None else int(win_hi)
    nwin = win_hi - int(win_lo)
    if ns == 1:
        keep, pY = Ys[0][0], Ys[0][1]
    elif all(_lib.is_device_array(y) for y in Y):
        import torch
        keep = torch.stack([y.reshape(-1) for y in Y]).contiguous()
        pY = C.c_void_p(keep.data_ptr())
    else:
        keep = np.ascontiguousarray(np.stack([_host(y) for y in Y]))
        pY = out_ptr(keep)
    zf = C.c_int64(0)
    fh = np.ascontiguousarray(np.asarray(_host(freqs), dtype=np.float64))
    check(lib().lpvs_check_freq_f64(out_ptr(fh), Nf, C.byref(zf)))
    nreg = 2 * Nf - 1 if zf.value else 2 * Nf
    x, z, u = (np.zeros((ns, max(nwin, 1), nreg)) for _ in range(3))
    its = np.zeros((ns, max(nwin, 1)), dtype=np.int64)
    kind, param, glen = eng["prox"]
    check(lib().lpvs_windows_estimate_state_f64(pY, ns, pt, Ly, int(n), int(noverlap), pw, pf, Nf, int(eng["estimator"]), float(eng["lam"]),
                                                int(kind), float(param), int(glen), float(eng["μ"]), float(eng["tol"]), int(eng["iters"]),
                                                int(eng["sign"]), int(win_lo), win_hi, int(device), out_ptr(x), out_ptr(z), out_ptr(u), out_ptr(its)))
    return x[:, :nwin], z[:, :nwin], u[:, :nwin], its[:, :nwin]


def windows_estimate_multi(Y, t, freqs, n, noverlap, W, eng, ngpus=0, devices=None):
    """``lpvs_windows_estimate_multi_f64``: ALL windows, split into contiguous ranges over ``ngpus`` devices driven by
    this one process (a host thread per device), the coefficients gathered by one RCCL all-gather.  Same return value as
    :func:`windows_estimate` over the full window range."""
    f32 = all(is_f32(y) and not _lib.is_device_array(y) for y in Y) and _all_f32(t, freqs)      # all-Float32 call: lpvs_windows_estimate_multi_f32
    dt = np.float32 if f32 else np.float64
    Ys = [np.ascontiguousarray(np.asarray(_host(y), dtype=dt)) for y in Y]
    ns, Ly = len(Ys), len(Ys[0])
    assert all(len(e) == Ly for e in Ys), "signals must have the same length"
    keep = np.ascontiguousarray(np.stack(Ys))
    th = np.ascontiguousarray(np.asarray(_host(t), dtype=dt)); fh = np.ascontiguousarray(np.asarray(_host(freqs), dtype=dt))
    Wh = None if W is None else np.ascontiguousarray(np.asarray(_host(W), dtype=dt))
    assert Ly == len(th), "y and t has to be the same length"
    assert Wh is None or len(Wh) == n, "W must have one weight per window sample"
    k = C.c_int64(0)
    check(lib().lpvs_window_count(Ly, int(n), int(noverlap), C.byref(k)))
    k, Nf = int(k.value), len(fh)
    dv = None if devices is None else np.ascontiguousarray(np.asarray(devices, dtype=np.int32))
    if dv is not None:
        ngpus = len(dv)
    xre, xim = np.zeros((ns, max(k, 1), Nf), dtype=dt), np.zeros((ns, max(k, 1), Nf), dtype=dt)
    its = np.zeros((ns, max(k, 1)), dtype=np.int64)
    kind, param, glen = eng["prox"]
    check((lib().lpvs_windows_estimate_multi_f32 if f32 else lib().lpvs_windows_estimate_multi_f64)(out_ptr(keep), ns, out_ptr(th), Ly, int(n), int(noverlap), None if Wh is None else out_ptr(Wh),
                                                out_ptr(fh), Nf, int(eng["estimator"]), float(eng["lam"]), int(kind), float(param), int(glen),
                                                float(eng["μ"]), float(eng["tol"]), int(eng["iters"]), int(eng["sign"]),
                                                None if dv is None else out_ptr(dv), int(ngpus), out_ptr(xre), out_ptr(xim), out_ptr(its)))
    return (xre + 1j * xim)[:, :k], its[:, :k]


def windowcsd_batched(y, u, t, freqs, n, noverlap, W, eng, win_lo=0, win_hi=None, device=0):
    """``lpvs_windowcsd_f64``: the accumulators ``(Syu, Syy, Suu)`` over the windows ``[win_lo, win_hi)`` in window order
    (one Gram / factorisation per window, two right-hand sides), plus the per-window ``xy, xu``."""
    f32 = (is_f32(y) and is_f32(u) and not _lib.is_device_array(y) and not _lib.is_device_array(u)
           and _all_f32(t, freqs))                             # all-Float32 call: lpvs_windowcsd_f32 (see windows_estimate)
    dt = np.float32 if f32 else np.float64
    conv = (lambda a: as_f32(None if a is None else np.asarray(_host(a), dtype=np.float32))) if f32 else as_f64
    ky, py, Ly = conv(y)
    ku, pu, Lu = conv(u)
    kt, pt, Lt = conv(t)
    kf, pf, Nf = conv(freqs)
    kw, pw, nW = conv(W)
    assert Ly == Lu == Lt, "y, u and t has to be the same length"
    assert W is None or nW == n, "W must have one weight per window sample"
    k = C.c_int64(0)
    check(lib().lpvs_window_count(Ly, int(n), int(noverlap), C.byref(k)))
    win_hi = int(k.value) if win_hi is None else int(win_hi)
    nwin = win_hi - int(win_lo)
    sre, sim, syy, suu = (np.zeros(Nf, dtype=dt) for _ in range(4))
    xre, xim = np.zeros((2, max(nwin, 1), Nf), dtype=dt), np.zeros((2, max(nwin, 1), Nf), dtype=dt)
    kind, param, glen = eng["prox"]
    check((lib().lpvs_windowcsd_f32 if f32 else lib().lpvs_windowcsd_f64)(py, pu, pt, Ly, int(n), int(noverlap), pw, pf, Nf, int(eng["estimator"]), float(eng["lam"]), int(kind),
                                   float(param), int(glen), float(eng["μ"]), float(eng["tol"]), int(eng["iters"]), int(eng["sign"]),
                                   int(win_lo), win_hi, int(device), out_ptr(sre), out_ptr(sim), out_ptr(syy), out_ptr(suu), out_ptr(xre),
                                   out_ptr(xim), None))
    x = (xre + 1j * xim)[:, :nwin]
    return sre + 1j * sim, syy, suu, x[0], x[1]


@_with_options
def ls_windowpsd(y, t, freqs=None, nw=8, noverlap=-1, window_func=rect, estimator=None, batched=True, **kwargs):
    """``ls_windowpsd(y,t,freqs; nw, noverlap, window_func, estimator=ls_spectral, kwargs...)``
    (src/lsfft.jl:112-126) -> ``(S, freqs)``.  ``estimator`` is any callable ``(y,t,f,W; kw...) -> (x, f)``
    (plugin boundary #1); it is always called with the window vector, as in the reference (:121).

    With this package's ``ls_spectral`` / ``ls_sparse_spectral`` as the estimator (and no per-iteration callback) the windows
    are solved as ONE device batch (per-iteration progress lines are not printed); ``batched=False`` forces the reference's
    sequential loop."""
    estimator = ls_spectral if estimator is None else estimator
    n = len(y) // nw                                                # :113
    if freqs is None:
        freqs = default_freqs(t, n=n)                               # :114
    windows = Windows2(y, t, n, noverlap, window_func)              # :115
    k = len(windows)                                                # :116
    ngpus = kwargs.pop("ngpus", 1)                                  # extension: devices driven by this process (0 = all visible)
    eng = _engine_args(estimator, kwargs, 2 * len(freqs)) if (batched and k > 0) else None
    if eng is not None:
        if ngpus != 1:
            x, _ = windows_estimate_multi([y], t, freqs, n, windows.noverlap, windows.W, eng, ngpus=ngpus)
        else:
            x, _ = windows_estimate([y], t, freqs, n, windows.noverlap, windows.W, eng, device=kwargs.get("device", 0))
        S = np.zeros(len(freqs))
        for i in range(k):
            S += abs2(x[0, i])                                      # :122, window order
        return S / k ** 2, freqs                                    # :125
    S = np.zeros(len(freqs))
    for yi, ti in windows:                                          # :120
        x = estimator(yi, ti, freqs, windows.W, **kwargs)[0]        # :121
        S += abs2(x)                                                # :122
    return S / k ** 2, freqs                                        # :125


@_with_options
def ls_windowcsd(y, u, t, freqs=None, nw=10, noverlap=-1, window_func=rect, estimator=None, batched=True, **kwargs):
    """``ls_windowcsd(y,u,t,freqs; nw, noverlap, window_func, estimator=ls_spectral)`` (src/lsfft.jl:140-156):
    cross spectral density, ``S += xy .* conj(xu)`` over the windows, returned as ``S/nw`` (the recomputed window count).

    Batched on the device engine as :func:`ls_windowpsd` is: one Gram and one factorisation per window serve both signals."""
    estimator = ls_spectral if estimator is None else estimator
    n = len(y) // nw
    if freqs is None:
        freqs = default_freqs(t, n=n)
    wy = Windows2(y, t, n, noverlap, window_func)
    wu = Windows2(u, t, n, noverlap, window_func)
    k = len(wy)
    ngpus = kwargs.pop("ngpus", 1)
    eng = _engine_args(estimator, kwargs, 2 * len(freqs)) if (batched and k > 0) else None
    if eng is not None and ngpus != 1:
        x, _ = windows_estimate_multi([y, u], t, freqs, n, wy.noverlap, wy.W, eng, ngpus=ngpus)
        S = np.zeros(len(freqs), dtype=np.complex128)
        for i in range(k):
            S = S + _mul_conj(x[0, i], x[1, i])                     # :152, window order
        return S / k, freqs
    if eng is not None:
        Syu, _, _, _, _ = windowcsd_batched(y, u, t, freqs, n, wy.noverlap, wy.W, eng, device=kwargs.get("device", 0))
        return Syu / k, freqs
    S = np.zeros(len(freqs), dtype=np.complex128)
    for (yi, ti), (ui, _) in zip(wy, wu):
        xy = estimator(yi, ti, freqs, wy.W, **kwargs)[0]
        xu = estimator(ui, ti, freqs, wu.W, **kwargs)[0]
        S = S + _mul_conj(xy, xu)                                 # src/lsfft.jl:152
    return S / k, freqs


@_with_options
def ls_cohere(y, u, t, freqs=None, nw=10, noverlap=-1, estimator=None, batched=True, **kwargs):
    """``ls_cohere(y,u,t,freqs; nw, noverlap, estimator=ls_spectral)`` (src/lsfft.jl:176-193): magnitude-squared
    coherence over Hann-weighted windows (``Windows3(y,t,u,n,noverlap,hanning)``, :182)."""
    from .windows import hanning
    estimator = ls_spectral if estimator is None else estimator
    n = len(y) // nw
    if freqs is None:
        freqs = default_freqs(t, n=n)
    windows = Windows3(y, t, u, n, noverlap, hanning)
    ngpus = kwargs.pop("ngpus", 1)
    eng = _engine_args(estimator, kwargs, 2 * len(freqs)) if (batched and len(windows) > 0) else None
    if eng is not None and ngpus != 1:
        x, _ = windows_estimate_multi([y, u], t, freqs, n, windows.noverlap, windows.W, eng, ngpus=ngpus)
        Syy, Suu = np.zeros(len(freqs)), np.zeros(len(freqs))
        Syu = np.zeros(len(freqs), dtype=np.complex128)
        for i in range(x.shape[1]):                                 # :183-190, window order
            Syu += _mul_conj(x[0, i], x[1, i]); Syy += abs2(x[0, i]); Suu += abs2(x[1, i])
        with np.errstate(invalid="ignore", divide="ignore"):        # 0/0 where neither signal has the frequency: NaN, silently, as in Julia
            return abs2(Syu) / (Suu * Syy), freqs
    if eng is not None:
        Syu, Syy, Suu, _, _ = windowcsd_batched(y, u, t, freqs, n, windows.noverlap, windows.W, eng, device=kwargs.get("device", 0))
        with np.errstate(invalid="ignore", divide="ignore"):
            return abs2(Syu) / (Suu * Syy), freqs                   # :191
    Syy, Suu = np.zeros(len(freqs)), np.zeros(len(freqs))
    Syu = np.zeros(len(freqs), dtype=np.complex128)
    for yi, ti, ui in windows:
        xy = np.asarray(estimator(yi, ti, freqs, windows.W, **kwargs)[0])
        xu = np.asarray(estimator(ui, ti, freqs, windows.W, **kwargs)[0])
        Syu += _mul_conj(xy, xu)
        Syy += abs2(xy)
        Suu += abs2(xu)
    with np.errstate(invalid="ignore", divide="ignore"):
        return abs2(Syu) / (Suu * Syy), freqs


@_with_options
def ls_windowpsd_lpv(Y, X, V, w, Nv, nw=10, noverlap=0, in_flight=4, **kwargs):
    """src/lsfft.jl:267-277.  The windows' regressors share nothing (each has its own samples of X and V): their Grams are built
    ``in_flight`` at a time (an extension; 1 = one after the other) into one batch, whose factorisations and refined ridge solves
    then run for all windows at once (``lpvs_windowpsd_lpv_f64``; 40 windows of 5000 samples with 1024 unknowns each: 19 ms with four
    Gram builds in flight, 22 with two, 35 with one -- the per-window loop of single-handle solves: 129 ms).  The sum over windows is taken in window order (:274), so the
    result does not depend on ``in_flight``; windows explaining less than 0.9 of their variance warn as the reference does (:255-256)."""
    w = np.ravel(_host(w))
    S = np.zeros(len(w))
    # the library's own driver (lpvs_windowpsd_lpv_f64: the same device solves, the library's worker threads) whenever the call has only
    # what it takes; a singular window (LPVS_ENUMERIC) sends the whole call down the per-window path below, which has the host-QR route
    if set(kwargs) <= {"λ", "coulomb", "normalize", "device"} and not any(_lib.is_device_array(a) for a in (Y, X, V)) and not _all_f32(Y, X, V):
        Yh, Xh, Vh = (np.ascontiguousarray(_host(a), dtype=np.float64) for a in (Y, X, V))
        assert len(Yh) == len(Xh) == len(Vh), "y, t and v has to be the same length"   # src/windows.jl:96
        wv = np.ascontiguousarray(w, dtype=np.float64)
        kcnt = C.c_int64(0)
        check(lib().lpvs_window_count(len(Yh), len(Yh) // int(nw), int(noverlap), C.byref(kcnt)))
        fva = np.ones(max(int(kcnt.value), 1))
        try:
            check(lib().lpvs_windowpsd_lpv_f64(out_ptr(Yh), out_ptr(Xh), out_ptr(Vh), len(Yh), out_ptr(wv), len(wv), int(Nv), len(Yh) // int(nw), int(noverlap),
                                               float(kwargs.get("λ", 1e-8)), int(bool(kwargs.get("normalize", True))), int(bool(kwargs.get("coulomb", False))),
                                               int(kwargs.get("device", 0)), max(1, min(8, int(in_flight))), out_ptr(S), out_ptr(fva)))
            for v in fva[:int(kcnt.value)]:
                if v < 0.9:                                            # src/lsfft.jl:255-256, once per window as the reference's loop does
                    import warnings
                    warnings.warn(f"Fraction of variance explained = {v}")
            return S
        except _lib.NumericError as e:
            log.info("ls_windowpsd_lpv: %s; per-window path", e)
            S = np.zeros(len(w))
    windows = list(Windows3(Y, X, V, len(Y) // nw, noverlap, rect))
    kwargs.setdefault("covariance", False)                         # the driver reads the parameters only (:273-274): no Σ, no second inverse
    solve = lambda win: ls_spectral_lpv(win[0], win[1], win[2], w, Nv, **kwargs)
    if in_flight > 1 and len(windows) > 1:
        from concurrent.futures import ThreadPoolExecutor
        opts = {k: get_default_option(k) for k in _lib.OPTIONS}    # (the option defaults are thread-local: carried into the workers)
        def worker(win):
            with default_options(**opts):
                return solve(win)
        with ThreadPoolExecutor(max_workers=int(in_flight)) as pool:
            ses = list(pool.map(worker, windows))                  # (ctypes releases the GIL inside the library; results in window order)
    else:
        ses = [solve(win) for win in windows]
    for se in ses:
        rp = reshape_params(se.x, len(w))
        S = S + abs2(rp.sum(axis=1))
    return S
