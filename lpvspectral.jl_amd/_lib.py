"""ctypes binding of ``liblpvspectral.so`` (the C-ABI declared in ``include/lpvspectral.h``).

The library is the product: there is no CPU fallback.  If the shared object is missing the
import fails loudly; if no HIP device is visible every compute entry point returns
``LPVS_EDEVICE`` and the wrappers raise :class:`DeviceError`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LPVS_LIBRARY: measurement tools load an instrumented build of the SAME sources, e.g. liblpvspectral_timeline.so -- never a fallback)
LIB_PATH = os.environ.get("LPVS_LIBRARY") or os.path.join(_HERE, "liblpvspectral.so")

LPVS_OK = 0
LPVS_EARGUMENT, LPVS_EASSERT, LPVS_EDOMAIN, LPVS_ENOMEM = -1, -2, -3, -4
LPVS_EDEVICE, LPVS_EUNSUPPORTED, LPVS_ENUMERIC, LPVS_ESTATE = -5, -6, -7, -8

PROX_L1, PROX_L0, PROX_BALL_L0, PROX_GROUP_L2 = 1, 2, 3, 4
LINEAR_LEAST_SQUARES, LINEAR_QUADRATIC_AS_WRITTEN = 1, -1
EST_SPARSE, EST_DENSE, EST_SPARSE_INIT = 1, 2, 3
# options (include/lpvspectral.h LPVS_OPT_*): name -> (option id, {value name -> value}); None / "default" = 0
OPTIONS = {
    "storage": (1, {"mixed": 1, "split": 2, "f64": 3, "mixed32": 4}),
    "iteration": (2, {"one": 1, "two": 2}),
    "gram_form": (3, {"ap": 1, "krs": 2, "kr": 3}),
    "nt_loads": (4, {"off": 1, "on": 2}),
    "slot_sums": (5, {"nufft": 1, "direct": 2}),
    # integer-valued: window_chunk_mb = MB of packed inverses per chunk of the batched-window engine ("uncut": one chunk),
    # windows_in_flight = 1 .. 4 parts of a chunk at a time, reserve_cus = CUs left to the factorisation's pivot chain ("none")
    "window_chunk_mb": (6, {"uncut": -1}),
    "windows_in_flight": (7, {}),
    "reserve_cus": (8, {"none": -1}),
    # the x-update correction (handles of n >= 2048; default: on for one right-hand side, off for several)
    "xupdate_correction": (9, {"on": 1, "off": 2}),
}
_INT_OPTIONS = {"window_chunk_mb", "windows_in_flight", "reserve_cus"}


def option_ids(name, value):
    """(option id, value id) of an option given by name (``storage="f64"``); ``None`` / ``"default"`` -> 0."""
    if name not in OPTIONS:
        raise KeyError(f"unknown option {name!r} (known: {sorted(OPTIONS)})")
    oid, vals = OPTIONS[name]
    if value is None or value == "default":
        return oid, 0
    if name in _INT_OPTIONS and not isinstance(value, (str, bool)):
        return oid, int(value)
    if isinstance(value, bool):
        value = "on" if value else "off"
    if value not in vals:
        raise ValueError(f"option {name}: unknown value {value!r} (known: {sorted(vals)})")
    return oid, vals[value]


class DeviceError(RuntimeError):
    """HIP runtime failure or no gfx950 device (LPVS_EDEVICE)."""


class DomainError(ValueError):
    """Julia's DomainError (DSP.arraysplit: noverlap >= n)."""


class NumericError(ArithmeticError):
    """(G + I/mu) not positive definite (LPVS_ENUMERIC)."""


# every exported symbol of include/lpvspectral.h with its signature
_I32, _I64, _F64, _P = C.c_int32, C.c_int64, C.c_double, C.c_void_p
_PI64 = C.POINTER(C.c_int64)
SIGNATURES = {
    "lpvs_version": (_I32, []),
    "lpvs_device_count": (_I32, []),
    "lpvs_last_error": (C.c_char_p, []),
    "lpvs_release_cached_memory": (_I32, []),
    "lpvs_set_default_option": (_I32, [_I32, _I32]),
    "lpvs_get_default_option": (_I32, [_I32, C.POINTER(_I32)]),
    "lpvs_problem_set_option": (_I32, [_P, _I32, _I32]),
    "lpvs_problem_get_option": (_I32, [_P, _I32, C.POINTER(_I32)]),
    "lpvs_check_freq_f64": (_I32, [_P, _I64, _PI64]),
    "lpvs_fourier_regressor_f64": (_I32, [_P, _I64, _P, _I64, _P, _PI64]),
    "lpvs_basis_activation_f64": (_I32, [_P, _I64, _I64, _I32, _I32, _P]),
    "lpvs_lpv_regressor_f64": (_I32, [_P, _P, _I64, _P, _I64, _I64, _I32, _I32, _I32, _P]),
    "lpvs_problem_create_fourier_f64": (_I32, [_P, _P, _I64, _P, _I64, _P, _I32, C.POINTER(_P)]),
    "lpvs_problem_create_lpv_f64": (_I32, [_P, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _I32, C.POINTER(_P)]),
    "lpvs_problem_create_lpv_multi_f64": (_I32, [_P, _I64, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _I32, C.POINTER(_P)]),
    "lpvs_problem_num_signals": (_I32, [_P, _PI64]),
    "lpvs_admm_status": (_I32, [_P, _I64, _PI64, C.POINTER(_F64), C.POINTER(_I32)]),
    "lpvs_problem_create_dense_f64": (_I32, [_P, _P, _I64, _I64, _P, _I32, C.POINTER(_P)]),
    "lpvs_problem_create_gram_f64": (_I32, [_P, _P, _I64, _I32, C.POINTER(_P)]),
    "lpvs_problem_destroy": (_I32, [_P]),
    "lpvs_problem_size": (_I32, [_P, _PI64]),
    "lpvs_problem_zerofreq": (_I32, [_P, _PI64]),
    "lpvs_problem_get_gram_f64": (_I32, [_P, _P, _P]),
    "lpvs_problem_get_rhs_f64": (_I32, [_P, _P]),
    "lpvs_problem_get_inverse_f64": (_I32, [_P, _F64, _P]),
    "lpvs_problem_solve_ridge_f64": (_I32, [_P, _F64, _P]),
    "lpvs_ls_spectral_f64": (_I32, [_P, _P, _I64, _P, _I64, _F64, _I32, _P, _P]),
    "lpvs_problem_set_prox": (_I32, [_P, _I32, _F64, _I64]),
    "lpvs_admm_init_f64": (_I32, [_P, _P, _F64, _F64, _I32]),
    "lpvs_admm_run": (_I32, [_P, _I64, _PI64, C.POINTER(_F64), C.POINTER(_I32)]),
    "lpvs_admm_set_state_f64": (_I32, [_P, _P, _P, _P, _I64]),
    "lpvs_admm_offset_len": (_I32, [_P, _PI64]),
    "lpvs_admm_get_offset_f64": (_I32, [_P, _P, _I64]),
    "lpvs_admm_set_offset_f64": (_I32, [_P, _P, _I64]),
    "lpvs_admm_matvec_kind": (_I32, [_P, C.POINTER(_I32)]),
    "lpvs_windowpsd_last_timing": (_I32, [_P, _I32]),
    "lpvs_windows_estimate_f64": (_I32, [_P, _I64, _P, _I64, _I64, _I64, _P, _P, _I64, _I32, _F64, _I32, _F64, _I64, _F64, _F64, _I64, _I32,
                                          _I64, _I64, _I32, _P, _P, _P]),
    "lpvs_windows_estimate_state_f64": (_I32, [_P, _I64, _P, _I64, _I64, _I64, _P, _P, _I64, _I32, _F64, _I32, _F64, _I64, _F64, _F64, _I64, _I32,
                                                _I64, _I64, _I32, _P, _P, _P, _P]),
    "lpvs_windows_estimate_multi_f64": (_I32, [_P, _I64, _P, _I64, _I64, _I64, _P, _P, _I64, _I32, _F64, _I32, _F64, _I64, _F64, _F64, _I64, _I32,
                                                _P, _I32, _P, _P, _P]),
    "lpvs_lpv_batch_multi_f64": (_I32, [_P, _I64, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _F64, _I64, _F64, _F64, _I64, _P, _I32, _P, _P, _P]),
    "lpvs_windowpsd_lpv_f64": (_I32, [_P, _P, _P, _I64, _P, _I64, _I64, _I64, _I64, _F64, _I32, _I32, _I32, _I32, _P, _P]),
    "lpvs_lpv_signals_multi_f64": (_I32, [_P, _P, _P, _I64, _I64, _P, _I64, _I64, _I32, _I32, _F64, _I64, _F64, _F64, _I64, _P, _I32, _I32, _P, _P, _P]),
    "lpvs_windowcsd_f64": (_I32, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _I64, _I32, _F64, _I32, _F64, _I64, _F64, _F64, _I64, _I32,
                                   _I64, _I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "lpvs_lpv_ranges_f64": (_I32, [_P, _P, _I64, _P]),
    "lpvs_problem_create_lpv_rows_f64": (_I32, [_P, _I64, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _P, _I32, C.POINTER(_P)]),
    "lpvs_problem_device_gram_f64": (_I32, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    "lpvs_problem_gram_modified": (_I32, [_P]),
    "lpvs_check_freq_f32": (_I32, [_P, _I64, C.POINTER(_I64)]),
    "lpvs_fourier_regressor_f32": (_I32, [_P, _I64, _P, _I64, _P, C.POINTER(_I64)]),
    "lpvs_lpv_regressor_f32": (_I32, [_P, _P, _I64, _P, _I64, _I64, _I32, _I32, _I32, _P]),
    "lpvs_problem_create_fourier_f32": (_I32, [_P, _P, _I64, _P, _I64, _P, _I32, C.POINTER(_P)]),
    "lpvs_problem_create_lpv_f32": (_I32, [_P, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _I32, C.POINTER(_P)]),
    "lpvs_problem_create_lpv_multi_f32": (_I32, [_P, _I64, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _I32, C.POINTER(_P)]),
    "lpvs_windows_estimate_f32": (_I32, [_P, _I64, _P, _I64, _I64, _I64, _P, _P, _I64, _I32, _F64, _I32, _F64, _I64, _F64, _F64, _I64, _I32,
                                          _I64, _I64, _I32, _P, _P, _P]),
    "lpvs_windows_estimate_multi_f32": (_I32, [_P, _I64, _P, _I64, _I64, _I64, _P, _P, _I64, _I32, _F64, _I32, _F64, _I64, _F64, _F64, _I64, _I32,
                                                _P, _I32, _P, _P, _P]),
    "lpvs_windowcsd_f32": (_I32, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _I64, _I32, _F64, _I32, _F64, _I64, _F64, _F64, _I64, _I32,
                                   _I64, _I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "lpvs_problem_create_lpv_rows_f32": (_I32, [_P, _I64, _P, _P, _I64, _P, _I64, _I64, _I32, _I32, _P, _I32, C.POINTER(_P)]),
    "lpvs_problem_solve_ridge_f32": (_I32, [_P, _F64, _P]),
    "lpvs_admm_set_state_f32": (_I32, [_P, _P, _P, _P, _I64]),
    "lpvs_admm_init_f32": (_I32, [_P, _P, _F64, _F64, _I32]),
    "lpvs_admm_get_f32": (_I32, [_P, _P, _P, _P]),
    "lpvs_problem_get_params_f32": (_I32, [_P, _I32, _P, _P]),
    "lpvs_ls_spectral_f32": (_I32, [_P, _P, _I64, _P, _I64, _F64, _I32, _P, _P]),
    "lpvs_admm_time_matvec": (_I32, [_P, _I32, C.POINTER(_F64), C.POINTER(_F64)]),
    "lpvs_admm_get_f64": (_I32, [_P, _P, _P, _P]),
    "lpvs_problem_get_params_f64": (_I32, [_P, _I32, _P, _P]),
    "lpvs_problem_pack_params_f64": (_I32, [_P, _P, _P, _P]),
    "lpvs_problem_get_timing": (_I32, [_P, _P, _I32]),
    "lpvs_windowpsd_sparse_f64": (_I32, [_P, _P, _I64, _I64, _I64, _P, _P, _I64, _I32, _F64, _I64, _F64, _F64, _I64, _I32, _I64, _I64,
                                         _I32, _P, _P, _P, _P]),
    "lpvs_window_count": (_I32, [_I64, _I64, _I64, _PI64]),
    "lpvs_window_offsets": (_I32, [_I64, _I64, _I64, _P, _I64, _PI64]),
    "lpvs_merge_f64": (_I32, [_P, _I64, _I64, _I64, _I64, _P]),
}

_lib = None


def lib():
    """Load the shared object (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C lpvspectral.jl_amd/csrc). "
                "There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  If torch is going
        # to be used in this process (tests, bench: device tensors, torch.distributed) it must be loaded
        # first, so that this library binds to the same runtime through the matching SONAME; two copies
        # of the runtime cannot both own the GPU.  Without torch the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError here == header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def last_error() -> str:
    return lib().lpvs_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map a status code to the exception type the reference raises."""
    if rc == LPVS_OK:
        return
    msg = last_error()
    if rc == LPVS_EARGUMENT:
        raise ValueError(msg)            # ArgumentError, src/lsfft.jl:22
    if rc == LPVS_EASSERT:
        raise AssertionError(msg)        # @assert, src/lasso.jl:143
    if rc == LPVS_EDOMAIN:
        raise DomainError(msg)           # DSP.arraysplit
    if rc == LPVS_ENOMEM:
        raise MemoryError(msg)
    if rc == LPVS_EDEVICE:
        raise DeviceError(msg)
    if rc == LPVS_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == LPVS_ENUMERIC:
        raise NumericError(msg)
    raise RuntimeError(f"lpvspectral error {rc}: {msg}")


def is_device_array(a) -> bool:
    return hasattr(a, "data_ptr") and hasattr(a, "is_cuda") and bool(a.is_cuda)


def as_f64(a):
    """Return (keepalive, pointer, length) of a contiguous float64 vector/matrix.

    numpy arrays / sequences stay on the host (column-major for 2-D); torch CUDA tensors are
    passed as device pointers (the library detects the address space)."""
    if a is None:
        return None, None, 0
    if is_device_array(a):
        import torch
        if a.dtype != torch.float64:
            raise TypeError("device arrays must be float64 (the reference's eltype)")
        t = a if a.is_contiguous() else a.contiguous()
        # the library works on its own streams: whatever torch has queued to produce this tensor must be complete
        torch.cuda.current_stream(t.device).synchronize()
        return t, C.c_void_p(t.data_ptr()), t.numel()
    if hasattr(a, "detach") and hasattr(a, "numpy"):  # CPU torch tensor
        a = a.detach().numpy()
    arr = np.asarray(a, dtype=np.float64)
    arr = np.asfortranarray(arr) if arr.ndim == 2 else np.ascontiguousarray(arr)
    return arr, C.c_void_p(arr.ctypes.data), arr.size


def is_f32(a) -> bool:
    """True for float32 numpy arrays / torch tensors: such inputs take the _f32 entry points (the reference is
    eltype-generic, src/lasso.jl:85,91,144)."""
    if a is None:
        return False
    if hasattr(a, "data_ptr") and hasattr(a, "dtype"):
        import torch
        return a.dtype == torch.float32
    return isinstance(a, np.ndarray) and a.dtype == np.float32


def as_f32(a):
    """(keepalive, pointer, length) of a contiguous float32 vector/matrix (host numpy or device torch)."""
    if a is None:
        return None, None, 0
    if is_device_array(a):
        import torch
        t = a.to(torch.float32)
        t = t if t.is_contiguous() else t.contiguous()
        torch.cuda.current_stream(t.device).synchronize()
        return t, C.c_void_p(t.data_ptr()), t.numel()
    if hasattr(a, "detach") and hasattr(a, "numpy"):
        a = a.detach().numpy()
    arr = np.asarray(a, dtype=np.float32)
    arr = np.asfortranarray(arr) if arr.ndim == 2 else np.ascontiguousarray(arr)
    return arr, C.c_void_p(arr.ctypes.data), arr.size


def out_ptr(arr: np.ndarray):
    return C.c_void_p(arr.ctypes.data)
