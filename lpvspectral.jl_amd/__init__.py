"""MI355X-native regression-matrix + ADMM hot path of LPVSpectral.jl.

The product is ``liblpvspectral.so`` (hand-written HIP for gfx950 behind the C-ABI of
``include/lpvspectral.h``); this package is the host-side mirror of the reference's Julia API
(same names / keywords) used by the parity tests and the benchmark.  No CPU fallback exists:
importing without the built library raises ImportError, calling without a GPU raises DeviceError.
"""
from ._lib import DeviceError, DomainError, NumericError, lib as _load

_load()  # fail loudly at import time if the HIP extension is missing

from .api import (ADMM, Problem, SpectralExt, basis_activation_func, check_freq, default_freqs,  # noqa: E402
                  fourier2complex, get_fourier_regressor, lpv_regressor, ls_sparse_spectral,
                  ls_cohere, ls_sparse_spectral_lpv, ls_sparse_spectral_lpv_multi, ls_sparse_spectral_lpv_rowsharded, lpv_ranges, lpv_batch_multi, lpv_signals_multi, ls_spectral, ls_spectral_lpv, tls_spectral, ls_windowcsd, ls_windowpsd, ls_windowpsd_lpv, psd,
                  reshape_params, set_default_option, get_default_option, default_options, windowpsd_sparse_batched, windowpsd_last_timing, windows_estimate, windows_estimate_multi, windowcsd_batched)
from .prox import IndBallL0, LeastSquares, NormL0, NormL1, NormL2, Quadratic, SlicedSeparableSum  # noqa: E402
from . import sharding  # noqa: E402
from .windows import Windows2, Windows3, hanning, mapwindows, merge, rect  # noqa: E402

__all__ = [
    "ADMM", "Problem", "SpectralExt", "basis_activation_func", "check_freq", "default_freqs", "fourier2complex",
    "get_fourier_regressor", "lpv_regressor", "ls_sparse_spectral", "ls_sparse_spectral_lpv", "ls_sparse_spectral_lpv_multi", "ls_sparse_spectral_lpv_rowsharded", "lpv_ranges", "lpv_batch_multi", "lpv_signals_multi", "ls_spectral",
    "ls_spectral_lpv", "tls_spectral", "ls_windowcsd", "ls_cohere", "ls_windowpsd", "ls_windowpsd_lpv", "psd", "reshape_params", "IndBallL0", "LeastSquares",
    "NormL0", "NormL1", "NormL2", "Quadratic", "SlicedSeparableSum", "Windows2", "Windows3", "hanning",
    "mapwindows", "merge", "rect", "set_default_option", "get_default_option", "default_options", "windowpsd_sparse_batched", "windowpsd_last_timing", "windows_estimate", "windows_estimate_multi", "windowcsd_batched", "DeviceError", "DomainError", "NumericError",
]
