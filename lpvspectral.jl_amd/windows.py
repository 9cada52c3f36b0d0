"""Window bookkeeping of src/windows.jl (integer arithmetic; results must be bit-exact).

``Windows2`` / ``Windows3`` wrap DSP.jl's ``arraysplit`` exactly as the reference does: window i
(0-based) covers samples ``[i*(n-noverlap), i*(n-noverlap)+n)``, the count is
``(L-n)÷(n-noverlap)+1``; the window function is NOT applied, it is exposed as ``.W``
(src/windows.jl:13-16).  Offsets come from the C-ABI (``lpvs_window_offsets``), the slices are views.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, out_ptr


def rect(n):
    """DSP.rect"""
    return np.ones(int(n))


def hanning(n):
    """DSP.hanning (symmetric, zero end points)."""
    n = int(n)
    return 0.5 * (1 + np.cos(2 * np.pi * np.linspace(-0.5, 0.5, n))) if n > 1 else np.ones(1)


def _offsets(L, n, noverlap):
    k = C.c_int64(0)
    check(lib().lpvs_window_count(int(L), int(n), int(noverlap), C.byref(k)))
    off = np.zeros(max(int(k.value), 1), dtype=np.int64)
    check(lib().lpvs_window_offsets(int(L), int(n), int(noverlap), out_ptr(off), len(off), C.byref(k)))
    return off[: int(k.value)]


class AbstractWindows:
    def __len__(self):
        return len(self.offsets)

    def collect(self):
        return [tuple(np.array(p, copy=True) for p in w) for w in self]


class Windows2(AbstractWindows):
    """``Windows2(y, t, n=length(y)>>3, noverlap=n>>1, window_func=rect)`` (src/windows.jl:7-42)."""

    def __init__(self, y, t, n=None, noverlap=-1, window_func=rect):
        self.y, self.t = y, t
        n = len(y) >> 3 if n is None else int(n)
        if noverlap < 0:                       # src/windows.jl:29
            noverlap = n >> 1
        assert len(y) == len(t), "y and t has to be the same length"   # src/windows.jl:31
        self.n, self.noverlap = n, int(noverlap)
        self.W = np.asarray(window_func(n), dtype=np.float64)          # src/windows.jl:32
        self.offsets = _offsets(len(y), n, self.noverlap)              # arraysplit, :33-34

    def __iter__(self):                        # src/windows.jl:39-42
        for o in self.offsets:
            yield self.y[o:o + self.n], self.t[o:o + self.n]


class Windows3(AbstractWindows):
    """``Windows3(y, t, v, n, noverlap, window_func)`` (src/windows.jl:84-110)."""

    def __init__(self, y, t, v, n=None, noverlap=-1, window_func=rect):
        self.y, self.t, self.v = y, t, v
        n = len(y) >> 3 if n is None else int(n)
        assert len(y) == len(t) == len(v), "y, t and v has to be the same length"  # src/windows.jl:96
        if noverlap < 0:
            noverlap = n >> 1
        self.n, self.noverlap = n, int(noverlap)
        self.W = np.asarray(window_func(n), dtype=np.float64)
        self.offsets = _offsets(len(y), n, self.noverlap)

    def __iter__(self):
        for o in self.offsets:
            yield self.y[o:o + self.n], self.t[o:o + self.n], self.v[o:o + self.n]


def merge(yf, w: AbstractWindows):
    """``merge(yf, w)`` (src/windows.jl:57-70): overlap-average per-window vectors back to full length."""
    arr = np.ascontiguousarray(np.asarray(yf, dtype=np.float64))
    assert arr.shape == (len(w), w.n), "one vector of window length per window"
    ym = np.zeros(len(w.y))
    check(lib().lpvs_merge_f64(out_ptr(arr), len(w), w.n, w.noverlap, len(w.y), out_ptr(ym)))
    return ym


def mapwindows(f, *args):
    """``mapwindows(f, W)`` / ``mapwindows(f, y, t, ...)`` (src/windows.jl:50-55): f takes ``(y,t)``."""
    W = args[0] if len(args) == 1 and isinstance(args[0], AbstractWindows) else Windows2(*args)
    return merge([f(w) for w in W], W)
