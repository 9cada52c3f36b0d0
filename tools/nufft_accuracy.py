#!/usr/bin/env python3
"""Accuracy of the type-1 non-uniform FFT of csrc/nufft.hip, restated in numpy on the host (no GPU): "exponential of semicircle"
kernel exp(beta (sqrt(1 - z^2) - 1)), w = 16 points wide, beta = 2.30 w, fine grid nf = power of two >= 3.9 nslots, deconvolution by
the kernel's Fourier transform (Gauss-Legendre quadrature) -- against the direct sums in long double, for several kernel widths and
for the fixed-point rounding of the spreading.

usage: nufft_accuracy.py [N] [nslots]         (defaults 20000 1032: the cfg3 slot count at a host-sized N)"""
import sys
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nslots = int(sys.argv[2]) if len(sys.argv) > 2 else 1032
rng = np.random.default_rng(0)
theta = rng.random(N) * 2 * np.pi
c = rng.standard_normal(N) * np.exp(rng.standard_normal(N))       # weights with a heavy tail (max / mean ~ 30)

# direct sums, long double
j = np.arange(nslots, dtype=np.longdouble)
ref = np.zeros(nslots, dtype=np.clongdouble)
tl = theta.astype(np.longdouble)
for lo in range(0, N, 2000):
    ph = np.outer(j, tl[lo:lo + 2000])
    ref += (np.cos(ph) + 1j * np.sin(ph)) @ c[lo:lo + 2000].astype(np.longdouble)
scale = np.abs(c).sum()

nf = 256
while nf < 3.9 * nslots:
    nf *= 2


def phihat(w, beta, nf, modes):
    """Fourier transform of the kernel at the modes (per grid spacing): int_{-w/2}^{w/2} phi(2 t / w) cos(2 pi j t / nf) dt"""
    xg, wg = np.polynomial.legendre.leggauss(96)
    t = 0.5 * w * xg.astype(np.longdouble)
    ph = np.exp(beta * (np.sqrt(1 - (2 * t / w) ** 2) - 1))
    return np.array([(0.5 * w * wg * ph * np.cos(2 * np.pi * m * t / nf)).sum() for m in modes.astype(np.longdouble)])


def nufft1(w, beta, fixed_point):
    p = theta / (2 * np.pi) * nf
    g0 = np.ceil(p - 0.5 * w).astype(np.int64)
    grid = np.zeros(nf, dtype=np.int64 if fixed_point else np.float64)
    quantum = N * np.abs(c).max() * 2.0 ** -62
    for k in range(w):
        z = ((g0 + k) - p) * (2.0 / w)
        tap = np.exp(beta * (np.sqrt(np.maximum(1 - z * z, 0.0)) - 1))
        if fixed_point:
            np.add.at(grid, (g0 + k) % nf, np.rint(c / quantum * tap).astype(np.int64))
        else:
            np.add.at(grid, (g0 + k) % nf, c * tap)
    g = grid.astype(np.float64) * (quantum if fixed_point else 1.0)
    # pruned DFT (e^{+i j theta}) of the wanted modes, then deconvolution
    modes = np.arange(nslots)
    F = np.exp(2j * np.pi * np.outer(modes, np.arange(nf)) / nf) @ g
    return F / phihat(w, beta, nf, modes).astype(np.float64)


print(f"N = {N}, nslots = {nslots}, nf = {nf} (oversampling {nf / (2 * nslots):.2f} of the two-sided range), sum|c| = {scale:.3e}, "
      f"max|c| / mean|c| = {np.abs(c).max() / np.abs(c).mean():.1f}")
for w in (8, 12, 14, 16):
    for fx in (False, True):
        got = nufft1(w, 2.30 * w, fx)
        err = np.abs(got - ref.astype(np.complex128)).max() / scale
        print(f"  w = {w:2d}  beta = {2.30 * w:5.2f}  {'64-bit fixed-point grid' if fx else 'double grid            '}: max |F - F_direct| / sum|c| = {err:.2e}")
