#!/usr/bin/env python3
"""cfg4: how large are the off-diagonal 128 x 128 tiles of a window's (Q + I/mu)^-1 against its largest entry?  (What a narrower
fixed-point format, or skipping numerically empty tiles, could save -- not built; numbers for DESIGN 7.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
n, Nf = 1 << 16, 256
y, t, f = bench.synth_windows(4, n, Nf, torch.device("cuda"))
for win, equidistant in ((1, True), (2, False)):
    yi = y[win * n:(win + 1) * n].cpu().numpy(); ti = t[win * n:(win + 1) * n].cpu().numpy()
    if not equidistant:
        ti = np.sort(ti[0] + np.random.default_rng(1).random(n) * n)           # the same span, sampled at random instants
    with L.Problem.fourier(yi, ti, f, np.ones(n)) as p:
        M = p.get_inverse(1.0 / 1e-4)
    nn = M.shape[0]; npad = -(-nn // 128) * 128
    Mp = np.zeros((npad, npad)); Mp[:nn, :nn] = M
    mx = np.abs(M).max()
    limit = 2.0 ** -44 * mx * np.sqrt(8192.0 / npad)
    print(f"window {win} ({'equidistant' if equidistant else 'random instants'}): n = {nn}, max|M| = {mx:.3e}, admissible step = 2^{np.log2(limit / mx):.1f} max|M|")
    for I in range(npad // 128):
        for J in range(I):
            T = np.abs(Mp[I * 128:(I + 1) * 128, J * 128:(J + 1) * 128])
            rowmax = T.max(axis=1)
            st = 2.0 ** (np.ceil(np.log2(np.maximum(rowmax, 1e-300))) - 35)
            print(f"   tile ({I},{J}): max|m| = 2^{np.log2(T.max() / mx):6.1f} max|M|; rows whose 36-bit step leaves >= 4 spare bits: {(16 * st <= limit).sum():3d} / 128;  >= 8 spare bits: {(256 * st <= limit).sum():3d}")
