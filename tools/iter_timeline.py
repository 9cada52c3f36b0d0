#!/usr/bin/env python3
"""Per-workgroup timeline of ONE launch of the one-launch ADMM iteration (admm_iter_mixed_kernel<FI_MID>) at the cfg3 size, from the
instrumented debug build (make -C lpvspectral.jl_amd/csrc timeline -> liblpvspectral_timeline.so): every workgroup stamps the 100 MHz
wall clock (s_memrealtime) at entry, after its update (prox + dual step of its two row blocks), after the prologue's last barrier, when
its tile has been consumed, and after its last atomic.  Two consecutive launches are recorded, so the launch boundary is on the record.

usage: iter_timeline.py [log2N] [Nf] [Nv]   (writes a text summary to stdout)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = "--full" in sys.argv                       # all five stamps (perturbs: the stamps in the middle serialise the kernel's overlapped phases)
sys.argv = [a for a in sys.argv if a != "--full"]
os.environ["LPVS_LIBRARY"] = os.path.join(ROOT, "lpvspectral.jl_amd", "liblpvspectral_timeline2.so" if FULL else "liblpvspectral_timeline.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import lpvspectral_jl_amd as L
from lpvspectral_jl_amd._lib import lib
import bench

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Nf = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Nv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
y, X, V, w = bench.synth_signal(1 << lg, Nf, 0, "cuda")
TICK_US = 0.01   # s_memrealtime: 100 MHz

setter = lib().lpvs_debug_set_timeline
setter.restype, setter.argtypes = C.c_int32, [C.c_void_p]
with L.Problem.lpv(y, X, V, w, Nv, True, False) as p:
    p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, Nf, 2 * Nv))
    p.admm_init(None, μ=0.05, tol=0.0)
    info = p.matvec_info()
    assert info["kernel"] == "admm_iter_mixed_kernel" and info["one_launch_iteration"], info
    nblk = (p.n + 127) // 128
    ntiles = nblk * (nblk + 1) // 2
    buf = torch.zeros(2 * ntiles * 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert setter(C.c_void_p(buf.data_ptr())) == 0
    p.admm_run(300)
    torch.cuda.synchronize()
    assert setter(None) == 0
    rec = buf.cpu().numpy().reshape(2, ntiles, 8)

nwg = int((rec[0, :, 1] != 0).sum())             # workgroups of a launch (several tiles per workgroup: fewer than tiles)
assert nwg == int((rec[1, :, 1] != 0).sum()) and (rec[:, :nwg, 1] != 0).all()
rec = rec[:, :nwg]
g = rec[:, :, 0]
assert (g[0] == g[0, 0]).all() and (g[1] == g[1, 0]).all() and abs(int(g[0, 0]) - int(g[1, 0])) == 1, "stamps of mixed launches"
first, second = (0, 1) if g[0, 0] < g[1, 0] else (1, 0)
A, B = rec[first].astype(np.int64), rec[second].astype(np.int64)
print(f"# workgroups per launch: {nwg}" + (" (one tile each)" if nwg == ntiles else ""))
ntiles_all, ntiles = ntiles, nwg
print(f"# stamps: {'all five (perturbing)' if FULL else 'entry and end only'}")
print(f"# admm_iter_mixed_kernel<FI_MID>, n = {p.n} ({nblk} row blocks, {ntiles} workgroups), launches g = {int(A[0, 0])} and {int(B[0, 0])}; times in us, clock 100 MHz (10 ns ticks)")


def pct(v, q):
    return float(np.percentile(v, q))


def describe(name, v):
    print(f"{name:58s} min {v.min() * TICK_US:6.2f}  p10 {pct(v, 10) * TICK_US:6.2f}  median {pct(v, 50) * TICK_US:6.2f}  p90 {pct(v, 90) * TICK_US:6.2f}  max {v.max() * TICK_US:6.2f}")


for tag, R in (("launch A", A), ("launch B", B)):
    t0 = R[:, 1].min()
    ent, upd, bar, cons, end = (R[:, k] - t0 for k in (1, 2, 3, 4, 5))
    print(f"\n== {tag}: span (first entry -> last workgroup's last atomic) {end.max() * TICK_US:.2f} us")
    describe("entry (after the launch's first entry)", ent)
    if FULL:
        describe("update done - entry (state loads, prox, dual step)", upd - ent)
        describe("prologue barrier passed - entry", bar - ent)
        describe("tile consumed - barrier (wait for the tile + product)", cons - bar)
        describe("last atomic - tile consumed (butterflies, LDS, atomics)", end - cons)
    describe("workgroup lifetime (entry -> last atomic)", end - ent)
    diag = np.arange(ntiles) < nblk
    describe("  diagonal (float-head, 96 KB) workgroups: lifetime", (end - ent)[diag])
    describe("  off-diagonal (36-bit, 74 KB per tile) workgroups: lifetime", (end - ent)[~diag])
    # occupancy over time: workgroups alive per microsecond
    edges = np.arange(0, end.max() + 100, 100)
    alive = [(int(((ent <= e) & (end > e)).sum())) for e in edges]
    print("workgroups alive at t = 0, 1, 2, ... us: " + " ".join(str(a) for a in alive))
    started = [(int((ent <= e).sum())) for e in edges]
    print("workgroups started by  t = 0, 1, 2, ... us: " + " ".join(str(a) for a in started))
    done = [(int((end <= e).sum())) for e in edges]
    print("workgroups finished by t = 0, 1, 2, ... us: " + " ".join(str(a) for a in done))
    xcc = R[:, 6] & 0xF
    print("per XCD: workgroups / last end (us): " + "  ".join(f"x{int(k)}: {int((xcc == k).sum())} / {end[xcc == k].max() * TICK_US:.2f}" for k in np.unique(xcc)))
    print(f"workgroup index -> XCD of the first 16: {[int(k) for k in xcc[:16]]}")
    last = np.argsort(end)[-8:]
    print("the 8 last workgroups to finish (index, entry, lifetime us): " + "  ".join(f"({int(i)}, {ent[i] * TICK_US:.2f}, {(end - ent)[i] * TICK_US:.2f})" for i in last))
gap = B[:, 1].min() - A[:, 5].max()
print(f"\nlaunch boundary: first entry of launch B - last atomic of launch A = {gap * TICK_US:.2f} us;  launch period (first entry to first entry) = {(B[:, 1].min() - A[:, 1].min()) * TICK_US:.2f} us")
