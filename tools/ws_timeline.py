#!/usr/bin/env python3
"""Where a launch of cfg5's multi-signal tile product (symv_tile_mfma_ws_kernel: 8 channels sharing M, n = 32768) spends its time, BY ROLE,
from the instrumented debug build (make -C lpvspectral.jl_amd/csrc timeline3 -> liblpvspectral_timeline3.so): one wave of each role of every
persistent workgroup -- P1 (row sums, wave 0), P2 (column sums, wave 2), loader (wave 4) -- accounts, with the 100-MHz wall clock, for the time it
WAITS at the stage barriers; the loader also for the time inside `put` (waiting for a stage's bytes + decoding them into the LDS image).
The role that waits least at the barriers is the one the others wait for.
usage: ws_timeline.py [--product]     (--product: the same launches on the product library, for the instrument's own cost)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = "--product" in sys.argv
if not PRODUCT:
    os.environ["LPVS_LIBRARY"] = os.path.join(ROOT, "lpvspectral.jl_amd", "liblpvspectral_timeline3.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import lpvspectral_jl_amd as L
from lpvspectral_jl_amd._lib import lib
import bench

C5 = bench.CFG5
Y, X, V, w = bench.synth_channels(1 << C5["log2n"], C5["Nf"], C5["channels_per_gpu"], torch.device("cuda"))
TICK_US = 0.01
with L.Problem.lpv_multi(Y, X, V, w, C5["Nv"]) as p:
    p.set_prox(L.IndBallL0(C5["r"]))
    p.admm_init(None, μ=C5["mu"], tol=0.0)
    info = p.matvec_info()
    assert info["kernel"] == "symv_tile_mfma_ws_kernel", info
    us, nbytes = p.time_matvec(30)
    print(f"# {'product' if PRODUCT else 'instrumented'} library: {us:.1f} us per launch of {info['kernel']} ({nbytes * 1e-9:.2f} GB of tiles, {nbytes / us * 1e-6:.2f} TB/s); {info['storage'][:90]}")
    if PRODUCT:
        sys.exit(0)
    setter = lib().lpvs_debug_set_timeline_ws
    setter.restype, setter.argtypes = C.c_int32, [C.c_void_p]
    nwg = 256
    buf = torch.zeros(nwg * 32, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert setter(C.c_void_p(buf.data_ptr())) == 0
    us2, _ = p.time_matvec(5)
    torch.cuda.synchronize()
    assert setter(None) == 0
    rec = buf.cpu().numpy().reshape(nwg, 32)
rec = rec[rec[:, 2] != 0]
print(f"# {len(rec)} persistent workgroups (one per CU), last of {5} back-to-back launches; {us2:.1f} us per launch while stamping; times in us (10-ns ticks)")
t0 = min(rec[:, 0].min(), rec[:, 8].min(), rec[:, 16].min())
pct = lambda v, q: float(np.percentile(v, q))


def describe(name, v):
    print(f"{name:64s} min {v.min() * TICK_US:7.2f}  p10 {pct(v, 10) * TICK_US:7.2f}  median {pct(v, 50) * TICK_US:7.2f}  p90 {pct(v, 90) * TICK_US:7.2f}  max {v.max() * TICK_US:7.2f}")


span = max(rec[:, 2].max(), rec[:, 10].max(), rec[:, 18].max()) - t0
print(f"launch span (first entry -> last wave's end): {span * TICK_US:.1f} us")
for r, name in ((0, "P1 wave (row sums; keeps the B operand in registers)"), (1, "P2 wave (column sums)"), (2, "loader wave (buffer loads -> decode -> LDS image)")):
    ent, end, wait, nbar, put = (rec[:, 8 * r + k].astype(np.int64) for k in (0, 2, 3, 4, 5))
    life = end - ent
    print(f"\n== {name}: {int(np.median(nbar))} barriers per workgroup (median)")
    describe("entry after the launch's first", ent - t0)
    describe("end after the launch's first entry", end - t0)
    describe("lifetime", life)
    describe("time waiting at the stage barriers", wait)
    print(f"{'  ... as a fraction of the lifetime':64s} min {(wait / life).min():7.3f}  p10 {pct(wait / life, 10):7.3f}  median {pct(wait / life, 50):7.3f}  p90 {pct(wait / life, 90):7.3f}  max {(wait / life).max():7.3f}")
    describe("  ... per barrier", wait / np.maximum(nbar, 1))
    if r == 2:
        describe("time inside put (wait for the stage's bytes + decode + LDS stores)", put)
        print(f"{'  ... as a fraction of the lifetime':64s} median {pct(put / life, 50):7.3f};   the rest (issuing the next stage's loads, scalar bookkeeping): median {pct((life - put - wait) / life, 50):7.3f}")
        describe("  ... per stage", put / np.maximum(nbar, 1))
        arr = rec[:, 8 * r + 6].astype(np.int64)
        describe("  of which waiting for the stage's bytes (until its youngest load returned)", arr)
        describe("  ... per stage", arr / np.maximum(nbar, 1))
        describe("  decode + LDS stores per stage", (put - arr) / np.maximum(nbar, 1))
xcc = rec[:, 24] & 0xF
end0 = rec[:, 2] - t0
print("\nper XCD: workgroups / median end / last end (us): " + "  ".join(f"x{int(k)}: {int((xcc == k).sum())} / {np.median(end0[xcc == k]) * TICK_US:.1f} / {end0[xcc == k].max() * TICK_US:.1f}" for k in np.unique(xcc)))
