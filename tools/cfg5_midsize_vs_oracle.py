#!/usr/bin/env python3
"""cfg5's kernel chain (symv_tile_mfma_ws_kernel -> symv_reduce_runs_kernel -> admm_prox_kernel, 8 channels sharing M, IndBallL0(32)) at the
bench's own horizon, at a size the CPU oracle can run: Nf = 256, Nv = 16 -> n = 8192, N = 2^20 (--log2n), 2000 iterations, against oracle.admm_gram_multi
(the Gram-form ADMM of src/lasso.jl:136-171 with ONE Cholesky factor for the channels compared) on the Gram and right-hand sides read back from the
device.  Prints rel-L2 of x, z, u per channel and count, with and without the x-update correction (LPVS_OPT_XUPDATE_CORRECTION), and with
--save FILE writes the oracle's iterates as the fixture tests/test_gpu_configs.py::test_cfg5_kernel_chain_long_horizon_vs_oracle holds the device to
(keyed by the sha256 of G and B: the device Gram is bit-reproducible).  ~4 CPU-minutes per pair of channels.
usage: cfg5_midsize_vs_oracle.py [--save FILE] [--channels 0,5] [counts ...]"""
import argparse, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
from oracle import oracle as o
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--save", default=None)
ap.add_argument("--channels", default="0,5")
ap.add_argument("--legs", default="default,off")
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("counts", nargs="*", type=int)
a = ap.parse_args()
counts = a.counts or [200, 500, 1000, 2000]
chans = [int(c) for c in a.channels.split(",")]
LOG2N, NF, NV, NS, R, MU = a.log2n, 256, 16, 8, 32, 0.05
Y, X, V, w = bench.synth_channels(1 << LOG2N, NF, NS, torch.device("cuda"))
rel = lambda p, q: float(np.linalg.norm(p - q) / max(np.linalg.norm(q), 1e-300))


def device_leg(xcorr):
    with L.Problem.lpv_multi(Y, X, V, w, NV) as p:
        if xcorr != "default":
            p.set_option("xupdate_correction", xcorr)
        p.set_prox(L.IndBallL0(R))
        p.admm_init(None, μ=MU, tol=0.0)
        info = p.matvec_info()
        info["storage"] += " [%.1f MB per launch]" % (p.time_matvec(5)[1] * 1e-6)
        out, done = {}, 0
        t0 = time.time()
        for c in counts:
            p.admm_run(c - done); done = c
            out[c] = p.admm_get()
        dt = time.time() - t0
        G, _ = p.get_gram(); B = p.get_rhs()
        return out, info, G, B, dt


legs = {}
for leg in a.legs.split(","):
    legs[leg], info, G, B, dt = device_leg(leg)
    print(f"device leg xupdate_correction={leg}: kernel {info['kernel']}, storage {info['storage']}, {counts[-1]} iterations in {dt:.2f} s", flush=True)
fp = hashlib.sha256(np.ascontiguousarray(G).tobytes() + np.ascontiguousarray(B).tobytes()).hexdigest()
print(f"n = {G.shape[0]}, sha256(G, B) = {fp[:16]}...; oracle.admm_gram_multi on channels {chans}, counts {counts} ({o.num_threads()} threads)", flush=True)
t0 = time.time()
ora = o.admm_gram_multi(G, B[:, chans], o.IndBallL0(R), counts, mu=MU)
print(f"oracle: {time.time() - t0:.0f} s", flush=True)
for leg, out in legs.items():
    for c in counts:
        x, z, u = out[c]
        for k, ch in enumerate(chans):
            ox, oz, ou = (v[:, k] for v in ora[c])
            same = np.array_equal(z[:, ch] != 0, oz != 0)
            print(f"  {leg:8s} {c:5d} iterations, channel {ch}: x {rel(x[:, ch], ox):.2e} z {rel(z[:, ch], oz):.2e} u {rel(u[:, ch], ou):.2e}  support {'identical' if same else 'DIFFERS'} (nnz {np.count_nonzero(oz)})")
if len(legs) > 1 and "on" in legs and "off" in legs:
    for c in counts:
        d = [max(rel(legs["on"][c][i][:, q], legs["off"][c][i][:, q]) for q in range(NS)) for i in range(3)]
        print(f"  corrected vs uncorrected, {c} iterations, worst channel: x {d[0]:.2e} z {d[1]:.2e} u {d[2]:.2e}")
if a.save:
    np.savez_compressed(a.save, sha256=fp, counts=np.array(counts), channels=np.array(chans),
                        oracle_x=np.stack([ora[c][0] for c in counts]), oracle_z=np.stack([ora[c][1] for c in counts]), oracle_u=np.stack([ora[c][2] for c in counts]))
    print("saved", a.save, os.path.getsize(a.save), "bytes")
