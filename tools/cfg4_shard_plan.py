#!/usr/bin/env python3
"""cfg4's per-GPU shard at 8 / 4 / 2 GPUs (128 / 256 / 512 windows of 2^16 samples, 2000 iterations): the engine's plan -- parts of a
chunk in flight (LPVS_OPT_WINDOWS_IN_FLIGHT) and chunk size -- against the time of the shard; what the strong-scaling curve of
`bench.py --gpus N` (cfg4_strong) will be made of.  usage: cfg4_shard_plan.py [nwin ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
shards = [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024]
n, Nf = 1 << 16, 256
y, t, f = bench.synth_windows(1024, n, Nf, torch.device("cuda"))
for nwin in shards:
    for fly in (1, 2, 3, 4):
        for chunk in (((None, 200, 140, 100, 70, "uncut") if fly > 1 else (None, "uncut")) if os.environ.get("PLAN_SCAN") else (None,)):
            with L.default_options(windows_in_flight=fly, window_chunk_mb=chunk):
                run = lambda: L.windowpsd_sparse_batched(y, t, f, n, 0, None, λ=0.2, μ=1e-4, tol=0.0, iters=2000, win_lo=0, win_hi=nwin)
                run(); torch.cuda.synchronize()
                dts = []
                for _ in range(3):
                    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dts.append(time.perf_counter() - t0)
                dt = min(dts)
            print(f"{nwin:5d} windows, {fly} part(s) in flight, chunks {'default' if chunk is None else str(chunk):8s}: {dt * 1e3:7.1f} ms   ({nwin / dt:7.0f} windows/s; three runs: " + " ".join("%.1f" % (d * 1e3) for d in dts) + ")", flush=True)
