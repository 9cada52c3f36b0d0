#!/usr/bin/env python3
"""cfg5's multi-signal product against the segment length of the run walk (LPVS_MULTI_RUNS = tiles per segment: a workgroup's contiguous
stretch of the packed inverse and the length of a P1 run) -- one process, one handle; 30 back-to-back launches each (HIP events)."""
import os; os.environ.setdefault("LPVS_EXPERIMENTS", "1")   # this tool flips experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lpvspectral_jl_amd as L
import bench
Y, X, V, w = bench.synth_channels(1 << 20, 1024, 8, torch.device("cuda"))
with L.Problem.lpv_multi(Y, X, V, w, 16) as p:
    p.set_prox(L.IndBallL0(32))
    for rep in range(2):
        for runs in ("0", "2", "4", "8", "16", "32", "64"):
            os.environ["LPVS_MULTI_RUNS"] = runs
            p.admm_init(None, μ=0.05, tol=0.0)
            us, nbytes = p.time_matvec(30)
            print(f"rep {rep} LPVS_MULTI_RUNS={runs:3s}: product {us:7.1f} us per launch", flush=True)
