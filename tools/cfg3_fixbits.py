#!/usr/bin/env python3
"""cfg3 at the judged size: how few significant bits the fixed-point tiles of the packed inverse may keep once the x-update correction
removes the storage error's systematic part (LPVS_FIX_BITS zeroes low bits of the 36 in the SAME format: accuracy only, the bytes do not
change), for several correction schedules -- against the extended-precision iterates of the fixture (same G, b by sha256).
usage: cfg3_fixbits.py [bits,...] [schedules,...]"""
import os; os.environ.setdefault("LPVS_EXPERIMENTS", "1")   # this tool flips experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
bits = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "36,32,30,28").split(",")]
scheds = (sys.argv[2] if len(sys.argv) > 2 else "d512,e256,e128,0").split(",")
fix = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfg3_extended_precision_iterates.npz"))
counts = [int(c) for c in fix["counts"]]
y, X, V, w = bench.synth_signal(1 << 20, 512, 0, torch.device("cuda"))
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
checked = False
for b in bits:
    for sc in scheds:
        os.environ["LPVS_FIX_BITS"] = str(b); os.environ["LPVS_XUPDATE_CORRECTION"] = sc
        with L.Problem.lpv(y, X, V, w, 8) as p:
            if not checked:
                G, bb = p.get_gram()
                assert hashlib.sha256(np.ascontiguousarray(G).tobytes() + np.ascontiguousarray(bb).tobytes()).hexdigest() == str(fix["sha256"]), "fixture of another G, b"
                checked = True; del G
            p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, 512, 16))
            p.admm_init(None, μ=0.05, tol=0.0)
            done, errs, erru, erro = 0, [], [], []
            for k, c in enumerate(counts):
                p.admm_run(c - done); done = c
                x, z, u = p.admm_get()
                errs.append(rel(z, fix["z"][k])); erru.append(rel(u, fix["u"][k]))
                erro.append(max(rel(x, fix["oracle_x"][k]), rel(z, fix["oracle_z"][k]), rel(u, fix["oracle_u"][k])))
            tm = p.timing()
        print(f"fixed-point tiles with {b} significant bits, correction schedule {sc:5s}: device vs exact after {counts}: z " + " ".join("%.2e" % e for e in errs)
              + " | u " + " ".join("%.2e" % e for e in erru) + " | max(x, z, u) vs the f64 oracle " + " ".join("%.2e" % e for e in erro)
              + f"   ({tm['xcorr_count']} corrections, {tm['xcorr_ms']:.2f} ms of {tm['admm_ms']:.2f})", flush=True)
