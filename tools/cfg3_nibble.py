#!/usr/bin/env python3
"""cfg3 at the judged size: 32-bit reads of the fixed-point tiles with a STALE NIBBLE PRODUCT refreshed every P iterations (the default of
this handle, P = 32; LPVS_NIB_PERIOD = P) against 36-bit reads (storage = "mixed" by name: leg "36") -- x, z, u against the
extended-precision iterates of the fixture (same G, b by sha256), the f64 oracle's, and the time of the 2000 iterations.  A leg "Ps" runs
the refresh as three kernels of its own (LPVS_NIB_FUSED=0) instead of inside the iteration's launch.
usage: cfg3_nibble.py [legs,...]      default: 36,32,32s,16,64"""
import os; os.environ.setdefault("LPVS_EXPERIMENTS", "1")   # this tool flips experiment knobs of the library (csrc/lpvs_internal.h: experiment_env)
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpvspectral_jl_amd as L
import bench
periods = (sys.argv[1] if len(sys.argv) > 1 else "36,32,32s,16,64").split(",")
fix = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfg3_extended_precision_iterates.npz"))
counts = [int(c) for c in fix["counts"]]
y, X, V, w = bench.synth_signal(1 << 20, 512, 0, torch.device("cuda"))
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
checked = False
for rep in range(2):
    for P in periods:
        os.environ.pop("LPVS_NIB_PERIOD", None); os.environ.pop("LPVS_NIB_FUSED", None); os.environ.pop("LPVS_NIB_RAMP", None)
        with L.Problem.lpv(y, X, V, w, 8) as p:
            if not checked:
                G, bb = p.get_gram()
                assert hashlib.sha256(np.ascontiguousarray(G).tobytes() + np.ascontiguousarray(bb).tobytes()).hexdigest() == str(fix["sha256"]), "fixture of another G, b"
                checked = True; del G
            if P == "36":
                p.set_option("storage", "mixed")
            else:
                if "r" in P:                                   # "32r8": denser first refreshes (LPVS_NIB_RAMP)
                    os.environ["LPVS_NIB_RAMP"] = P.split("r")[1]
                os.environ["LPVS_NIB_PERIOD"] = P.split("r")[0].rstrip("s")
                if P.endswith("s"):
                    os.environ["LPVS_NIB_FUSED"] = "0"
            p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, 512, 16))
            p.admm_init(None, μ=0.05, tol=0.0)
            us, nbytes = p.time_matvec(50)
            p.admm_init(None, μ=0.05, tol=0.0)
            done, ez, eu, eo = 0, [], [], []
            for k, c in enumerate(counts):
                p.admm_run(c - done); done = c
                x, z, u = p.admm_get()
                ez.append(max(rel(x, fix["x"][k]), rel(z, fix["z"][k]))); eu.append(rel(u, fix["u"][k]))
                eo.append(max(rel(x, fix["oracle_x"][k]), rel(z, fix["oracle_z"][k]), rel(u, fix["oracle_u"][k])))
            tm = p.timing()
        print(f"rep {rep} nibble period {P:8s}: x, z vs exact " + " ".join("%.2e" % e for e in ez) + " | u vs exact " + " ".join("%.2e" % e for e in eu)
              + " | max(x, z, u) vs the f64 oracle " + " ".join("%.2e" % e for e in eo)
              + f" | 2000 iterations {tm['admm_ms']:.2f} ms ({tm['xcorr_count']} corrections {tm['xcorr_ms']:.2f} ms; {tm['nibble_refreshes']} refreshes, one stand-alone {tm['nibble_refresh_us']:.1f} us), {nbytes/1e6:.1f} MB per launch, stand-alone product {us:.2f} us", flush=True)
