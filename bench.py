#!/usr/bin/env python3
"""Headline benchmark: ls_sparse_spectral_lpv (frequency-grouped lasso) at BASELINE.json's quoted
size N=2^20, Nf=512, Nv=8 (n = 8192 unknowns) on N GPUs of one node, one process per GPU.

A step = one complete solve of one synthetic signal whose inputs (y, X, V, w) are already resident
in HBM: basis tables -> Gram + rhs -> factorisation of (G + I/mu) -> 2000 ADMM iterations (tol = 0,
so exactly 2000) -> parameter read-back.  The workload's w is the reference's uniform grid, so the
library takes its structured Gram (nudft.hip, VALU f64 -- the MFMA Gram is NOT on this workload's path);
the dense f64-MFMA Gram used for arbitrary w is measured once more outside the timed region and reported
under "gram_general_path".  Ranks solve independent signals (weak scaling, no data-path collective);
RCCL is used only for the final gather of the coefficient vectors.

`--workload cfg4` is BASELINE.json's batched-window configuration (ls_windowpsd with the sparse estimator,
1024 windows x 2^16 samples, Nf = 256): a step = all 1024 windows, the window range is sharded over the
ranks (strong scaling), one RCCL all_gather of the per-window coefficients, PSD summed in window order.

`--workload cfg2` (ls_sparse_spectral NormL1(0.01), N = 2^18, Nf = 512, 5000 iterations) and `--workload cfg5` (multichannel LPV,
8 channels per GPU sharing (X, V), N = 2^20, Nf = 1024, Nv = 16 -> n = 32768, IndBallL0(32)) are BASELINE.json's other two GPU
configurations, each with its own `roofline` and `cpu_baseline`.

With the default workload every run ALSO measures cfg4 with the window range sharded over the ranks and puts it on the same JSON line
as `cfg4_strong` (windows/s, per-rank window counts, the collective's backend and rank count): the driver's fixed command
`bench.py --gpus N` thus yields both curves north_star asks for -- independent signals (weak) and batched windows (strong).

`python3 bench.py --gpus N` starts the N ranks itself (a torch.distributed.run CHILD process, started before this process touches
the GPU) when it was not launched by one; under a profiler preload it refuses to (profile single-rank runs).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement) with
  roofline     : the dominant kernel (the ADMM mat-vec, HBM-bound) against the HBM peak, from HIP-event
                 timings taken inside the library on the stream the kernel runs on
  cpu_baseline : the faithful CPU restatement of the reference algorithm (oracle, "port") timed on
                 this host on a bounded sample and extrapolated (rank 0, N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

LOG2N, NF, NV = 20, 512, 8
ADMM_ITERS, LAMBDA, MU = 2000, 5.0, 0.05
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s (spec)
HBM_STREAM_GBS = 6290.0        # MI355X_MICROARCH.md: 6.29 TB/s measured (float4 copy); tools/stream_read.hip reads this workload's shape at 7.0
F64_MFMA_PEAK_TFLOPS = 78.6   # AMD datasheet FP64 matrix; MI355X_MICROARCH.md lists no f64 MFMA row
# cfg4: L = 2^26 equidistant, 1024 windows of 2^16, freqs (0:255)/512, L1 lam = 0.2, mu = 1e-4, 2000 its
CFG4 = dict(nwin=1024, log2n=16, Nf=256, lam=0.2, mu=1e-4, iters=2000)
# cfg2: N = 2^18 non-equidistant t, f = (1:512)/1024 (no zero frequency: n = 1024), NormL1(0.01), mu = 0.05, 5000 iterations, tol = 0
CFG2 = dict(log2n=18, Nf=512, lam=0.01, mu=0.05, iters=5000)
# cfg5: channels sharing (X, V), N = 2^20, Nf = 1024, Nv = 16 (n = 32768), IndBallL0(32); 8 channels per GPU
CFG5 = dict(log2n=20, Nf=1024, Nv=16, r=32, mu=0.05, iters=2000, channels_per_gpu=8)


def roof_fracs(achieved_gbs):
    """Both fractions: against the 8 TB/s datasheet peak (`frac`, the contract's figure) and against the streaming rate the guide
    measured on this part (6.29 TB/s)."""
    return {"frac": achieved_gbs / HBM_PEAK_GBS, "frac_of_measured_stream_6290_GBps": achieved_gbs / HBM_STREAM_GBS}


def guarded(fn, what):
    """A sub-record must never cost the main line: any exception inside one becomes {"error": ...} in its place."""
    try:
        return fn()
    except (KeyboardInterrupt, SystemExit):
        raise
    except BaseException as e:                                  # noqa: BLE001 (a sub-record: report, do not propagate)
        import traceback
        sys.stderr.write("[bench] sub-record %s failed:\n%s\n" % (what, traceback.format_exc()))
        return {"error": ("%s: %s" % (type(e).__name__, e))[:600]}


DRIVER_MAX_KEYS, DRIVER_MAX_STR = 20, 120


def _scalar(v):
    """A value the driver's parser keeps: bool / None / number (rounded to 6 significant digits: the record has a length budget too) / short string."""
    if isinstance(v, bool) or v is None:
        return True, v
    if isinstance(v, (int, np.integer)):
        return True, int(v)
    if isinstance(v, (float, np.floating)):
        if not np.isfinite(v):
            return True, None
        return True, int(v) if (float(v).is_integer() and abs(v) < 2 ** 53) else float("%.6g" % float(v))   # (byte / launch counts stay exact)
    if isinstance(v, str):
        return True, v[:DRIVER_MAX_STR]
    return False, None


def _take(priority, rest):
    """The first DRIVER_MAX_KEYS scalar items: the priority list first (entries that are None or not scalars are skipped), then whatever scalars of
    the full record still fit, in its own order."""
    out = {}
    for k, v in list(priority) + [(k, v) for k, v in (rest or {}).items()]:
        ok, val = _scalar(v)
        if not ok or val is None or k in out or len(out) >= DRIVER_MAX_KEYS or len(k) > 36:
            continue
        out[k] = val
    return out


def driver_record(o):
    """Make the JSON line fit the driver's parser (BENCH_r05.json: of `config`, `roofline` and `cpu_baseline` it kept the first ~21 SCALAR entries -- about 1100
    characters each --, dropped nested records and unknown top-level keys; round 5's 70 flattened keys lost cfg4, cfg5 and the 8-byte figure that way).
    So: `config`, `roofline`, `cpu_baseline` become at most 20 scalars each, numbers first, short names, no prose -- every BASELINE configuration's value and
    roofline fraction, the iteration rate, the phase times, the like-for-like figures (8-byte M, host arrays, two solves in flight), the MFMA fractions --
    and the FULL records (nested sub-records, notes, formulas) move to `detail`, which the driver drops and a reader keeps."""
    cfg, roof, cpu = o.get("config") or {}, o.get("roofline") or {}, o.get("cpu_baseline")
    o["detail"] = {"config": cfg, "roofline": roof, **({"cpu_baseline": cpu} if cpu is not None else {})}
    g = lambda d, *ks: (g(d.get(ks[0]) or {}, *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    ph = o.get("phase_ms") or {}
    errs = [k for k in ("cfg2", "cfg4_strong", "cfg5", "cfg3_row_sharded", "gram_general_path") if isinstance(o.get(k), dict) and "error" in o[k]]
    errs += ["single_process_" + k for k in ("cfg3", "cfg4", "cfg5") if "error" in (g(o, "single_process", "single_process_" + k) or {})]
    errs += [k for k in ("two_solves_in_flight", "whole_step_with_8_byte_storage", "from_host_arrays") if "error" in (cfg.get(k) or {})]
    gram_ms = sum(ph[k] for k in ("basis_ms", "gram_ms", "reduce_rhs_ms")) if all(k in ph for k in ("basis_ms", "gram_ms", "reduce_rhs_ms")) else None
    one4 = g(o, "single_process", "single_process_cfg4")
    prio_cfg = [
        ("workload", cfg.get("workload")),
        ("admm_iters_per_sec", o.get("admm_iters_per_sec")),
        ("phase_factor_ms", ph.get("factor_ms")), ("phase_admm_ms", ph.get("admm_ms")), ("phase_gram_rhs_ms", gram_ms),
        ("cfg4_windows_per_s", g(o, "cfg4_strong", "value")), ("cfg4_roofline_frac", g(o, "cfg4_strong", "roofline", "frac")),
        ("cfg4_iteration_us", g(o, "cfg4_strong", "roofline", "launch_us")),
        ("cfg5_signals_per_s", g(o, "cfg5", "value")), ("cfg5_roofline_frac", g(o, "cfg5", "roofline", "frac")),
        ("cfg5_iteration_ms", g(o, "cfg5", "iteration_ms_all_channels")),
        ("cfg2_signals_per_s", g(o, "cfg2", "value")), ("cfg2_iteration_us", g(o, "cfg2", "roofline", "launch_us")),
        ("signals_per_s_8_byte_M", g(cfg, "whole_step_with_8_byte_storage", "signals_per_s_per_gpu")),
        ("signals_per_s_host_arrays", g(cfg, "from_host_arrays", "signals_per_s_per_gpu")),
        ("two_in_flight_signals_per_s", g(cfg, "two_solves_in_flight", "value")),
        ("sub_record_errors", len(errs)),
        ("first_error", (errs[0] + ": " + str((o.get(errs[0]) or g(o, "single_process", errs[0]) or cfg.get(errs[0]) or {}).get("error"))) if errs else None),
        ("rccl_ranks", o.get("rccl_ranks")), ("xcorr_per_solve", cfg.get("xupdate_corrections_per_solve")),
        ("rowsharded_signals_per_s", g(o, "cfg3_row_sharded", "value")),
        ("one_process_cfg4_windows_per_s", (one4 or {}).get("value") if "error" not in (one4 or {}) else None),
        ("nibble_refreshes_per_solve", cfg.get("nibble_refreshes_per_solve")),
    ]
    # the workloads' own lines (--workload cfg2 | cfg4 | cfg5): their few scalars follow in the record's own order
    o["config"] = _take(prio_cfg, cfg)
    if roof:
        m8 = roof.get("same_matvec_with_8_byte_storage") or {}
        fz, gg = o.get("factorisation") or {}, o.get("gram_general_path") or {}
        prio_roof = [
            ("bound", roof.get("bound")), ("kernel", (roof.get("kernel") or "").split(" ")[0] or None), ("achieved", roof.get("achieved")), ("peak", roof.get("peak")),
            ("unit", roof.get("unit")), ("frac", roof.get("frac")), ("traffic", roof.get("traffic")),
            ("bytes_per_launch", roof.get("algorithmic_bytes_per_launch")), ("launch_us", roof.get("launch_us")),
            ("launch_us_all_of_admm", roof.get("launch_us_all_of_admm")), ("launches_per_step", roof.get("launches_per_step")), ("share_of_step", roof.get("share_of_step")),
            ("frac_of_stream_6290_GBps", roof.get("frac_of_measured_stream_6290_GBps")),
            ("matvec_8_byte_frac", m8.get("frac_of_hbm_peak")), ("frac_with_36_bit_bytes", roof.get("frac_if_priced_with_36_bit_bytes")),
            ("factor_frac_of_f64_mfma", fz.get("frac", fz.get("frac_of_f64_mfma_peak"))), ("factor_ms", fz.get("ms")),
            ("gram_mfma_frac_of_f64_peak", gg.get("frac")), ("gram_mfma_TFLOPs_issued", gg.get("achieved")), ("gram_mfma_ms", gg.get("launch_ms")),
        ]
        o["roofline"] = _take(prio_roof, {k: v for k, v in roof.items() if k not in ("note", "traffic_source", "kernel", "frac_of_measured_stream_6290_GBps", "algorithmic_bytes_per_launch", "frac_if_priced_with_36_bit_bytes")})
    if isinstance(cpu, dict):
        prio_cpu = [(k, cpu.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "admm_iters_per_sec", "cpus_visible", "threads_used", "error")]
        prio_cpu.append(("gemv_stream_GBps", cpu.get("achieved_gemv_stream_GBps")))
        o["cpu_baseline"] = _take(prio_cpu, None)


def synth_signal(N, Nf, seed, device):
    """README generator, three true frequencies w[{41,205,410}] (1-based)."""
    g = torch.Generator(device=device).manual_seed(0x1B5EC + 3 + seed)
    X = torch.sort(torch.rand(N, dtype=torch.float64, device=device, generator=g) * (10.0 * N / 500)).values
    V = torch.linspace(0, 1, N, dtype=torch.float64, device=device)
    w = torch.tensor(2 * np.pi * (np.arange(Nf) + 1.0) * 25.0 / Nf, dtype=torch.float64, device=device)
    deps = [2 * V ** 2, 2 / (5 * V + 1), 3 * torch.exp(-10 * (V - 0.5) ** 2)]
    idx = [min(40, Nf - 1), min(204, Nf - 1), min(409, Nf - 1)]
    y = sum(d * torch.cos(w[i] * X - 0.5 * d) for d, i in zip(deps, idx))
    y = y + 0.1 * torch.randn(N, dtype=torch.float64, device=device, generator=g)
    return y.contiguous(), X.contiguous(), V.contiguous(), w.contiguous()


def synth_windows(nwin, n, Nf, device):
    """cfg4: one long equidistant record (dt = 1) cut into nwin windows of n samples; two tones + noise."""
    g = torch.Generator(device=device).manual_seed(0x1B5EC + 4)
    Lr = nwin * n
    t = torch.arange(Lr, dtype=torch.float64, device=device)
    f = np.arange(Nf) / (2.0 * Nf)
    y = (torch.sin(2 * np.pi * f[33] * t) + 0.5 * torch.sin(2 * np.pi * f[100] * t)
         + 0.3 * torch.randn(Lr, dtype=torch.float64, device=device, generator=g))
    return y.contiguous(), t, f


def synth_fourier(N, Nf, device):
    """cfg2: t = sorted N*U(0,1) (non-equidistant), f = (1:Nf)/(2Nf) (no zero frequency), five sinusoids at
    f[{17,100,257,300,480}] (1-based) with amplitudes {2,1,.5,.25,.1} + 0.1 N(0,1)."""
    g = torch.Generator(device=device).manual_seed(0x1B5EC + 2)
    t = torch.sort(torch.rand(N, dtype=torch.float64, device=device, generator=g) * N).values
    f = np.arange(1, Nf + 1) / (2.0 * Nf)
    amp = [(2, 16), (1, 99), (.5, 256), (.25, 299), (.1, 479)]
    y = sum(a * torch.sin(2 * np.pi * f[min(i, Nf - 1)] * t + 0.3 * i) for a, i in amp)
    y = y + 0.1 * torch.randn(N, dtype=torch.float64, device=device, generator=g)
    return y.contiguous(), t.contiguous(), f


def synth_channels(N, Nf, ns, device, first=0):
    """cfg5: ns channels sharing (X, V), 3-6 true frequencies each (channel index first + q decides which)."""
    g = torch.Generator(device=device).manual_seed(0x1B5EC + 5)
    X = torch.sort(torch.rand(N, dtype=torch.float64, device=device, generator=g) * (10.0 * N / 500)).values
    V = torch.linspace(0, 1, N, dtype=torch.float64, device=device)
    w = torch.tensor(2 * np.pi * (np.arange(Nf) + 1.0) * 25.0 / Nf, dtype=torch.float64, device=device)
    cols = []
    for q in range(first, first + ns):
        gq = torch.Generator(device=device).manual_seed(0x1B5EC + 50 + q)
        col = sum((1.0 + 0.3 * k) * torch.cos(w[(37 * q + 101 * k) % Nf] * X + 0.1 * k) * (1 + V * (k % 2)) for k in range(3 + q % 4))
        cols.append(col + 0.1 * torch.randn(N, dtype=torch.float64, device=device, generator=gq))
    return torch.stack(cols, dim=1).contiguous(), X.contiguous(), V.contiguous(), w.contiguous()


def solve(L, y, X, V, w, Nv, iters, device_index):
    with L.Problem.lpv(y, X, V, w, Nv, True, False, device=device_index) as p:
        p.set_prox(L.SlicedSeparableSum.frequency_groups(LAMBDA, len(w), 2 * Nv))
        p.admm_init(None, μ=MU, tol=0.0)
        it, nxz, conv = p.admm_run(iters)
        params = p.params(0)
        tm = p.timing()
    return params, it, nxz, tm


def solve_rowsharded(L, ys, Xs, Vs, w, Nv, iters, device_index, dist):
    """--row-sharded: ONE signal, sample rows split over the ranks: partial Gram per rank, one RCCL
    all-reduce of G and b, ADMM replicated."""
    ranges = L.sharding.allreduce_ranges(L.lpv_ranges(Xs, Vs), dist)
    with L.Problem.lpv_rows(ys, Xs, Vs, w, Nv, ranges, True, False, device=device_index) as p:
        G, b = p.device_gram()
        L.sharding.allreduce_sum_(G, dist)
        L.sharding.allreduce_sum_(b, dist)
        p.gram_modified()
        p.set_prox(L.SlicedSeparableSum.frequency_groups(LAMBDA, len(w), 2 * Nv))
        p.admm_init(None, μ=MU, tol=0.0)
        it, nxz, conv = p.admm_run(iters)
        params = p.params(0)
        tm = p.timing()
    return params, it, nxz, tm


# ---------------------------------------------------------------------------------------------------------------- CPU baseline
# The CPU port (oracle/, "kind": "port") is timed in a CHILD process so that its OpenMP runtime starts with pinned threads
# (OMP_PROC_BIND=spread, OMP_PLACES=cores) and no other thread pool of this process competes for the cores: round 1 and round 2 timed
# the same sample at 23 s and 44 s on "128 cores".  The child first picks the thread count that streams fastest on a small probe
# (memory-bound gemv stops scaling long before 128 threads, and a GPU box may grant this job a fraction of its cores), then times
# TWO sample sizes, best of 3 each, so that the linear-in-N law the extrapolation rests on is on the record next to it.
def _set_threads(o, T):
    import ctypes
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(T))
    return o.num_threads()


def _cpu_child(which):
    from oracle import oracle as o
    ncpu = len(os.sched_getaffinity(0))
    cands = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu})
    rec = {"which": which, "cpus_visible": ncpu}
    if which == "cfg3":
        def run(lg, iters):
            y, X, V, w = [a.numpy() for a in synth_signal(1 << lg, NF, 0, "cpu")]
            t0 = time.time(); Phi = o.lpv_regressor(X, V, w, NV); t_asm = time.time() - t0
            t0 = time.time(); r = o.admm_ls(Phi, y, o.GroupL2(LAMBDA, 2 * NV), iters=iters, tol=0.0, mu=MU); t = time.time() - t0
            return dict(asm_s=t_asm, admm_s=t, iters=r["iters"], cg=r["cg_iters"])
        probe = {}
        for T in cands:                                        # one ADMM iteration at N = 2^12 (0.25 GiB regressor) per candidate
            _set_threads(o, T); run(12, 1); probe[T] = run(12, 1)["admm_s"]
        T = min(probe, key=probe.get); _set_threads(o, T)
        rec.update(threads=T, probe_s={str(k): round(v, 3) for k, v in probe.items()}, sizes={})
        for lg in (13, 14):
            reps = [run(lg, 2) for _ in range(3)]
            best = min(reps, key=lambda r: r["admm_s"])
            rec["sizes"][str(lg)] = dict(best, all_admm_s=[round(r["admm_s"], 3) for r in reps])
    elif which == "cfg4":
        n, Nf = 1 << CFG4["log2n"], CFG4["Nf"]
        def run(iters):
            y, t, f = synth_windows(1, n, Nf, "cpu")
            y, t = y.numpy(), t.numpy()
            t1 = time.time(); A, zf = o.get_fourier_regressor(t, f); Q, q = o.gram(A, y, np.ones(n)); t2 = time.time()
            r = o.admm_quadratic(Q, q, o.NormL1(CFG4["lam"]), iters=iters, tol=0.0, mu=CFG4["mu"])
            return dict(gram_s=t2 - t1, admm_s=time.time() - t2, iters=iters, cg=r.get("cg_iters"))
        probe = {}
        for T in cands:
            _set_threads(o, T); probe[T] = (lambda r: r["gram_s"] + r["admm_s"])(run(10))
        T = min(probe, key=probe.get); _set_threads(o, T)
        reps = [run(40) for _ in range(3)]
        rec.update(threads=T, probe_s={str(k): round(v, 3) for k, v in probe.items()},
                   best=min(reps, key=lambda r: r["gram_s"] + r["admm_s"]), all_s=[round(r["gram_s"] + r["admm_s"], 3) for r in reps])
    elif which == "cfg2":
        Nf = CFG2["Nf"]
        def run(lg, iters):
            y, t, f = [a.numpy() if hasattr(a, "numpy") else a for a in synth_fourier(1 << lg, Nf, "cpu")]
            t0 = time.time(); A, zf = o.get_fourier_regressor(t, f); t_asm = time.time() - t0
            t0 = time.time(); r = o.admm_ls(A, y, o.NormL1(CFG2["lam"]), iters=iters, tol=0.0, mu=CFG2["mu"]); t = time.time() - t0
            return dict(asm_s=t_asm, admm_s=t, iters=r["iters"], cg=r["cg_iters"])
        probe = {}
        for T in cands:
            _set_threads(o, T); run(13, 2); probe[T] = run(13, 2)["admm_s"]
        T = min(probe, key=probe.get); _set_threads(o, T)
        rec.update(threads=T, probe_s={str(k): round(v, 3) for k, v in probe.items()}, sizes={})
        for lg in (15, 16):
            reps = [run(lg, 6) for _ in range(3)]
            best = min(reps, key=lambda r: r["admm_s"])
            rec["sizes"][str(lg)] = dict(best, all_admm_s=[round(r["admm_s"], 3) for r in reps])
    elif which == "cfg5":
        Nf, Nv = CFG5["Nf"], CFG5["Nv"]
        def run(lg, iters):
            Y, X, V, w = [a.numpy() for a in synth_channels(1 << lg, Nf, 1, "cpu")]
            t0 = time.time(); Phi = o.lpv_regressor(X, V, w, Nv); t_asm = time.time() - t0
            t0 = time.time(); r = o.admm_ls(Phi, Y[:, 0], o.IndBallL0(CFG5["r"]), iters=iters, tol=0.0, mu=CFG5["mu"]); t = time.time() - t0
            return dict(asm_s=t_asm, admm_s=t, iters=r["iters"], cg=r["cg_iters"])
        probe = {}
        for T in cands:
            _set_threads(o, T); probe[T] = run(10, 1)["admm_s"]
        T = min(probe, key=probe.get); _set_threads(o, T)
        rec.update(threads=T, probe_s={str(k): round(v, 3) for k, v in probe.items()}, sizes={})
        for lg in (11, 12):                                    # 0.5 and 1 GiB of regressor at the full n = 32768
            reps = [run(lg, 2) for _ in range(2)]
            best = min(reps, key=lambda r: r["admm_s"])
            rec["sizes"][str(lg)] = dict(best, all_admm_s=[round(r["admm_s"], 3) for r in reps])
    print("CPU_BASELINE_JSON " + json.dumps(rec), flush=True)


def _run_cpu_child(which):
    env = dict(os.environ)
    env.update(OMP_PROC_BIND="spread", OMP_PLACES="cores", OMP_DYNAMIC="false")
    env.pop("OMP_NUM_THREADS", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", which], env=env, capture_output=True, text=True, timeout=900)
    line = next((l for l in out.stdout.splitlines() if l.startswith("CPU_BASELINE_JSON ")), None)
    if line is None:
        raise RuntimeError("CPU baseline child failed: " + out.stderr[-2000:])
    rec = json.loads(line[len("CPU_BASELINE_JSON "):])
    rec["wall_s"] = round(time.time() - t0, 1)
    return rec


def _linear_law(sizes, unit_rows):
    """seconds per CG iteration and sample row at each size: equal figures = the cost is a pure N-proportional stream"""
    return {f"N=2^{lg}": {"ns_per_cg_iteration_per_row": round(r["admm_s"] / max(r["cg"], 1) / (1 << int(lg)) * 1e9, 3),
                           "cg_iterations_per_admm_iteration": round(r["cg"] / r["iters"], 1), "admm_s_best_of": r["all_admm_s"]} for lg, r in sizes.items()}


def cpu_baseline():
    """cfg3 on the CPU port: the faithful form (dense Phi in memory, warm-started CG on the lazy Phi'Phi + I/mu, the extra Phi*x per
    iteration) at N_s = 2^13 and 2^14 rows of the full n = 8192; the cost per CG iteration is a pure N*n stream, so the larger sample
    is extrapolated linearly in N to 2^20 and in iterations to 2000 (optimistic for the CPU: CG needs more iterations as N grows)."""
    rec = _run_cpu_child("cfg3")
    big = rec["sizes"]["14"]
    scale = float(1 << (LOG2N - 14))
    per_iter = big["admm_s"] / big["iters"] * scale
    total = big["asm_s"] * scale + per_iter * ADMM_ITERS
    return {"value": 1.0 / total, "unit": "signals/s", "cores": rec["threads"], "kind": "port",
            "sample": f"N=2^14 rows at full n=8192, {big['iters']} ADMM iterations ({big['cg'] / big['iters']:.1f} CG iterations each, best of 3: "
                      f"{big['admm_s']:.1f} s) + regressor assembly ({big['asm_s']:.1f} s); extrapolated x{int(scale)} in N and to {ADMM_ITERS} iterations; "
                      f"{rec['threads']} pinned OpenMP threads (the fastest of {sorted(int(k) for k in rec['probe_s'])} on a probe; {rec['cpus_visible']} CPUs visible); "
                      f"{rec['wall_s']} s of CPU work in all",
            "admm_iters_per_sec": 1.0 / per_iter, "linear_in_N_check": _linear_law(rec["sizes"], None), "thread_probe_s": rec["probe_s"],
            # to rescale this figure to another host: the port is a memory-bound stream of Phi (twice per CG iteration: Phi x, Phi' r)
            "cpus_visible": rec["cpus_visible"], "threads_used": rec["threads"],
            "achieved_gemv_stream_GBps": 2.0 * (1 << 14) * (2 * NF * NV) * 8.0 * big["cg"] / big["admm_s"] * 1e-9,
            "bytes_streamed_per_cg_iteration_at_full_size": 2.0 * (1 << LOG2N) * (2 * NF * NV) * 8.0}


def cpu_baseline_cfg4():
    """cfg4 on the CPU port: one window of the full size (n = 2^16, Nreg = 511), regressor + explicit Gram (as the reference does for
    the weighted method, src/lasso.jl:118-120) + 40 CG-based ADMM iterations, best of 3, extrapolated linearly in iterations."""
    rec = _run_cpu_child("cfg4")
    b = rec["best"]
    per_window = b["gram_s"] + b["admm_s"] / b["iters"] * CFG4["iters"]
    return {"value": 1.0 / per_window, "unit": "windows/s", "cores": rec["threads"], "kind": "port",
            "sample": f"one window of 2^16 samples at Nf=256: regressor + Gram {b['gram_s']:.2f} s, {b['iters']} ADMM iterations {b['admm_s']:.2f} s "
                      f"(best of 3: {rec['all_s']}); extrapolated to {CFG4['iters']} iterations per window; {rec['threads']} pinned OpenMP threads "
                      f"({rec['cpus_visible']} CPUs visible); {rec['wall_s']} s of CPU work in all", "thread_probe_s": rec["probe_s"]}


def cpu_baseline_cfg2():
    rec = _run_cpu_child("cfg2")
    big = rec["sizes"]["16"]
    scale = float(1 << (CFG2["log2n"] - 16))
    per_iter = big["admm_s"] / big["iters"] * scale
    total = big["asm_s"] * scale + per_iter * CFG2["iters"]
    return {"value": 1.0 / total, "unit": "signals/s", "cores": rec["threads"], "kind": "port",
            "sample": f"N=2^16 rows at full n=1024, {big['iters']} ADMM iterations ({big['cg'] / big['iters']:.1f} CG iterations each, best of 3: "
                      f"{big['admm_s']:.2f} s) + regressor assembly ({big['asm_s']:.2f} s); extrapolated x{int(scale)} in N and to {CFG2['iters']} iterations; "
                      f"{rec['threads']} pinned OpenMP threads ({rec['cpus_visible']} CPUs visible); {rec['wall_s']} s of CPU work in all",
            "admm_iters_per_sec": 1.0 / per_iter, "linear_in_N_check": _linear_law(rec["sizes"], None), "thread_probe_s": rec["probe_s"]}


def cpu_baseline_cfg5():
    rec = _run_cpu_child("cfg5")
    big = rec["sizes"]["12"]
    scale = float(1 << (CFG5["log2n"] - 12))
    per_iter = big["admm_s"] / big["iters"] * scale
    total = big["asm_s"] * scale + per_iter * CFG5["iters"]          # per channel (the reference has no shared-regressor batch: every channel pays all of it)
    return {"value": 1.0 / total, "unit": "signals/s", "cores": rec["threads"], "kind": "port",
            "sample": f"ONE channel, N=2^12 rows at full n=32768, {big['iters']} ADMM iterations ({big['cg'] / big['iters']:.1f} CG iterations each, best of 2: "
                      f"{big['admm_s']:.1f} s) + regressor assembly ({big['asm_s']:.1f} s); extrapolated x{int(scale)} in N and to {CFG5['iters']} iterations "
                      f"(the regressor of the full size, 256 GiB, does not fit a host: SURVEY 8(a) a4); {rec['threads']} pinned OpenMP threads "
                      f"({rec['cpus_visible']} CPUs visible); {rec['wall_s']} s of CPU work in all",
            "admm_iters_per_sec": 1.0 / per_iter, "linear_in_N_check": _linear_law(rec["sizes"], None), "thread_probe_s": rec["probe_s"]}


def profiler_preloaded():
    """A rocprofv3 / roctracer preload has already initialised the GPU in THIS process before main() runs."""
    if any(k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_")) for k in os.environ):
        return True
    return any(tag in os.environ.get(v, "") for v in ("LD_PRELOAD", "HSA_TOOLS_LIB") for tag in ("rocprof", "roctracer", "rocprofiler"))


def spawn_ranks(args, argv):
    """`python3 bench.py --gpus N` outside a launcher: start the N ranks as a CHILD process (python -m torch.distributed.run, a plain
    subprocess of this one, which exits with its status) before this process has made any HIP call, so no GPU state is inherited.
    Under a profiler preload the GPU IS already initialised here, and starting another program from such a process is what this pool
    forbids: refuse, and profile single-rank runs (tools/collect_round.sh does)."""
    if profiler_preloaded():
        raise SystemExit("bench.py --gpus %d under a profiler preload: the profiler has initialised the GPU in this process, so it must not "
                         "start the ranks itself.  Profile a single-rank run (--gpus 1), or put the profiler inside each rank." % args.gpus)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 10 (cfg3) / 3 (cfg4)")
    ap.add_argument("--warmup", type=int, default=None, help="default 2 (cfg3) / 1 (cfg4)")
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg4", "cfg2", "cfg5"],
                    help="cfg3 (default, the judged line): one LPV group-lasso signal per GPU per step (+ the cfg4_strong sub-record); cfg4: 1024 "
                         "batched windows per step, window range sharded over the ranks; cfg2: ls_sparse_spectral NormL1 N=2^18 Nf=512, 5000 "
                         "iterations; cfg5: multichannel LPV n=32768 IndBallL0(32), 8 channels per GPU")
    ap.add_argument("--no-cfg4-strong", action="store_true", help="cfg3 runs: skip the cfg4 strong-scaling sub-record")
    ap.add_argument("--no-baseline-configs", action="store_true", help="cfg3 runs: skip the compact cfg2 / cfg5 records")
    ap.add_argument("--no-row-sharded", action="store_true", help="cfg3 runs with several ranks: skip the row-sharded-Gram sub-record (SURVEY 8(e)(2))")
    ap.add_argument("--no-single-process", action="store_true", help="cfg3 runs: skip the sub-records in which rank 0 alone drives all devices through the C-ABI's several-device drivers")
    ap.add_argument("--rehearse-sub-records", action="store_true", help="rehearsals (tests): run the sub-records (cfg4_strong, single_process) at the reduced sizes of the diagnostic flags too")
    ap.add_argument("--sub-timeout", type=int, default=600, help="seconds the sub-records (after the main record is complete) may take before a watchdog ends the run with the main line")
    ap.add_argument("--watchdog-exit-code", type=int, default=0, help="exit status when the watchdog ended a hung sub-record phase AFTER the main line was printed "
                    "(default 0: the line is valid and a driver may discard the output of a failed command; the hang is on the line as sub_records_note)")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--channels", type=int, default=CFG5["channels_per_gpu"], help="cfg5: channels per GPU (configured: 8)")
    ap.add_argument("--log2n", type=int, default=LOG2N, help="diagnostic only; the judged size is 20")
    ap.add_argument("--iters", type=int, default=None, help="ADMM iterations (default 2000 for both workloads)")
    ap.add_argument("--nwin", type=int, default=CFG4["nwin"], help="cfg4 diagnostic only; the configured count is 1024")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-general-path", action="store_true", help="skip the extra (untimed) dense-MFMA Gram measurement")
    ap.add_argument("--no-alt-storage", action="store_true",
                    help="skip the extra (untimed) mat-vec / whole-step measurements with the other storages of the inverse (profiling runs)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI, the judged path) or gloo (functional rehearsal)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="f32: float32 inputs through the _f32 entry points (double assembly, single-precision copy of M streamed by "
                         "the ADMM mat-vec); the judged line is f64")
    ap.add_argument("--streams", type=int, default=1,
                    help="host threads per GPU solving independent signals concurrently (each handle owns its HIP stream): the "
                         "VALU/MFMA-bound Gram and factorisation of one solve overlap the HBM-bound iterations of another; "
                         "the judged line uses 1")
    ap.add_argument("--no-concurrent", action="store_true",
                    help="skip the extra measurement with two solves in flight per GPU (the `two_solves_in_flight` sub-record; profiling runs)")
    ap.add_argument("--row-sharded", action="store_true",
                    help="strong-scaling variant: one signal per step, its sample rows sharded over the ranks (one all-reduce of the Gram)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: map every rank onto the visible GPUs modulo their count")
    args = ap.parse_args()
    if args.cpu_baseline_child:                           # the CPU port in a child of its own (pinned OpenMP threads); never touches the GPU
        _cpu_child(args.cpu_baseline_child)
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))            # before any torch.cuda / library call in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.row_sharded and (world == 1 or args.workload != "cfg3"):
        raise SystemExit("--row-sharded needs --gpus > 1 and the cfg3 workload")
    import lpvspectral_jl_amd as L   # loads torch's HIP runtime first, then liblpvspectral.so
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if args.share_gpu:
        local = local % ndev
    elif local >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local} but only {ndev} GPU(s) visible (use --share-gpu for a rehearsal)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collective buffers live

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def max_over_ranks(elapsed):
        if dist is None:
            return elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- the main record.  Everything after it is a sub-record that can no longer cost it: local failures become {"error": ...}
    # (guarded / measure_cfg4), and should a rank die or a collective hang in a sub-record, a watchdog on every rank ends the run
    # with rank 0 printing the line it already holds.
    state = {"out": None, "printed": False}
    import threading
    plock = threading.Lock()

    def emit(note=None):
        with plock:
            if rank != 0 or state["printed"] or state["out"] is None:
                return
            o = state["out"]
            o.pop("_params_rank0", None)
            if note:
                o["sub_records_note"] = note
                o.setdefault("config", {})["sub_records_note"] = note[:136]
            o["backend"] = "none (single process)" if dist is None else ("rccl (torch.distributed nccl)" if args.backend == "nccl" else args.backend)
            o["rccl_ranks"] = world if (dist is not None and args.backend == "nccl") else 0
            o["collective_ranks"] = world
            driver_record(o)
            print(json.dumps(o), flush=True)
            state["printed"] = True

    def watchdog():
        sys.stderr.write("[bench] rank %d: sub-records exceeded %d s -- ending the run with the main line\n" % (rank, args.sub_timeout))
        emit("aborted by the watchdog after %d s (a rank died or a collective hung in a sub-record); the main record was complete" % args.sub_timeout)
        sys.stdout.flush(); sys.stderr.flush()
        # the line is out (rank 0) and carries the hang as sub_records_note; --watchdog-exit-code N additionally reports it through the exit
        # status (default 0: a driver that discards the output of a failed command would lose the valid main record)
        os._exit(args.watchdog_exit_code if (rank != 0 or state["printed"]) else 1)

    timer = None
    if args.workload == "cfg4":
        state["out"] = run_cfg4(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks)
    elif args.workload == "cfg2":
        state["out"] = run_cfg2(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks)
    elif args.workload == "cfg5":
        state["out"] = run_cfg5(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks)
    else:
        state["out"] = run_cfg3(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks)
        timer = threading.Timer(args.sub_timeout, watchdog)
        timer.daemon = True
        timer.start()
        try:
            full_size = not args.row_sharded and args.log2n == LOG2N
            if not args.no_cfg4_strong and (full_size or args.rehearse_sub_records):
                # the batched-window configuration, window range sharded over the same ranks (north_star: "near-linear 1->8 on batched windows")
                sub = measure_cfg4(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks, steps=3 if full_size else 1, warmup=1,
                                   iters=CFG4["iters"] if full_size else (args.iters or 80), nwin=CFG4["nwin"] if full_size else args.nwin)
                if rank == 0:
                    state["out"]["cfg4_strong"] = sub
            if not args.no_baseline_configs and (full_size or args.rehearse_sub_records):
                # BASELINE.json's other two GPU configurations as compact records of the same line (one signal / 8 channels per rank; with
                # their own roofline and, at one GPU, cpu_baseline): the driver's fixed command witnesses all four configurations
                s2 = dict(steps=5, warmup=1, iters=CFG2["iters"]) if full_size else dict(steps=1, warmup=1, iters=args.iters or 100)
                sub = run_cfg2(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks, sub=s2)
                if rank == 0:
                    state["out"]["cfg2"] = sub
                s5 = (dict(steps=1, warmup=1, iters=CFG5["iters"], log2n=CFG5["log2n"]) if full_size else
                      dict(steps=1, warmup=0, iters=args.iters or 30, log2n=min(args.log2n, 16), Nf=256, Nv=4, channels=2))
                sub = run_cfg5(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks, sub=s5)
                torch.cuda.empty_cache()
                L._lib.lib().lpvs_release_cached_memory()
                if rank == 0:
                    state["out"]["cfg5"] = sub
            if world > 1 and not args.no_row_sharded and (full_size or args.rehearse_sub_records):
                sub = measure_rowsharded(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks, steps=2 if full_size else 1, warmup=1,
                                         iters=ADMM_ITERS if full_size else (args.iters or 80), log2n=args.log2n)
                if rank == 0:
                    state["out"]["cfg3_row_sharded"] = sub
            if not args.no_single_process and (full_size or args.rehearse_sub_records) and not profiler_preloaded():
                # ---- ONE host process driving ALL devices through the C-ABI's several-device drivers (what a Julia host calls).  The other
                # ranks release their cached device memory and wait on a HOST barrier (a gloo group: no kernel spinning on their GPUs).
                hostgrp = None
                if dist is not None:
                    import datetime
                    hostgrp = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=args.sub_timeout + 60))
                torch.cuda.empty_cache()
                L._lib.lib().lpvs_release_cached_memory()
                if hostgrp is not None:
                    dist.barrier(group=hostgrp)
                if rank == 0:
                    devs = [r % ndev for r in range(world)] if args.share_gpu else list(range(min(world, ndev)))   # the devices this run's ranks own, no others
                    ref = state["out"].pop("_params_rank0", None)
                    kw = dict(reference_params=ref if full_size else None)
                    if not full_size:                                   # rehearsals (tests): sizes follow the diagnostic flags
                        kw.update(log2n3=args.log2n, iters3=args.iters or ADMM_ITERS, iters4=args.iters or 80, nwin4=min(args.nwin, 64), log2n4=12,
                                  iters5=args.iters or 30, log2n5=min(args.log2n, 16), nf5=64, nv5=4, channels5=2)
                    if args.share_gpu and len(set(devs)) < len(devs):
                        os.environ["LPVS_MULTI_ALLOW_SHARED_DEVICE"] = "1"
                    state["out"]["single_process"] = guarded(lambda: single_process_records(L, devs, **kw), "single_process")
                if hostgrp is not None:
                    dist.barrier(group=hostgrp)
        except (KeyboardInterrupt, SystemExit):
            raise
        except BaseException as e:                                       # noqa: BLE001 (a collective of a sub-record raised on THIS rank)
            import traceback
            sys.stderr.write("[bench] rank %d: sub-record phase failed:\n%s\n" % (rank, traceback.format_exc()))
            if rank == 0:
                state["out"].setdefault("cfg4_strong", {"error": ("%s: %s" % (type(e).__name__, e))[:600]})
            else:
                time.sleep(args.sub_timeout + 5)                           # stay alive (the launcher kills every rank when one exits non-zero): the watchdog ends this rank
        finally:
            if rank == 0:
                state["out"].pop("_params_rank0", None)
            emit()
    emit()
    if dist is not None:
        # (the watchdog stays armed across the destroy: after a failed collective it can block on peers that are asleep or hung)
        try:
            dist.destroy_process_group()
        except Exception:                                                # noqa: BLE001
            pass
    if timer is not None:
        timer.cancel()


def measure_cfg4(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks, steps, warmup, iters, nwin):
    """cfg4, the window range sharded over the ranks (strong scaling): the compact record that rides on the cfg3 line.
    The ranks stay in lock-step whatever happens on one of them: a rank whose LOCAL part (allocation, engine call) raises still
    takes part in every collective with a zero contribution, the failure flag is all-reduced after the loop, and the record then
    reads {"error": ...} on rank 0 -- the cfg3 line this rides on is never at stake (main() prints it in any case)."""
    n, Nf = 1 << CFG4["log2n"], CFG4["Nf"]
    lo, hi = L.sharding.shard_range(nwin, world, rank)
    failed = []
    try:
        y, t, f = synth_windows(nwin, n, Nf, dev)
    except Exception as e:                                     # noqa: BLE001
        failed.append("%s: %s" % (type(e).__name__, e)); y = t = None; f = np.arange(Nf) / (2.0 * Nf)

    def step():
        x, its = np.zeros((hi - lo, Nf), dtype=np.complex128), np.zeros(hi - lo, dtype=np.int64)
        inject = os.environ.get("LPVS_BENCH_INJECT")                  # tests/test_gpu_bench_spawn.py: "fail:<rank>" / "hang:<rank>" in this sub-record
        if inject == "hang:%d" % rank:
            time.sleep(10 ** 6)
        if not failed:
            try:
                if inject == "fail:%d" % rank:
                    raise RuntimeError("injected failure in rank %d's local part of cfg4_strong" % rank)
                x, S_part, its = L.windowpsd_sparse_batched(y, t, f, n, 0, None, λ=CFG4["lam"], μ=CFG4["mu"], tol=0.0, iters=iters,
                                                            win_lo=lo, win_hi=hi, device=local)
            except Exception as e:                             # noqa: BLE001
                failed.append("%s: %s" % (type(e).__name__, e))
                x, its = np.zeros((hi - lo, Nf), dtype=np.complex128), np.zeros(hi - lo, dtype=np.int64)
        full = L.sharding.gather_units(x, nwin, dist, cdev)   # ONE all_gather of the per-window coefficients (RCCL)
        return L.sharding.reduce_psd_in_order(full), its

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        S, its = step()
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    anyfail = max_over_ranks(1.0 if failed else 0.0) > 0      # (an all-reduce MAX: every rank learns of any rank's failure)
    tm = L.windowpsd_last_timing() if not failed else {}
    # the dominant kernel's bytes per iteration of this rank's shard (one untimed call of 8 iterations that appends 200 stand-alone batch mat-vec
    # launches and reports the bytes of packed inverses a launch of the iteration reads); local, no collective
    roof = None
    if rank == 0 and not failed and tm.get("one_launch_iteration") and iters > 0:
        def _roof():
            os.environ["LPVS_WINDOW_MATVEC_TIMING"] = "1"
            try:
                L.windowpsd_sparse_batched(y, t, f, n, 0, None, λ=CFG4["lam"], μ=CFG4["mu"], tol=0.0, iters=8, win_lo=lo, win_hi=hi, device=local)
                tm_mv = L.windowpsd_last_timing()
            finally:
                del os.environ["LPVS_WINDOW_MATVEC_TIMING"]
            it_us = tm["solve_ms"] * 1e3 / iters              # HIP events around the ADMM loops of the last timed step / iterations: all windows of the shard
            ach = tm_mv["matvec_bytes_per_launch"] / (it_us * 1e-6) * 1e-9
            return {"bound": "hbm", "kernel": "admm_iter_mixed_kernel<batch>", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", **roof_fracs(ach),
                    "algorithmic_bytes_per_launch": tm_mv["matvec_bytes_per_launch"], "launch_us": it_us, "windows_per_launch": hi - lo, "launches_per_step": iters,
                    "reads_32_bits": bool(tm.get("reads_32_bits")),
                    "note": "launch_us = one ADMM iteration of ALL the shard's windows (cache-sized chunks, two halves in flight: DESIGN 4.5.3); the chunks' "
                            "inverses are re-read from the Infinity Cache, so `achieved` may exceed the HBM stream rate"}
        roof = guarded(_roof, "cfg4_strong.roofline")
    del y, t
    torch.cuda.empty_cache()
    L._lib.lib().lpvs_release_cached_memory()
    if rank != 0:
        return None
    if anyfail:
        return {"error": failed[0] if failed else "a rank other than 0 failed in its local part (see its stderr)", "n_gpus": world}
    return {"metric": "windows/sec, ls_windowpsd estimator=ls_sparse_spectral (NormL1), %d windows x N=2^%d, Nf=%d, %d ADMM iters per window"
                      % (nwin, CFG4["log2n"], Nf, iters),
            "value": nwin * steps / elapsed, "unit": "windows/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "scaling": "strong", "windows_per_gpu": [L.sharding.shard_range(nwin, world, r)[1] - L.sharding.shard_range(nwin, world, r)[0] for r in range(world)],
            "final_gather": "none" if world == 1 else ("rccl" if args.backend == "nccl" else args.backend) + " all_gather of per-window coefficients, PSD summed in window order",
            "one_launch_iteration": bool(tm.get("one_launch_iteration")), "gram_form": tm.get("gram_form"),
            "phase_ms_rank0": {k: v for k, v in tm.items() if k.endswith("_ms")}, "psd_argmax": int(np.argmax(S)),
            "iters_min_max": [int(its.min()), int(its.max())], "roofline": roof}


def measure_rowsharded(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks, steps, warmup, iters, log2n):
    """SURVEY 8(e)(2) as a sub-record of the default line: ONE cfg3 signal, its sample rows sharded over the ranks, partial Grams summed
    by one all-reduce of the device Gram (RCCL), ADMM replicated.  Lock-step like measure_cfg4: a rank whose local part fails still
    enters the three collectives of a step (ranges, G, b) with neutral contributions."""
    N, n_p = 1 << log2n, 2 * NF * NV
    failed = []
    ys = Xs = Vs = w = None
    try:
        y, X, V, w = synth_signal(N, NF, 0, dev)              # every rank generates the same signal and keeps its rows
        lo, hi = L.sharding.shard_range(N, world, rank)
        ys, Xs, Vs = y[lo:hi].contiguous(), X[lo:hi].contiguous(), V[lo:hi].contiguous()
        del y, X, V
    except Exception as e:                                     # noqa: BLE001
        failed.append("%s: %s" % (type(e).__name__, e))
    t_ex = []

    def neutral():
        L.sharding.allreduce_sum_(torch.zeros((n_p, n_p), dtype=torch.float64, device=dev), dist)
        L.sharding.allreduce_sum_(torch.zeros((1, n_p), dtype=torch.float64, device=dev), dist)

    def step():
        r4 = None
        if not failed:
            try:
                r4 = L.lpv_ranges(Xs, Vs)
            except Exception as e:                             # noqa: BLE001
                failed.append("%s: %s" % (type(e).__name__, e))
        ranges = L.sharding.allreduce_ranges(r4 if r4 is not None else np.array([np.inf, -np.inf, 0.0, 0.0]), dist)
        if failed:
            neutral(); return None
        try:
            p = L.Problem.lpv_rows(ys, Xs, Vs, w, NV, ranges, True, False, device=local)
        except Exception as e:                                 # noqa: BLE001
            failed.append("%s: %s" % (type(e).__name__, e)); neutral(); return None
        with p:
            G, b = p.device_gram()
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            L.sharding.allreduce_sum_(G, dist)
            L.sharding.allreduce_sum_(b, dist)
            torch.cuda.synchronize(dev); t_ex.append(time.perf_counter() - t1)
            try:
                p.gram_modified()
                p.set_prox(L.SlicedSeparableSum.frequency_groups(LAMBDA, NF, 2 * NV))
                p.admm_init(None, μ=MU, tol=0.0)
                it, nxz, conv = p.admm_run(iters)
                return p.params(0), it, nxz
            except Exception as e:                             # noqa: BLE001
                failed.append("%s: %s" % (type(e).__name__, e)); return None

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    res = None
    for _ in range(steps):
        res = step()
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    anyfail = max_over_ranks(1.0 if failed else 0.0) > 0
    del ys, Xs, Vs
    torch.cuda.empty_cache()
    L._lib.lib().lpvs_release_cached_memory()
    if rank != 0:
        return None
    if anyfail:
        return {"error": failed[0] if failed else "a rank other than 0 failed in its local part (see its stderr)", "n_gpus": world}
    return {"metric": "signals/sec, ONE ls_sparse_spectral_lpv signal N=2^%d, sample rows sharded over the ranks (%d ADMM iters)" % (log2n, iters),
            "value": steps / elapsed, "unit": "signals/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "scaling": "strong", "exchange": "one all-reduce of the device Gram (%d x %d f64) + rhs, %s" % (n_p, n_p, "rccl" if args.backend == "nccl" else args.backend),
            "allreduce_ms_rank0": float(np.mean(t_ex[-steps:])) * 1e3 if t_ex else None, "iters": int(res[1]) if res else None,
            "note": "the Gram of this workload is 2 ms of a 72 ms step (structured form), so row sharding cannot speed it up: the record exists so that SURVEY "
                    "8(e)(2)'s exchange step runs on hardware; it pays for arbitrary-w problems, whose dense MFMA Gram is 0.7 s"}


# ---------------------------------------------------------------------------------------------------------------- one host process, all devices
def single_process_records(L, devices, iters3=ADMM_ITERS, log2n3=LOG2N, nf3=NF, iters4=None, nwin4=None, log2n4=None, iters5=None, log2n5=None,
                           nf5=None, nv5=None, channels5=None, gen_device=None, which=("cfg3", "cfg4", "cfg5"), reference_params=None):
    """north_star's process model ("one process -- the Julia host -- driving all 8 devices through the shim"): THIS process
    alone drives every device of `devices` through the C-ABI's several-device drivers, exactly as julia/LPVSpectralAMD.jl would:
      cfg3  lpvs_lpv_signals_multi_f64      2 signals per device, own (X, V) each, in_flight = 2, no collective
      cfg4  lpvs_windows_estimate_multi_f64 all windows, contiguous ranges per device, the library's own (dlopen'ed) RCCL all-gather
      cfg5  lpvs_lpv_batch_multi_f64        `channels5` channels per device sharing (X, V), one Gram / factorisation per device
    Inputs are HOST arrays (what a host-language caller holds), so these rates include the uploads -- unlike the per-rank lines, whose
    inputs are resident.  Each record is guarded on its own."""
    nd = len(devices)
    gdev = gen_device if gen_device is not None else torch.device("cuda", devices[0])
    iters4 = CFG4["iters"] if iters4 is None else iters4; nwin4 = CFG4["nwin"] if nwin4 is None else nwin4; log2n4 = CFG4["log2n"] if log2n4 is None else log2n4
    iters5 = CFG5["iters"] if iters5 is None else iters5; log2n5 = CFG5["log2n"] if log2n5 is None else log2n5
    nf5 = CFG5["Nf"] if nf5 is None else nf5; nv5 = CFG5["Nv"] if nv5 is None else nv5; channels5 = CFG5["channels_per_gpu"] if channels5 is None else channels5
    out = {"n_devices": nd, "devices": list(devices), "host_process": "one (pid %d); inputs are host arrays, uploads inside the timed calls" % os.getpid()}

    def free():
        torch.cuda.empty_cache()
        L._lib.lib().lpvs_release_cached_memory()

    def rec3():
        nsig = 2 * nd
        cols = [[a.cpu().numpy() for a in synth_signal(1 << log2n3, nf3, q // 2, gdev)] for q in range(nsig)]   # device r's two signals = rank r's signal
        w = cols[0][3]
        Y, X, V = (np.asfortranarray(np.stack([c[i] for c in cols], axis=1)) for i in range(3))
        prox = L.SlicedSeparableSum.frequency_groups(LAMBDA, nf3, 2 * NV)
        call = lambda: L.lpv_signals_multi(Y, X, V, w, NV, proxg=prox, μ=MU, tol=0.0, iters=iters3, devices=list(devices), in_flight=2)
        call()
        t0 = time.perf_counter(); P, its = call(); e = time.perf_counter() - t0
        r = {"entry_point": "lpvs_lpv_signals_multi_f64", "value": nsig / e, "unit": "signals/s", "signals": nsig, "in_flight_per_device": 2, "ms_per_call": e * 1e3,
             "iters_min_max": [int(its.min()), int(its.max())], "collective": "none"}
        if reference_params is not None:
            r["same_coefficients_as_the_timed_steps"] = bool(np.array_equal(P[:, 0], np.asarray(reference_params).ravel()))
        return r

    def rec4():
        n, Nf = 1 << log2n4, CFG4["Nf"]
        y, t, f = synth_windows(nwin4, n, Nf, gdev)
        yh, th = y.cpu().numpy(), t.cpu().numpy()
        del y, t
        free()
        eng = dict(estimator=L._lib.EST_SPARSE, lam=0.0, prox=(L._lib.PROX_L1, CFG4["lam"], 0), μ=CFG4["mu"], tol=0.0, iters=iters4, sign=L._lib.LINEAR_QUADRATIC_AS_WRITTEN)
        call = lambda: L.windows_estimate_multi([yh], th, f, n, 0, None, eng, devices=list(devices))
        call()
        t0 = time.perf_counter(); x, its = call(); e = time.perf_counter() - t0
        tm = L.windowpsd_last_timing()
        S = L.sharding.reduce_psd_in_order(x[0])
        return {"entry_point": "lpvs_windows_estimate_multi_f64", "value": nwin4 / e, "unit": "windows/s", "windows": nwin4, "ms_per_call": e * 1e3,
                "collective": "the library's own RCCL all-gather (librccl.so.1 by dlopen), %d-rank communicator" % tm["rccl_gather_ranks"] if tm["rccl_gather_ranks"] else ("none (one device)" if len(set(devices)) == len(devices) and nd == 1 else "none (shards share a device: the shard images are taken from the host staging)"),
                "rccl_gather_ranks": tm["rccl_gather_ranks"], "psd_argmax": int(np.argmax(S)), "iters_min_max": [int(its.min()), int(its.max())]}

    def rec5():
        ns = channels5 * nd
        Y, X, V, w = synth_channels(1 << log2n5, nf5, ns, gdev)
        Yh, Xh, Vh, wh = (a.cpu().numpy() for a in (Y, X, V, w))
        del Y, X, V
        free()
        call = lambda: L.lpv_batch_multi(Yh, Xh, Vh, wh, nv5, proxg=L.IndBallL0(CFG5["r"]), μ=CFG5["mu"], tol=0.0, iters=iters5, devices=list(devices))
        if nd > 1:
            call()                                              # (one device: the cfg5 workload's own warm-up is minutes of budget; a cold call is what is timed)
        t0 = time.perf_counter(); P, its = call(); e = time.perf_counter() - t0
        return {"entry_point": "lpvs_lpv_batch_multi_f64", "value": ns / e, "unit": "signals/s", "channels": ns, "channels_per_device": channels5, "ms_per_call": e * 1e3,
                "warm": nd > 1, "nnz_per_channel_min_max": [int(np.count_nonzero(P, axis=0).min()), int(np.count_nonzero(P, axis=0).max())],
                "iters_min_max": [int(its.min()), int(its.max())], "collective": "none"}

    for name, fn in (("cfg3", rec3), ("cfg4", rec4), ("cfg5", rec5)):
        if name in which:
            out["single_process_" + name] = guarded(fn, "single_process_" + name)
            guarded(free, "release")
    return out


def run_cfg4(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks):
    steps = 3 if args.steps is None else args.steps
    warmup = 1 if args.warmup is None else args.warmup
    iters = CFG4["iters"] if args.iters is None else args.iters
    nwin, n, Nf = args.nwin, 1 << CFG4["log2n"], CFG4["Nf"]
    y, t, f = synth_windows(nwin, n, Nf, dev)                # resident in HBM before the timed region (every rank holds the record)
    lo, hi = L.sharding.shard_range(nwin, world, rank)       # this rank's windows: src/lsfft.jl:120 is the loop being sharded

    def step():
        x, S_part, its = L.windowpsd_sparse_batched(y, t, f, n, 0, None, λ=CFG4["lam"], μ=CFG4["mu"], tol=0.0, iters=iters,
                                                    win_lo=lo, win_hi=hi, device=local)
        full = L.sharding.gather_units(x, nwin, dist, cdev)   # ONE all_gather of the per-window coefficients (RCCL)
        S = L.sharding.reduce_psd_in_order(full)              # S .+= abs2.(x) in window order, / k^2   src/lsfft.jl:122,125
        return S, its

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        S, its = step()
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    tm = L.windowpsd_last_timing()
    # dominant kernel of the step (the batch mat-vec, one launch per iteration for all windows of the shard): its launch time from
    # one extra UNTIMED call that appends 200 back-to-back launches after its last pass (HIP events on the library's stream)
    os.environ["LPVS_WINDOW_MATVEC_TIMING"] = "1"
    try:
        L.windowpsd_sparse_batched(y, t, f, n, 0, None, λ=CFG4["lam"], μ=CFG4["mu"], tol=0.0, iters=8, win_lo=lo, win_hi=hi, device=local)
        tm_mv = L.windowpsd_last_timing()
    finally:
        del os.environ["LPVS_WINDOW_MATVEC_TIMING"]
    if dist is not None:
        dist.barrier()
    if rank != 0:
        return None
    tm["matvec_us_per_iteration"] = tm_mv.get("matvec_us_per_iteration")
    tm["matvec_windows"] = tm_mv.get("matvec_windows")
    st = L.get_default_option("storage") or os.environ.get("LPVS_M_STORAGE", "mixed")   # storage of the packed inverses (admm.hip)
    mv_us = tm.get("matvec_us_per_iteration")
    one_launch = bool(tm_mv.get("one_launch_iteration"))
    mv_only_us = mv_us
    if one_launch and mv_us:
        # the iteration IS one launch of the batch kernel: its duration inside the timed region = HIP events around the ADMM loops of
        # the last timed step / launches (every 256 iterations one extra launch without an update and one update-only launch)
        mv_us = tm["solve_ms"] * 1e3 / iters
    roof = None
    if mv_us:
        nmv = tm.get("matvec_windows") or (hi - lo)
        mv_bytes = tm_mv.get("matvec_bytes_per_launch") or nmv * (8 if st == "f64" else 6) * (512 * (512 + 128) // 2)
        achieved = mv_bytes / (mv_us * 1e-6) * 1e-9
        kern = {"f64": "symv_tile_batch_kernel (8-byte elements)", "split": "symv_tile_split_batch_kernel (6-byte elements)"}.get(
            st, ("admm_iter_mixed_kernel<batch> (the whole ADMM iteration of every window in one launch; 36-bit fixed-point tiles" +
                 (" of which the launch reads 32 bits -- the 4-bit planes ride, up to 32 iterations stale, in the offset vectors" if tm.get("reads_32_bits") else "") +
                 ", tile partials added into x by 64-bit fixed-point atomics)") if one_launch else
                "symv_tile_mixed_batch_kernel (6-byte float-head tiles on the diagonal, 36-bit fixed-point tiles elsewhere)")
        roof = {"bound": "hbm", "kernel": kern + ": one tile-packed (Q + I/mu)^-1 per window, all windows of the shard per launch",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", **roof_fracs(achieved),
                **dict(zip(("traffic_of_one_part_launch", "traffic_source"), pmc_traffic("admm_iter_mixed_kernel<1", "cfg4") if (one_launch and nwin == CFG4["nwin"]) else (None, None))),
                "traffic": None,
                "algorithmic_bytes_per_launch": mv_bytes, "launch_us": mv_us, "windows_per_launch": nmv, "launches_per_step": iters,
                "matvec_only_launch_us": mv_only_us,
                "note": ("launch_us = the time one ADMM iteration of ALL the shard's windows takes = HIP events around the ADMM loops of the last timed "
                         "step / iterations; the engine runs the shard in chunks of ~285 MB of packed inverses (384 windows here), two halves of a chunk "
                         "in flight on two streams, one launch per iteration and half: a chunk's inverses are re-read from the 256 MiB Infinity Cache "
                         "iteration after iteration, which is why `achieved` may exceed the HBM stream rate (LPVS_WINDOW_CHUNK_MB=0 "
                         "LPVS_WINDOWS_IN_FLIGHT=1: every window in one launch per iteration, HBM-bound: 126 us = 6.06 TB/s); "
                         "matvec_only_launch_us = 200 back-to-back launches of the two-launch scheme's stand-alone batch mat-vec over ALL the shard's windows in one launch (uncut: HBM-bound); "
                         "`traffic` is null on purpose: an iteration of the shard is SEVERAL launches under the chunked plan (one per chunk half), whose fabric bytes "
                         "the PMC summary reports per launch (traffic_of_one_part_launch, ~171 windows each) -- mostly served by the Infinity Cache, not HBM") if one_launch else
                        "HIP events around 200 back-to-back launches of the batch mat-vec on the library's stream (rank 0's shard)"}
    out = {
        "metric": "windows/sec, ls_windowpsd estimator=ls_sparse_spectral (NormL1), %d windows x N=2^%d, Nf=%d, %d ADMM iters per window"
                  % (nwin, CFG4["log2n"], Nf, iters),
        "value": nwin * steps / elapsed, "unit": "windows/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "cfg4: ls_windowpsd(estimator=ls_sparse_spectral) %d windows x 2^%d samples, Nf=%d (zero frequency first, Nreg=%d), "
                               "NormL1(%g), mu=%g, iters=%d, tol=0" % (nwin, CFG4["log2n"], Nf, 2 * Nf - 1, CFG4["lam"], CFG4["mu"], iters),
                   "windows_per_gpu": [L.sharding.shard_range(nwin, world, r)[1] - L.sharding.shard_range(nwin, world, r)[0] for r in range(world)],
                   "sharding": "contiguous window ranges per rank, no data-path collective",
                   "final_gather": "none" if world == 1 else ("rccl" if args.backend == "nccl" else args.backend) + " all_gather of per-window coefficients, PSD summed in window order",
                   "gram": {"ap-nufft": "structured, slot sums by a non-uniform FFT per window (nufft.hip); MFMA path not taken",
                            "ap": "structured (VALU f64, nudft.hip); MFMA path not taken"}.get(tm.get("gram_form"), "dense f64 MFMA (panel form)")},
        "admm_iters_per_sec": nwin * iters * steps / elapsed,
        "phase_ms_rank0": {k: v for k, v in tm.items() if k.endswith("_ms")},
        "psd_argmax": int(np.argmax(S)), "iters_min_max": [int(its.min()), int(its.max())],
        "roofline": roof,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = guarded(cpu_baseline_cfg4, "cpu_baseline")
    return out


def run_cfg3(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks):
    steps = 10 if args.steps is None else args.steps
    warmup = 2 if args.warmup is None else args.warmup
    iters = ADMM_ITERS if args.iters is None else args.iters
    N = 1 << args.log2n
    y, X, V, w = synth_signal(N, NF, 0 if args.row_sharded else rank, dev)   # inputs resident in HBM before the timed region
    if args.dtype == "f32":
        y, X, V, w = (a.to(torch.float32) for a in (y, X, V, w))
    rowsh = args.row_sharded and world > 1
    if rowsh:
        lo, hi = L.sharding.shard_range(N, world, rank)
        y, X, V = y[lo:hi].contiguous(), X[lo:hi].contiguous(), V[lo:hi].contiguous()
    run = (lambda: solve_rowsharded(L, y, X, V, w, NV, iters, local, dist)) if rowsh else (lambda: solve(L, y, X, V, w, NV, iters, local))

    for _ in range(warmup):
        run()
    sync()
    t0 = time.perf_counter()
    tms, params = [], None
    def run_threads(nthreads, nsteps):
        import threading
        res, lock = [], threading.Lock()
        counter = iter(range(nsteps))
        def worker():
            while True:
                with lock:
                    k = next(counter, None)
                if k is None:
                    return
                r = run()                      # ctypes releases the GIL inside the library calls
                with lock:
                    res.append(r)
        th = [threading.Thread(target=worker) for _ in range(nthreads)]
        [t_.start() for t_ in th]
        [t_.join() for t_ in th]
        return res
    if args.streams > 1 and not rowsh:
        for params, it, nxz, tm in run_threads(args.streams, steps):
            tms.append(tm)
    else:
        for _ in range(steps):
            params, it, nxz, tm = run()
            tms.append(tm)
    if dist is not None and not rowsh:                   # final gather of the coefficients (RCCL)
        mine = torch.view_as_real(torch.tensor(params, device=cdev)).contiguous()
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if rank != 0:
        return None

    phase = {k: float(np.mean([t[k] for t in tms])) for k in ("basis_ms", "gram_ms", "reduce_rhs_ms", "factor_ms", "admm_ms", "xcorr_ms")}
    n_xcorr = int(tms[0].get("xcorr_count", 0))
    n_nib = int(tms[0].get("nibble_refreshes", 0))
    form = tms[0]["gram_form"]
    # ---- the same signals with TWO solves in flight per GPU (two host threads, each handle its own stream): the matrix-core-bound
    # factorisation of one solve runs under the HBM-bound iterations of the other.  Reported beside the judged line (whose `value`,
    # ms_per_step and roofline are the one-solve-at-a-time figures above); not part of the timed region.
    def _two_in_flight():
        nst = max(4, steps - steps % 2)
        run_threads(2, 2)
        sync()
        t2 = time.perf_counter()
        res2 = run_threads(2, nst)
        sync()
        e2 = time.perf_counter() - t2
        two = {"value": nst / e2, "unit": "signals/s", "steps": nst, "ms_per_signal": e2 / nst * 1e3, "host_threads": 2,
               "admm_launch_us_under_contention": float(np.mean([r[3]["admm_ms"] for r in res2])) * 1e3 / iters}
        if args.dtype == "f64":
            # ... and the same through ONE call of the C-ABI's batch driver (lpvs_lpv_signals_multi_f64: a single-threaded host, the
            # library's own worker threads), device-resident column-major inputs
            def through():
                col = lambda v: torch.stack([v] * nst, dim=0).T
                Ym, Xm, Vm = col(y), col(X), col(V)
                prox = L.SlicedSeparableSum.frequency_groups(LAMBDA, len(w), 2 * NV)
                sync()
                t3 = time.perf_counter()
                Pm, itm = L.lpv_signals_multi(Ym, Xm, Vm, w, NV, proxg=prox, μ=MU, tol=0.0, iters=iters, devices=[local], in_flight=2)
                e3 = time.perf_counter() - t3
                return {"value": nst / e3, "ms_per_signal": e3 / nst * 1e3, "signals": nst, "in_flight": 2,
                        "same_coefficients_as_the_timed_steps": bool(np.array_equal(Pm[:, 0], np.asarray(params).ravel()))}
            two["through_lpvs_lpv_signals_multi_f64"] = guarded(through, "through_lpvs_lpv_signals_multi_f64")
        return two
    two_in_flight = None
    if world == 1 and args.streams == 1 and not rowsh and not args.no_concurrent:
        two_in_flight = guarded(_two_in_flight, "two_solves_in_flight")
    # ---- dominant kernel of the step: the ADMM mat-vec (one launch per iteration, HBM-bound: it streams the
    # tile-packed lower triangle of M once).  Launch duration measured live with HIP events on the library's stream.
    with L.Problem.lpv(y, X, V, w, NV, True, False, device=local) as p:
        p.set_prox(L.SlicedSeparableSum.frequency_groups(LAMBDA, len(w), 2 * NV))
        p.admm_init(None, μ=MU, tol=0.0)
        mv_us, mv_bytes = p.time_matvec(300)
        mv_info = p.matvec_info()
        nib_us = p.timing()["nibble_refresh_us"]           # one refresh of the stale nibble product where it is kernels of its own (0: none, or part of the iteration's launches)
    phase["nibble_refresh_ms"] = n_nib * nib_us * 1e-3     # (count of the timed steps' solves x that duration: inside admm_ms)
    # the same mat-vec with the inverse stored in doubles (LPVS_M_STORAGE=f64) and in uniform 6-byte elements (=split), for the
    # record: not on the timed path
    def _alt(st):
        with L.Problem.lpv(y, X, V, w, NV, True, False, device=local) as p:
            p.set_option("storage", st)                   # lpvs_problem_set_option(h, LPVS_OPT_M_STORAGE, ...)
            p.set_prox(L.SlicedSeparableSum.frequency_groups(LAMBDA, len(w), 2 * NV))
            p.admm_init(None, μ=MU, tol=0.0)
            a_us, a_bytes = p.time_matvec(300)
            return {"kernel": p.matvec_info()["kernel"], "launch_us": a_us, "bytes_per_launch": a_bytes,
                    "achieved_GBps": a_bytes / (a_us * 1e-6) * 1e-9, "frac_of_hbm_peak": a_bytes / (a_us * 1e-6) * 1e-9 / HBM_PEAK_GBS}
    alt, alt6 = None, None
    if args.dtype == "f64" and not args.no_alt_storage and mv_info["kernel"] in ("symv_tile_split_kernel", "symv_tile_mixed_kernel", "admm_iter_mixed_kernel"):
        alt = guarded(lambda: _alt("f64"), "same_matvec_with_8_byte_storage")
        if mv_info["kernel"] != "symv_tile_split_kernel":
            alt6 = guarded(lambda: _alt("split"), "same_matvec_with_uniform_6_byte_storage")
    # ... and the whole step with that storage (a few untimed-for-`value` solves), so that both end-to-end rates are on the record
    def _alt_step():
        with L.default_options(storage="f64"):               # lpvs_set_default_option: every handle the runs below create
            run()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(3):
                run()
            torch.cuda.synchronize(dev)
            ms8 = (time.perf_counter() - t1) / 3 * 1e3
            return {"ms_per_step": ms8, "signals_per_s_per_gpu": 1e3 / ms8, "steps": 3}
    alt_step = guarded(_alt_step, "whole_step_with_8_byte_storage") if (alt is not None and "error" not in alt and not rowsh) else None
    # ... and with every iteration reading all 36 bits of the fixed-point tiles (storage = "mixed" by name: no stale nibble product -- round 4's reads)
    def _alt36_step():
        with L.default_options(storage="mixed"):
            run()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(3):
                run()
            torch.cuda.synchronize(dev)
            ms36 = (time.perf_counter() - t1) / 3 * 1e3
            return {"ms_per_step": ms36, "signals_per_s_per_gpu": 1e3 / ms36, "steps": 3}
    alt36_step = guarded(_alt36_step, "whole_step_with_36_bit_reads") if (alt is not None and "error" not in alt and not rowsh and n_nib > 0) else None
    # ... and from HOST arrays to the host result (SURVEY 8(d) defines signals/s that way; `value` starts from HBM-resident inputs as the bench contract
    # asks): the same solve with numpy inputs -- 3 x 8 MB uploaded by the constructor inside the timed call
    def _host_step():
        yh, Xh, Vh, wh = (a.cpu().numpy() for a in (y, X, V, w))
        solve(L, yh, Xh, Vh, wh, NV, iters, local)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(3):
            solve(L, yh, Xh, Vh, wh, NV, iters, local)
        torch.cuda.synchronize(dev)
        msh = (time.perf_counter() - t1) / 3 * 1e3
        return {"ms_per_step": msh, "signals_per_s_per_gpu": 1e3 / msh, "steps": 3, "uploaded_bytes_per_step": int(3 * yh.nbytes + wh.nbytes)}
    host_step = guarded(_host_step, "from_host_arrays") if (not rowsh and not args.no_alt_storage) else None
    mv_only_us = mv_us
    all_us = phase["admm_ms"] * 1e3 / iters                # everything inside the ADMM loops' events / iterations: corrections and refreshes included (rounds <= 4 priced this)
    if mv_info.get("one_launch_iteration"):
        # the iteration IS one launch of this kernel (update in its prologue, fixed-point accumulation at its end): its duration inside
        # the timed region = HIP events around the ADMM loop of every timed step / launches (the loop holds nothing else but the first
        # launch without an update and one update-only launch per 2000); mv_only_us = the same kernel without its update, back to back
        # ... minus the x-update corrections (three in 2000 iterations: two packed products and one accurate Gram product each), which are
        # other kernels: their time is phase_xcorr_ms.  The refreshes of the stale nibble product (103 per solve) stay IN: each is the launch
        # it follows, which then also reads the 4-bit planes (156.5 instead of 140.2 MB), and a vector kernel of 2 us -- +0.4 us on the
        # average (phase_nibble_refresh_ms is non-zero only where the refresh is three kernels of its own: LPVS_NIB_FUSED=0)
        mv_us = (phase["admm_ms"] - phase["xcorr_ms"] - phase["nibble_refresh_ms"]) * 1e3 / iters
    mv_share = iters * mv_us * 1e-3 / (elapsed / steps * 1e3)
    # fixed-point tiles of the packed inverse (from the byte count: 66 048 B per fixed-point tile read at 32 bits, 98 304 B per float-head tile)
    _np = -(-(2 * NF * NV) // 128) * 128
    _nt = (_np // 128) * (_np // 128 + 1) // 2
    n_fixed_tiles = int(round((_nt * 98304 - mv_bytes) / (98304 - 66048))) if n_nib > 0 else 0
    # (the one-launch kernel's instance that carries the update: <1, ...>; <0, ...> is a chunk's first launch, <2, ...> its last update)
    # (the PMC summary is collected from the f64 bench: the _f32 handles run another instance of the kernel on other bytes)
    traffic, traffic_src = pmc_traffic(mv_info["kernel"] + ("<1" if mv_info.get("one_launch_iteration") else "")) if args.log2n == LOG2N and args.dtype == "f64" else (None, None)
    achieved = mv_bytes / (mv_us * 1e-6) * 1e-9
    # ---- the dense f64-MFMA Gram the library uses when w is NOT an arithmetic progression: measured once outside
    # the timed region (same inputs, LPVS_GRAM_FORM=krs) so both rooflines are on the record.
    def _general():
        with L.default_options(gram_form="krs"):
            gt = None
            for _ in range(2):
                with L.Problem.lpv(y, X, V, w, NV, True, False, device=local) as p:
                    gt = p.timing()
        g_alg = gt["gram_flops"] / (gt["gram_ms"] * 1e-3) * 1e-12
        g_iss = gt["gram_issued_flops"] / (gt["gram_ms"] * 1e-3) * 1e-12
        return {"bound": "mfma", "kernel": "gram_kernel<KRS> (v_mfma_f64_16x16x4_f64)", "launch_ms": gt["gram_ms"],
                "achieved": g_iss, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": g_iss / F64_MFMA_PEAK_TFLOPS,
                "achieved_algorithmic": g_alg, "algorithmic_flops_per_launch": gt["gram_flops"], "issued_flops_per_launch": gt["gram_issued_flops"],
                "step_ms_with_this_form": elapsed / steps * 1e3 - phase["gram_ms"] - phase["reduce_rhs_ms"] - phase["basis_ms"]
                                          + gt["gram_ms"] + gt["reduce_rhs_ms"] + gt["basis_ms"],
                "note": "arbitrary-w path, NOT taken by this workload (its w is an arithmetic progression -> structured Gram). achieved / frac = "
                        "flops the matrix cores actually issue per second (the symmetric-pair contraction issues 2Nv/(Nv+1) = 1.78x fewer "
                        "flops than the n x n lower triangle N*n*(n+1) that achieved_algorithmic is priced with); issue ceiling measured "
                        "by tools/mfma_f64_peak.hip: 66-67 TFLOP/s"}
    general = guarded(_general, "gram_general_path") if (not args.no_general_path and not rowsh) else None
    out = {
        "metric": "signals/sec, ls_sparse_spectral_lpv group lasso N=2^%d Nf=%d Nv=%d (%d ADMM iters; iters/sec in admm_iters_per_sec)" % (args.log2n, NF, NV, iters),
        "value": (1 if rowsh else world) * steps / elapsed, "unit": "signals/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if rowsh else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "cfg3 ls_sparse_spectral_lpv group lasso N=2^%d Nf=%d Nv=%d n=%d lam=%g mu=%g iters=%d tol=0; 1 signal/GPU"
                               % (args.log2n, NF, NV, 2 * NF * NV, LAMBDA, MU, iters),
                   "signals_per_step_per_gpu": 1.0 / world if rowsh else 1, "gram_form": form,
                   "gram": ("structured, slot sums by a non-uniform FFT (nufft.hip); MFMA path not taken" if form == "ap-nufft" else
                            "structured (VALU f64, nudft.hip); MFMA path not taken" if form == "ap" else "dense f64 MFMA (%s)" % form),
                   "matvec_storage": mv_info["storage"], "xupdate_corrections_per_solve": n_xcorr, "nibble_refreshes_per_solve": n_nib, "nibble_refresh_us": nib_us,
                   "xupdate_correction": "after iterations 16, 128, 256, 512, 1024, ...: one step of iterative refinement of the x-update's offset vector, residual in twice the mantissa (DESIGN 6.1)",
                   "whole_step_with_8_byte_storage": alt_step, "whole_step_with_36_bit_reads": alt36_step, "from_host_arrays": host_step, "concurrent_solves_per_gpu": args.streams, "two_solves_in_flight": two_in_flight,
                   "sharding": "sample rows of one signal over the ranks, one all-reduce of the Gram (SURVEY 8(e)(2))" if rowsh else "independent signals",
                   "final_gather": "none" if (world == 1 or rowsh) else ("rccl" if args.backend == "nccl" else args.backend) + " all_gather"},
        "admm_iters_per_sec": iters / (phase["admm_ms"] * 1e-3),
        "phase_ms": phase,
        "factor_ms": phase["factor_ms"],
        "factorisation": {"bound": "mfma", "kernel": "rank_updatem_kernel + pivot chain (blocked symmetric sweep, two 128-wide steps per pass over A, v_mfma_f64_16x16x4_f64)",
                          "flops": float(2 * NF * NV) ** 3, "ms": phase["factor_ms"],
                          "achieved": float(2 * NF * NV) ** 3 / (phase["factor_ms"] * 1e-3) * 1e-12, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": float(2 * NF * NV) ** 3 / (phase["factor_ms"] * 1e-3) * 1e-12 / F64_MFMA_PEAK_TFLOPS,
                          "note": "n^3 flop on the lower triangle for M = (G + I/mu)^-1; whole phase incl. the pivot chains and hand-overs"},
        "final_nxz": nxz,
        "roofline": {"bound": "hbm", "kernel": mv_info["kernel"] + " (ADMM mat-vec with the tile-packed lower triangle of (G + I/mu)^-1)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", **roof_fracs(achieved),
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": mv_bytes,
                     "launch_us": mv_us, "launch_us_all_of_admm": all_us, "launches_per_step": iters, "share_of_step": mv_share, "same_matvec_with_8_byte_storage": alt, "same_matvec_with_uniform_6_byte_storage": alt6,
                     # (for comparisons across rounds: the same launch priced with the bytes round 4's iteration read -- all 36 bits of the fixed-point
                     # tiles -- and with the doubles SURVEY 8(d) counts; neither is `frac`, which counts the bytes the launch reads now)
                     **({"frac_if_priced_with_36_bit_bytes": (mv_bytes + 8192.0 * n_fixed_tiles) / (mv_us * 1e-6) * 1e-9 / HBM_PEAK_GBS} if n_nib > 0 and n_fixed_tiles else {}),
                     "matvec_only_launch_us": mv_only_us,
                     "note": ("algorithmic bytes = %s (+ 0.2 MB of state vectors); M is read once per iteration; ONE launch per iteration: the kernel "
                              "rebuilds its right-hand-side blocks (prox + dual update) in the prologue and adds its partial sums into x with 64-bit "
                              "fixed-point atomics; launch_us = (HIP events around the ADMM loops of the timed steps - the x-update corrections inside them, timed by events of their own, ; 103 of the 2000 launches also multiply the 4-bit planes of their tile and are followed by a vector kernel -- the stale nibble product, +0.4 us on this average) / launches (the two-launch scheme, "
                              "LPVS_ITERATION=two: mat-vec 24.7-25.7 us + update 5.3 us = 31.6 us per iteration); matvec_only_launch_us = that "
                              "scheme's stand-alone mat-vec kernel (the same product, no update), 300 back-to-back launches" % mv_info["bytes_formula"])
                             if mv_info.get("one_launch_iteration") else
                             "algorithmic bytes = %s; M is read once per iteration; duration = HIP events around 300 back-to-back launches on "
                             "the library's stream" % mv_info["bytes_formula"]},
        "gram_general_path": general,
        "_params_rank0": np.asarray(params).ravel() if (params is not None and not rowsh) else None,   # (popped by main(): the single-process record compares with it)
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = guarded(cpu_baseline, "cpu_baseline")
    return out


def lockstep(run, failed):
    """Sub-records (`sub` mode of run_cfg2 / run_cfg5): a rank whose LOCAL part raises keeps taking part in every collective of the
    record (zero contribution); the failure is all-reduced afterwards and the record reads {"error": ...} -- as measure_cfg4 does."""
    def safe():
        if failed:
            return None
        try:
            return run()
        except Exception as e:                                 # noqa: BLE001
            failed.append("%s: %s" % (type(e).__name__, e))
            return None
    return safe


def run_cfg2(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks, sub=None):
    """BASELINE.json config 2: ls_sparse_spectral NormL1(0.01), N = 2^18, Nf = 512 (n = 1024), 5000 iterations; one signal per GPU.
    sub = dict(steps, warmup, iters): the compact record that rides on the default (cfg3) line."""
    steps = sub["steps"] if sub else (10 if args.steps is None else args.steps)
    warmup = sub["warmup"] if sub else (2 if args.warmup is None else args.warmup)
    iters = sub["iters"] if sub else (CFG2["iters"] if args.iters is None else args.iters)
    N, Nf = 1 << CFG2["log2n"], CFG2["Nf"]
    failed = []

    def run():
        with L.Problem.fourier(y, t, ft, None, device=local) as p:
            p.set_prox(L.NormL1(CFG2["lam"]))
            p.admm_init(None, μ=CFG2["mu"], tol=0.0)
            it, nxz, conv = p.admm_run(iters)
            return p.params(0), it, nxz, p.timing()
    if sub:
        run = lockstep(run, failed)
    y = t = ft = f = None
    try:
        y, t, f = synth_fourier(N, Nf, dev)
        ft = torch.tensor(f, dtype=torch.float64, device=dev)
    except Exception as e:                                     # noqa: BLE001
        if not sub:
            raise
        failed.append("%s: %s" % (type(e).__name__, e))

    for _ in range(warmup):
        run()
    sync()
    t0 = time.perf_counter()
    tms = []
    for _ in range(steps):
        r = run()
        if r is not None:
            params, it, nxz, tm = r
            tms.append(tm)
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if sub and max_over_ranks(1.0 if failed else 0.0) > 0:
        return {"error": failed[0] if failed else "a rank other than 0 failed in its local part (see its stderr)", "n_gpus": world} if rank == 0 else None
    if rank != 0:
        return None
    phase = {k: float(np.mean([q[k] for q in tms])) for k in ("basis_ms", "gram_ms", "reduce_rhs_ms", "factor_ms", "admm_ms")}
    # ---- several of these latency-bound solves in flight (host threads, each handle its own stream: the dependent-launch gaps of one
    # solve's iterations are filled by the others' launches); beside the judged one-at-a-time line, not part of the timed region
    in_flight = None
    if world == 1 and not args.no_concurrent and not sub:
        import threading
        in_flight = {}
        for nth in (2, 4):
            nst = 4 * nth
            def worker():
                for _ in range(nst // nth):
                    run()
            th = [threading.Thread(target=worker) for _ in range(nth)]
            sync()
            t1 = time.perf_counter()
            [q.start() for q in th]
            [q.join() for q in th]
            sync()
            in_flight[str(nth)] = {"signals_per_s": nst / (time.perf_counter() - t1), "signals": nst}
    with L.Problem.fourier(y, t, ft, None, device=local) as p:
        p.set_prox(L.NormL1(CFG2["lam"]))
        p.admm_init(None, μ=CFG2["mu"], tol=0.0)
        mv_us, mv_bytes = p.time_matvec(500)
        info = p.matvec_info()
    it_us = phase["admm_ms"] * 1e3 / iters
    one_launch = bool(info.get("one_launch_iteration"))
    mv_only_us = mv_us
    if one_launch:
        mv_us = it_us                                                  # the iteration IS one launch of admm_small_iter_kernel (measured over the timed steps)
    achieved = mv_bytes / (mv_us * 1e-6) * 1e-9
    traffic2, traffic2_src = pmc_traffic("admm_small_iter_kernel<1" if one_launch else "symv_kernel", "cfg2")
    npad = -(-2 * Nf // 128) * 128
    out = {"metric": "signals/sec, ls_sparse_spectral NormL1(%g) N=2^%d Nf=%d (%d ADMM iters; iters/sec in admm_iters_per_sec)" % (CFG2["lam"], CFG2["log2n"], Nf, iters),
           "value": world * steps / elapsed, "unit": "signals/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "cfg2: ls_sparse_spectral NormL1(%g), N=2^%d non-equidistant, Nf=%d (no zero frequency, n=%d), mu=%g, iters=%d, tol=0, one signal per GPU"
                                  % (CFG2["lam"], CFG2["log2n"], Nf, 2 * Nf, CFG2["mu"], iters), "gram_form": tms[0]["gram_form"], "matvec_storage": info["storage"],
                      "solves_in_flight": in_flight},
           "admm_iters_per_sec": iters / (phase["admm_ms"] * 1e-3), "phase_ms": phase, "final_nxz": nxz,
           "peaks_1based": sorted((np.argsort(-np.abs(params))[:5] + 1).tolist()),
           "roofline": {"bound": "hbm", "kernel": info["kernel"] + " (full symmetric (G + I/mu)^-1, n = 1024: 8.4 MB, Infinity-Cache resident)",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", **roof_fracs(achieved), "traffic": traffic2, "traffic_source": traffic2_src,
                        "algorithmic_bytes_per_launch": mv_bytes, "launch_us": mv_us, "launches_per_step": iters, "iteration_us": it_us,
                        "matvec_only_launch_us": mv_only_us,
                        "redundant_state_bytes_per_launch": float(npad // 4) * 3 * 8 * npad if one_launch else 0.0,
                        "note": ("launch-latency bound, not bandwidth bound: the iteration is ONE launch of np / 4 workgroups (admm_small_iter_kernel: every workgroup redoes "
                                 "the update from x, u, b -- 24 KB each, 6.3 MB per launch besides the 8.4 MB of M -- and multiplies its four rows), replayed as hipGraph chunks "
                                 "of 250 iterations; launch_us = iteration_us = HIP events around the ADMM loop / iterations; matvec_only_launch_us = the two-launch scheme's "
                                 "stand-alone mat-vec, 500 back-to-back launches.  The floor is the dependent-kernel boundary (1.4-1.8 us measured between two launches of "
                                 "the cfg3 iteration, profiles/r04_iteration_timeline.txt) plus one fabric round trip for state written by other XCDs, not the bytes.")
                                if one_launch else
                                "launch-latency bound, not bandwidth bound: an iteration is two dependent launches of a few microseconds (mat-vec + prox / "
                                "dual update), replayed as hipGraph chunks of 50 iterations; iteration_us = HIP events around the ADMM loop / iterations; "
                                "launch_us = the mat-vec alone, 500 back-to-back launches.  The floor is the dependent-kernel boundary "
                                "(MI355X_MICROARCH.md: 1.2-1.9 us each), not the 8.4 MB the kernel reads."}}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = guarded(cpu_baseline_cfg2, "cpu_baseline")
    return out


def run_cfg5(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks, sub=None):
    """BASELINE.json config 5: multichannel LPV, `channels` channels per GPU sharing (X, V): ONE Gram / factorisation per GPU, every
    ADMM kernel advances all its channels in one pass over the inverse; IndBallL0(32), n = 32768.  Ranks take disjoint channel ranges
    (weak scaling: 8 channels per GPU -> 64 on 8 GPUs), no data-path collective, one all_gather of the coefficients.
    sub = dict(steps, warmup, iters, log2n[, Nf, Nv, channels]): the compact record that rides on the default (cfg3) line."""
    steps = sub["steps"] if sub else (2 if args.steps is None else args.steps)
    warmup = sub["warmup"] if sub else (1 if args.warmup is None else args.warmup)
    iters = sub["iters"] if sub else (CFG5["iters"] if args.iters is None else args.iters)
    log2n = sub["log2n"] if sub else args.log2n
    N, Nf, Nv, ns = 1 << log2n, (sub or {}).get("Nf", CFG5["Nf"]), (sub or {}).get("Nv", CFG5["Nv"]), (sub or {}).get("channels", args.channels)
    failed = []

    def run():
        with L.Problem.lpv_multi(Y, X, V, w, Nv, True, False, device=local) as p:
            p.set_prox(L.IndBallL0(CFG5["r"]))
            p.admm_init(None, μ=CFG5["mu"], tol=0.0)
            it, nxz, conv = p.admm_run(iters)
            return p.params(0), it, nxz, p.timing()
    if sub:
        run = lockstep(run, failed)
    Y = X = V = w = None
    try:
        Y, X, V, w = synth_channels(N, Nf, ns, dev, first=rank * ns)
    except Exception as e:                                     # noqa: BLE001
        if not sub:
            raise
        failed.append("%s: %s" % (type(e).__name__, e))

    for _ in range(warmup):
        run()
    sync()
    t0 = time.perf_counter()
    tms, params = [], None
    for _ in range(steps):
        r = run()
        if r is not None:
            params, it, nxz, tm = r
            tms.append(tm)
    if dist is not None:
        mine_np = np.ascontiguousarray(params.T) if params is not None and not failed else np.zeros((ns, Nf * Nv), dtype=np.complex128)
        mine = torch.view_as_real(torch.tensor(mine_np, device=cdev)).contiguous()
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if sub and max_over_ranks(1.0 if failed else 0.0) > 0:
        del Y, X, V
        return {"error": failed[0] if failed else "a rank other than 0 failed in its local part (see its stderr)", "n_gpus": world} if rank == 0 else None
    if rank != 0:
        return None
    phase = {k: float(np.mean([q[k] for q in tms])) for k in ("basis_ms", "gram_ms", "reduce_rhs_ms", "factor_ms", "admm_ms")}
    with L.Problem.lpv_multi(Y, X, V, w, Nv, True, False, device=local) as p:
        p.set_prox(L.IndBallL0(CFG5["r"]))
        p.admm_init(None, μ=CFG5["mu"], tol=0.0)
        mv_us, mv_bytes = p.time_matvec(30)
        info = p.matvec_info()
    n = 2 * Nf * Nv
    # the launch also WRITES its tile partials: one record per run of tiles for the row sums, one per tile for the
    # column sums, 128 rows x `signals_per_pass` doubles each -- algorithmic bytes of the kernel as built, counted from this round on
    nblk5 = n // 128
    ntiles5 = nblk5 * (nblk5 + 1) // 2
    spp = info.get("signals_per_pass", 16)
    nruns = min(ntiles5, (ntiles5 + 7) // 8 + nblk5)
    partial_bytes = float(ntiles5 + nruns) * 128 * spp * 8
    tile_bytes = mv_bytes
    mv_bytes = tile_bytes + partial_bytes
    achieved = mv_bytes / (mv_us * 1e-6) * 1e-9
    traffic5, traffic5_src = pmc_traffic("symv_tile_mfma_ws_kernel", "cfg5") if (log2n == CFG5["log2n"] and n == 32768) else (None, None)
    out = {"metric": "signals/sec, multichannel ls_sparse_spectral_lpv IndBallL0(%d) N=2^%d Nf=%d Nv=%d, %d channels per GPU (%d ADMM iters)"
                     % (CFG5["r"], log2n, Nf, Nv, ns, iters),
           "value": world * ns * steps / elapsed, "unit": "signals/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "cfg5: %d channels per GPU sharing (X, V), N=2^%d, Nf=%d, Nv=%d (n=%d), IndBallL0(%d), mu=%g, iters=%d, tol=0"
                                  % (ns, log2n, Nf, Nv, n, CFG5["r"], CFG5["mu"], iters), "gram_form": tms[0]["gram_form"],
                      "matvec_storage": info["storage"], "sharding": "disjoint channel ranges per rank", "final_gather": "none" if world == 1 else "all_gather of the coefficients"},
           "admm_iters_per_sec": iters / (phase["admm_ms"] * 1e-3), "iteration_ms_all_channels": phase["admm_ms"] / iters,
           "phase_ms": phase, "final_nxz": nxz, "nnz_per_channel": [int(np.count_nonzero(params[:, q])) for q in range(ns)],
           "factorisation": {"flops": float(n) ** 3, "ms": phase["factor_ms"], "achieved_TFLOPs": float(n) ** 3 / (phase["factor_ms"] * 1e-3) * 1e-12,
                             "frac_of_f64_mfma_peak": float(n) ** 3 / (phase["factor_ms"] * 1e-3) * 1e-12 / F64_MFMA_PEAK_TFLOPS},
           "roofline": {"bound": "hbm", "kernel": info["kernel"] + " (tile product of the packed (G + I/mu)^-1 with all channels on %s)" % info.get("mfma", "the matrix cores"),
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", **roof_fracs(achieved), "traffic": traffic5, "traffic_source": traffic5_src,
                        "algorithmic_bytes_per_launch": mv_bytes, "tile_bytes_per_launch": tile_bytes, "partial_bytes_written_per_launch": partial_bytes,
                        "achieved_on_tile_bytes_only": tile_bytes / (mv_us * 1e-6) * 1e-9, "launch_us": mv_us, "launches_per_step": iters,
                        "share_of_step": iters * mv_us * 1e-3 / (elapsed / steps * 1e3),
                        "mfma_flops_per_launch": 4.0 * float(n) * float(n + 128) / 2 * info.get("signals_per_pass", 16) * 2 / 2,
                        "note": "algorithmic bytes = %s + the tile partials the launch writes (one record per run of tiles for the row sums, one per tile "
                                "for the column sums: ~34 + 270 MB at n = 32768); HIP events around 30 back-to-back launches on the library's stream; "
                                "the same launch also issues n(n+128)/2 x %d signal columns x 2 products of f64 MFMA work; launch time varies by ~10 %% with the "
                                "box and with where the 3.2 GB buffer landed (DESIGN.md 4.5.1)" % (info["bytes_formula"], info.get("signals_per_pass", 16))}}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = guarded(cpu_baseline_cfg5, "cpu_baseline")
    return out


# sources of the kernel a workload's roofline is about (the same table in tools/pmc_summary.py): a PMC summary is stamped with their sha256
KERNEL_SOURCES = {"cfg3": ("admm_device.h", "admm_one_launch.hip"), "cfg4": ("admm_device.h", "admm_one_launch.hip"),
                  "cfg2": ("admm_device.h", "admm_small.hip"), "cfg5": ("admm_device.h", "admm_multi.hip")}


def source_sha16(workload="cfg3"):
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES[workload]:
        with open(os.path.join(ROOT, "lpvspectral.jl_amd", "csrc", rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_prefix, workload="cfg3"):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of `workload`
    (PMC passes cannot run inside the timed bench).  tools/pmc_summary.py stamps a summary with the sha256 of the kernel source
    it was collected from (KERNEL_SOURCES); a summary whose stamp is missing or differs from the present sources is NOT quoted (a stale
    figure is worse than none)."""
    import glob
    now = source_sha16(workload)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc_traffic*.json")), reverse=True):
        try:
            rows = json.load(open(path))
        except Exception:
            continue
        meta = next((r for r in rows if r.get("kernel") == "__meta__"), None)
        rel = os.path.relpath(path, ROOT)
        if not meta or meta.get("kernel_sources_sha16") != now or meta.get("workload", "cfg3") != workload:
            continue
        for r in rows:
            # (the instance <..., true> of the one-launch iteration also multiplies the nibble planes -- 103 launches in 2000: not the dominant one)
            if r.get("kernel", "").startswith(kernel_prefix) and not r["kernel"].endswith(", true>"):
                return (r["fetch_corrected_bytes_per_launch"] + r["write_bytes_per_launch"],
                        f"{rel} (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, bytes per launch; kernel sources sha256 {now})")
    return None, "no PMC summary of %s under profiles/ was collected from the present kernel sources (sha256 %s): not quoted" % (workload, now)


if __name__ == "__main__":
    main()
