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

`python3 bench.py --gpus N` starts the N ranks itself (a torch.distributed.run child process, before this
process touches the GPU) when it was not launched by one.

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
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
F64_MFMA_PEAK_TFLOPS = 78.6   # AMD datasheet FP64 matrix; MI355X_MICROARCH.md lists no f64 MFMA row
# cfg4: L = 2^26 equidistant, 1024 windows of 2^16, freqs (0:255)/512, L1 lam = 0.2, mu = 1e-4, 2000 its
CFG4 = dict(nwin=1024, log2n=16, Nf=256, lam=0.2, mu=1e-4, iters=2000)


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


def cpu_baseline(log2n_sample=14, iters=6):
    """Faithful CPU form (dense Phi in memory, warm-started CG on the lazy Phi'Phi + I/mu, extra Phi*x per
    iteration) on a bounded sample: N_s = 2^14 rows at the full n = 8192, a few ADMM iterations; the cost
    per ADMM iteration is a pure N*n stream, so it is extrapolated linearly in N to 2^20 and in iterations
    to 2000 (optimistic for the CPU: CG needs more iterations as N grows)."""
    from oracle import oracle as o
    Ns = 1 << log2n_sample
    y, X, V, w = [a.cpu().numpy() for a in synth_signal(Ns, NF, 0, "cpu")]
    t0 = time.time()
    Phi = o.lpv_regressor(X, V, w, NV)
    t_asm = time.time() - t0
    t0 = time.time()
    r = o.admm_ls(Phi, y, o.GroupL2(LAMBDA, 2 * NV), iters=iters, tol=0.0, mu=MU)
    t_admm = time.time() - t0
    scale = float(1 << (LOG2N - log2n_sample))
    per_iter = t_admm / r["iters"] * scale
    total = t_asm * scale + per_iter * ADMM_ITERS
    return {
        "value": 1.0 / total, "unit": "signals/s", "cores": o.num_threads(), "kind": "port",
        "sample": f"N=2^{log2n_sample} rows at full n=8192, {r['iters']} ADMM iterations "
                  f"({r['cg_iters'] / r['iters']:.1f} CG iterations each, {t_admm:.1f} s) + regressor assembly ({t_asm:.1f} s); "
                  f"extrapolated x{int(scale)} in N and to {ADMM_ITERS} iterations",
        "admm_iters_per_sec": 1.0 / per_iter,
    }


def cpu_baseline_cfg4(n_sample_windows=2, iters=40):
    """cfg4 on the CPU port: a few windows of the full size (n = 2^16, Nreg = 511), regressor + explicit Gram (as the
    reference does for the weighted method, src/lasso.jl:118-120) + `iters` CG-based ADMM iterations, extrapolated
    linearly in windows and iterations."""
    from oracle import oracle as o
    n, Nf = 1 << CFG4["log2n"], CFG4["Nf"]
    y, t, f = synth_windows(n_sample_windows, n, Nf, "cpu")
    y, t = y.numpy(), t.numpy()
    t0 = time.time()
    spent_gram, spent_admm = 0.0, 0.0
    for i in range(n_sample_windows):
        yi, ti = y[i * n:(i + 1) * n], t[i * n:(i + 1) * n]
        t1 = time.time()
        A, zf = o.get_fourier_regressor(ti, f)
        Q, q = o.gram(A, yi, np.ones(n))
        t2 = time.time()
        o.admm_quadratic(Q, q, o.NormL1(CFG4["lam"]), iters=iters, tol=0.0, mu=CFG4["mu"])
        spent_gram += t2 - t1; spent_admm += time.time() - t2
    per_window = spent_gram / n_sample_windows + spent_admm / n_sample_windows / iters * CFG4["iters"]
    return {"value": 1.0 / per_window, "unit": "windows/s", "cores": o.num_threads(), "kind": "port",
            "sample": f"{n_sample_windows} windows of 2^16 samples at Nf=256: regressor + Gram {spent_gram / n_sample_windows:.2f} s/window, "
                      f"{iters} ADMM iterations {spent_admm / n_sample_windows:.2f} s/window ({time.time() - t0:.1f} s in all); extrapolated to "
                      f"{CFG4['iters']} iterations per window"}


def spawn_ranks(args, argv):
    """`python3 bench.py --gpus N` outside a launcher: start N ranks as a CHILD process tree (torch.distributed.run) before
    this process has made any HIP call, and exit with its status.  (Nothing is exec'ed and no GPU state is inherited.)"""
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
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg4"],
                    help="cfg3 (default, the judged line): one LPV group-lasso signal per GPU per step; cfg4: 1024 batched windows per "
                         "step, window range sharded over the ranks")
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
    ap.add_argument("--row-sharded", action="store_true",
                    help="strong-scaling variant: one signal per step, its sample rows sharded over the ranks (one all-reduce of the Gram)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: map every rank onto the visible GPUs modulo their count")
    args = ap.parse_args()
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

    if args.workload == "cfg4":
        out = run_cfg4(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks)
    else:
        out = run_cfg3(L, args, world, rank, local, dev, cdev, dist, sync, max_over_ranks)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


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
    st = os.environ.get("LPVS_M_STORAGE", "mixed")                 # storage of the packed inverses (admm.hip)
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
            st, ("admm_iter_mixed_kernel<batch> (the whole ADMM iteration of every window in one launch; 36-bit fixed-point tiles, tile partials "
                 "added into x by 64-bit fixed-point atomics)") if one_launch else
                "symv_tile_mixed_batch_kernel (6-byte float-head tiles on the diagonal, 36-bit fixed-point tiles elsewhere)")
        roof = {"bound": "hbm", "kernel": kern + ": one tile-packed (Q + I/mu)^-1 per window, all windows of the shard per launch",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": mv_bytes, "launch_us": mv_us, "windows_per_launch": nmv, "launches_per_step": iters,
                "matvec_only_launch_us": mv_only_us,
                "note": ("launch_us = HIP events around the ADMM loop of the last timed step / iterations (one launch per iteration); "
                         "matvec_only_launch_us = 200 back-to-back launches of the two-launch scheme's stand-alone batch mat-vec (rank 0's shard)") if one_launch else
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
        out["cpu_baseline"] = cpu_baseline_cfg4()
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
    if args.streams > 1 and not rowsh:
        import threading
        res, lock = [], threading.Lock()
        counter = iter(range(steps))
        def worker():
            while True:
                with lock:
                    k = next(counter, None)
                if k is None:
                    return
                r = run()                      # ctypes releases the GIL inside the library calls
                with lock:
                    res.append(r)
        th = [threading.Thread(target=worker) for _ in range(args.streams)]
        [t_.start() for t_ in th]
        [t_.join() for t_ in th]
        for params, it, nxz, tm in res:
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

    phase = {k: float(np.mean([t[k] for t in tms])) for k in ("basis_ms", "gram_ms", "reduce_rhs_ms", "factor_ms", "admm_ms")}
    form = tms[0]["gram_form"]
    # ---- dominant kernel of the step: the ADMM mat-vec (one launch per iteration, HBM-bound: it streams the
    # tile-packed lower triangle of M once).  Launch duration measured live with HIP events on the library's stream.
    with L.Problem.lpv(y, X, V, w, NV, True, False, device=local) as p:
        p.set_prox(L.SlicedSeparableSum.frequency_groups(LAMBDA, len(w), 2 * NV))
        p.admm_init(None, μ=MU, tol=0.0)
        mv_us, mv_bytes = p.time_matvec(300)
        mv_info = p.matvec_info()
    # the same mat-vec with the inverse stored in doubles (LPVS_M_STORAGE=f64) and in uniform 6-byte elements (=split), for the
    # record: not on the timed path
    alt, alt6 = None, None
    if args.dtype == "f64" and not args.no_alt_storage and mv_info["kernel"] in ("symv_tile_split_kernel", "symv_tile_mixed_kernel", "admm_iter_mixed_kernel"):
        for st in ("f64", "split"):
            if st == "split" and mv_info["kernel"] == "symv_tile_split_kernel":
                continue
            os.environ["LPVS_M_STORAGE"] = st
            try:
                with L.Problem.lpv(y, X, V, w, NV, True, False, device=local) as p:
                    p.set_prox(L.SlicedSeparableSum.frequency_groups(LAMBDA, len(w), 2 * NV))
                    p.admm_init(None, μ=MU, tol=0.0)
                    a_us, a_bytes = p.time_matvec(300)
                    rec = {"kernel": p.matvec_info()["kernel"], "launch_us": a_us, "bytes_per_launch": a_bytes,
                           "achieved_GBps": a_bytes / (a_us * 1e-6) * 1e-9, "frac_of_hbm_peak": a_bytes / (a_us * 1e-6) * 1e-9 / HBM_PEAK_GBS}
                    if st == "f64":
                        alt = rec
                    else:
                        alt6 = rec
            finally:
                del os.environ["LPVS_M_STORAGE"]
    # ... and the whole step with that storage (a few untimed-for-`value` solves), so that both end-to-end rates are on the record
    alt_step = None
    if alt is not None and not rowsh:
        os.environ["LPVS_M_STORAGE"] = "f64"
        try:
            run()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(3):
                run()
            torch.cuda.synchronize(dev)
            ms8 = (time.perf_counter() - t1) / 3 * 1e3
            alt_step = {"ms_per_step": ms8, "signals_per_s_per_gpu": 1e3 / ms8, "steps": 3}
        finally:
            del os.environ["LPVS_M_STORAGE"]
    mv_only_us = mv_us
    if mv_info.get("one_launch_iteration"):
        # the iteration IS one launch of this kernel (update in its prologue, fixed-point accumulation at its end): its duration inside
        # the timed region = HIP events around the ADMM loop of every timed step / launches (the loop holds nothing else but the first
        # launch without an update and one update-only launch per 2000); mv_only_us = the same kernel without its update, back to back
        mv_us = phase["admm_ms"] * 1e3 / iters
    mv_share = iters * mv_us * 1e-3 / (elapsed / steps * 1e3)
    # (the one-launch kernel's instance that carries the update: <1, ...>; <0, ...> is a chunk's first launch, <2, ...> its last update)
    traffic, traffic_src = pmc_traffic(mv_info["kernel"] + ("<1" if mv_info.get("one_launch_iteration") else "")) if args.log2n == LOG2N else (None, None)
    achieved = mv_bytes / (mv_us * 1e-6) * 1e-9
    # ---- the dense f64-MFMA Gram the library uses when w is NOT an arithmetic progression: measured once outside
    # the timed region (same inputs, LPVS_GRAM_FORM=krs) so both rooflines are on the record.
    general = None
    if not args.no_general_path and not rowsh:
        os.environ["LPVS_GRAM_FORM"] = "krs"
        try:
            gt = None
            for _ in range(2):
                with L.Problem.lpv(y, X, V, w, NV, True, False, device=local) as p:
                    gt = p.timing()
        finally:
            del os.environ["LPVS_GRAM_FORM"]
        g_alg = gt["gram_flops"] / (gt["gram_ms"] * 1e-3) * 1e-12
        g_iss = gt["gram_issued_flops"] / (gt["gram_ms"] * 1e-3) * 1e-12
        general = {"bound": "mfma", "kernel": "gram_kernel<KRS> (v_mfma_f64_16x16x4_f64)", "launch_ms": gt["gram_ms"],
                   "achieved": g_iss, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": g_iss / F64_MFMA_PEAK_TFLOPS,
                   "achieved_algorithmic": g_alg, "algorithmic_flops_per_launch": gt["gram_flops"], "issued_flops_per_launch": gt["gram_issued_flops"],
                   "step_ms_with_this_form": elapsed / steps * 1e3 - phase["gram_ms"] - phase["reduce_rhs_ms"] - phase["basis_ms"]
                                             + gt["gram_ms"] + gt["reduce_rhs_ms"] + gt["basis_ms"],
                   "note": "arbitrary-w path, NOT taken by this workload (its w is an arithmetic progression -> structured Gram). achieved / frac = "
                           "flops the matrix cores actually issue per second (the symmetric-pair contraction issues 2Nv/(Nv+1) = 1.78x fewer "
                           "flops than the n x n lower triangle N*n*(n+1) that achieved_algorithmic is priced with); issue ceiling measured "
                           "by tools/mfma_f64_peak.hip: 66-67 TFLOP/s"}
    out = {
        "metric": "signals/sec, ls_sparse_spectral_lpv group lasso N=2^%d Nf=%d Nv=%d (%d ADMM iters; iters/sec in admm_iters_per_sec)" % (args.log2n, NF, NV, iters),
        "value": (1 if rowsh else world) * steps / elapsed, "unit": "signals/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if rowsh else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "cfg3: ls_sparse_spectral_lpv group-lasso N=2^%d Nf=%d Nv=%d n=%d lambda=%g mu=%g iters=%d tol=0, one signal per GPU"
                               % (args.log2n, NF, NV, 2 * NF * NV, LAMBDA, MU, iters),
                   "signals_per_step_per_gpu": 1.0 / world if rowsh else 1, "gram_form": form,
                   "gram": ("structured, slot sums by a non-uniform FFT (nufft.hip); MFMA path not taken" if form == "ap-nufft" else
                            "structured (VALU f64, nudft.hip); MFMA path not taken" if form == "ap" else "dense f64 MFMA (%s)" % form),
                   "matvec_storage": mv_info["storage"], "whole_step_with_8_byte_storage": alt_step, "concurrent_solves_per_gpu": args.streams,
                   "sharding": "sample rows of one signal over the ranks, one all-reduce of the Gram (SURVEY 8(e)(2))" if rowsh else "independent signals",
                   "final_gather": "none" if (world == 1 or rowsh) else ("rccl" if args.backend == "nccl" else args.backend) + " all_gather"},
        "admm_iters_per_sec": iters / (phase["admm_ms"] * 1e-3),
        "phase_ms": phase,
        "final_nxz": nxz,
        "roofline": {"bound": "hbm", "kernel": mv_info["kernel"] + " (ADMM mat-vec with the tile-packed lower triangle of (G + I/mu)^-1)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": mv_bytes,
                     "launch_us": mv_us, "launches_per_step": iters, "share_of_step": mv_share, "same_matvec_with_8_byte_storage": alt, "same_matvec_with_uniform_6_byte_storage": alt6,
                     "matvec_only_launch_us": mv_only_us,
                     "note": ("algorithmic bytes = %s (+ 0.2 MB of state vectors); M is read once per iteration; ONE launch per iteration: the kernel "
                              "rebuilds its right-hand-side blocks (prox + dual update) in the prologue and adds its partial sums into x with 64-bit "
                              "fixed-point atomics; launch_us = HIP events around the ADMM loops of the timed steps / launches (the two-launch scheme, "
                              "LPVS_ITERATION=two: mat-vec 24.7-25.7 us + update 5.3 us = 31.6 us per iteration); matvec_only_launch_us = that "
                              "scheme's stand-alone mat-vec kernel (the same product, no update), 300 back-to-back launches" % mv_info["bytes_formula"])
                             if mv_info.get("one_launch_iteration") else
                             "algorithmic bytes = %s; M is read once per iteration; duration = HIP events around 300 back-to-back launches on "
                             "the library's stream" % mv_info["bytes_formula"]},
        "gram_general_path": general,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    return out


def source_sha16(rel="lpvspectral.jl_amd/csrc/admm.hip"):
    import hashlib
    with open(os.path.join(ROOT, rel), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the mat-vec kernel from the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary
    (PMC passes cannot run inside the timed bench).  tools/pmc_summary.py stamps a summary with the sha256 of the kernel source
    it was collected from; a summary whose stamp is missing or differs from the present csrc/admm.hip is NOT quoted (a stale
    figure is worse than none)."""
    import glob
    now = source_sha16()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc_traffic*.json")), reverse=True):
        try:
            rows = json.load(open(path))
        except Exception:
            continue
        meta = next((r for r in rows if r.get("kernel") == "__meta__"), None)
        rel = os.path.relpath(path, ROOT)
        if not meta or meta.get("admm_hip_sha16") != now:
            continue
        for r in rows:
            if r.get("kernel", "").startswith(kernel_prefix):
                return (r["fetch_corrected_bytes_per_launch"] + r["write_bytes_per_launch"],
                        f"{rel} (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, bytes per launch; collected from csrc/admm.hip sha256 {now})")
    return None, "no PMC summary under profiles/ was collected from the present csrc/admm.hip (sha256 %s): not quoted" % now


if __name__ == "__main__":
    main()
