#!/usr/bin/env python3
"""Headline benchmark: ls_sparse_spectral_lpv (frequency-grouped lasso) at BASELINE.json's quoted
size N=2^20, Nf=512, Nv=8 (n = 8192 unknowns) on N GPUs of one node, one process per GPU.

A step = one complete solve of one synthetic signal whose inputs (y, X, V, w) are already resident
in HBM: basis tables -> f64 MFMA Gram + rhs -> factorisation of (G + I/mu) -> 2000 ADMM iterations
(tol = 0, so exactly 2000) -> parameter read-back.  Ranks solve independent signals (weak scaling,
no data-path collective); RCCL is used only for the final gather of the coefficient vectors.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement) with
  roofline     : the dominant kernel (Gram) against the f64 MFMA peak, from HIP-event timings taken
                 inside the library on the stream the kernel runs on
  cpu_baseline : the faithful CPU restatement of the reference algorithm (oracle, "port") timed on
                 this host on a bounded sample and extrapolated (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

LOG2N, NF, NV = 20, 512, 8
ADMM_ITERS, LAMBDA, MU = 2000, 5.0, 0.05
F64_MFMA_PEAK_TFLOPS = 78.6   # AMD datasheet FP64 matrix; MI355X_MICROARCH.md lists no f64 MFMA row


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


def solve(L, y, X, V, w, Nv, iters, device_index):
    with L.Problem.lpv(y, X, V, w, Nv, True, False, device=device_index) as p:
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=LOG2N, help="diagnostic only; the judged size is 20")
    ap.add_argument("--iters", type=int, default=ADMM_ITERS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI, the judged path) or gloo (functional rehearsal)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: map every rank onto the visible GPUs modulo their count")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import lpvspectral_jl_amd as L   # loads torch's HIP runtime first, then liblpvspectral.so
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.share_gpu:
        local = local % torch.cuda.device_count()
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
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collective buffers live

    N = 1 << args.log2n
    y, X, V, w = synth_signal(N, NF, rank, dev)          # inputs resident in HBM before the timed region

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        solve(L, y, X, V, w, NV, args.iters, local)
    sync()
    t0 = time.perf_counter()
    tms, params = [], None
    for _ in range(args.steps):
        params, it, nxz, tm = solve(L, y, X, V, w, NV, args.iters, local)
        tms.append(tm)
    if dist is not None:                                 # final gather of the coefficients (RCCL)
        mine = torch.view_as_real(torch.tensor(params, device=cdev)).contiguous()
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        gram_ms = float(np.mean([t["gram_ms"] for t in tms]))
        admm_ms = float(np.mean([t["admm_ms"] for t in tms]))
        flops = tms[0]["gram_flops"]
        issued = tms[0]["gram_issued_flops"]
        achieved = flops / (gram_ms * 1e-3) * 1e-12
        issued_rate = issued / (gram_ms * 1e-3) * 1e-12
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_final_pmc_traffic.json")
        if os.path.exists(pmc) and args.log2n == LOG2N:   # PMC passes cannot run inside the timed bench: use the committed
            for r in json.load(open(pmc)):                 # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of this command
                if r["kernel"].startswith("gram_kernel"):
                    traffic = r["fetch_corrected_bytes_per_launch"] + r["write_bytes_per_launch"]
                    traffic_src = "profiles/r01_final_pmc_traffic.json (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, bytes per launch)"
        out = {
            "metric": "signals/sec, ls_sparse_spectral_lpv group lasso N=2^%d Nf=%d Nv=%d (%d ADMM iters; iters/sec in admm_iters_per_sec)" % (args.log2n, NF, NV, args.iters),
            "value": world * args.steps / elapsed, "unit": "signals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg3: ls_sparse_spectral_lpv group-lasso N=2^%d Nf=%d Nv=%d n=%d lambda=%g mu=%g iters=%d tol=0, one signal per GPU"
                                   % (args.log2n, NF, NV, 2 * NF * NV, LAMBDA, MU, args.iters),
                       "signals_per_step_per_gpu": 1, "final_gather": ("rccl" if args.backend == "nccl" else args.backend) + " all_gather" if world > 1 else "none"},
            "admm_iters_per_sec": args.iters / (admm_ms * 1e-3),
            "phase_ms": {k: float(np.mean([t[k] for t in tms])) for k in ("basis_ms", "gram_ms", "reduce_rhs_ms", "factor_ms", "admm_ms")},
            "final_nxz": nxz,
            "roofline": {"bound": "mfma", "kernel": "gram_kernel<KRS> (v_mfma_f64_16x16x4_f64)", "achieved": achieved,
                         "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_flops_per_launch": flops, "launch_ms": gram_ms,
                         "issued_flops_per_launch": issued, "issued_tflops": issued_rate,
                         "issued_frac_of_peak": issued_rate / F64_MFMA_PEAK_TFLOPS,
                         "note": "achieved = N*n*(n+1) algorithmic flops / launch time; it can exceed the MFMA peak because the "
                                 "symmetric-pair contraction issues 2Nv/(Nv+1) (1.78x) fewer flops than the n x n lower triangle; "
                                 "issued_tflops is the rate the matrix cores actually run at",
                         "peak_source": "AMD datasheet FP64 matrix (no f64 MFMA row in MI355X_MICROARCH.md); issue-rate ceiling measured on this pool by tools/mfma_f64_peak.hip: 60-66 TFLOP/s"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
