"""N>1 path on CPU: world_size-2 gloo processes exercise the unit sharding and the final gather that
bench.py / the windowed drivers use on GPUs with RCCL (same code, different backend)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_units, m, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lpvspectral_jl_amd import sharding
    lo, hi = sharding.shard_range(n_units, world, rank)
    rng = np.random.default_rng(1234)
    allx = rng.standard_normal((n_units, m)) + 1j * rng.standard_normal((n_units, m))   # same on every rank
    full = sharding.gather_units(allx[lo:hi], n_units, dist, torch.device("cpu"))
    ok = np.array_equal(full, allx)
    S = sharding.reduce_psd_in_order(full)
    S_ref = np.zeros(m)
    for i in range(n_units):
        S_ref += np.abs(allx[i]) ** 2
    ok = ok and np.array_equal(S, S_ref / n_units ** 2)            # bit-identical: same order as src/lsfft.jl:122
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # the max-over-ranks timing reduction of bench.py
    ok = ok and t.item() == world
    dist.barrier()
    q.put((rank, bool(ok), lo, hi))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units,m", [(7, 5), (8, 3), (1, 4)])
def test_gather_two_ranks_gloo(n_units, m):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, m, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(r[1] for r in res), res
    assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == n_units     # contiguous, complete


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    from lpvspectral_jl_amd import sharding
    for n in (0, 1, 7, 64, 1024, 1000):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(1024, 8, 3) == (384, 512)          # cfg4: 128 windows per GPU
    assert sharding.window_sample_span(384, 512, 65536, 0) == (384 * 65536, 512 * 65536)
    assert sharding.window_sample_span(1, 3, 10, 1) == (9, 28)     # windows start at 9 and 18, each 10 long


# ---- sample rows of ONE signal sharded, partial Grams summed by one all-reduce -------------------
def _row_worker(rank, world, port, N, Nf, Nv, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lpvspectral_jl_amd import sharding
    from oracle import oracle as o
    rng = np.random.default_rng(99)
    X = np.sort(rng.uniform(0, 10, N)); V = rng.uniform(-1, 2, N); y = rng.standard_normal(N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) / 4
    lo, hi = sharding.shard_range(N, world, rank)
    Xs, Vs = X[lo:hi], V[lo:hi]
    local = np.array([Vs.min(), Vs.max(), np.abs(Vs).max(), np.abs(Xs).max()])
    ranges = sharding.allreduce_ranges(local, dist)
    ok = np.array_equal(ranges, [V.min(), V.max(), np.abs(V).max(), np.abs(X).max()])
    # rows of the regressor depend on the other rows only through the basis centres, i.e. through `ranges`
    Phi = o.lpv_regressor(X, V, w, Nv)
    Gr = torch.from_numpy(Phi[lo:hi].T @ Phi[lo:hi]); br = torch.from_numpy(Phi[lo:hi].T @ y[lo:hi])
    sharding.allreduce_sum_(Gr, dist); sharding.allreduce_sum_(br, dist)
    G = Phi.T @ Phi; b = Phi.T @ y
    ok = ok and np.allclose(Gr.numpy(), G, rtol=0, atol=1e-12 * np.abs(G).max())
    ok = ok and np.allclose(br.numpy(), b, rtol=0, atol=1e-12 * np.abs(b).max())
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_row_sharded_gram_exchange_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_row_worker, args=(r, world, port, 301, 6, 3, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(r[1] for r in res), res
