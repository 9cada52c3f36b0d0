"""Sharding of independent units (signals of a batch, windows of ``ls_windowpsd``) over the ranks of a
node and the final gather -- SURVEY.md §8(e).  One process per GPU; no data-path collective; one
``all_gather`` (RCCL on GPUs, gloo in the CPU tests) of fixed-size per-unit results at the end.  The
reference loops over windows sequentially (src/lsfft.jl:120-123); results are re-assembled in unit
order so that any reduction over units (``S .+= abs2.(x)``) can be done in the reference's order."""
from __future__ import annotations

import numpy as np


def shard_range(n_units: int, world: int, rank: int):
    """Contiguous block partition: rank r owns units [lo, hi); sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_units), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def window_sample_span(lo: int, hi: int, n: int, noverlap: int):
    """Sample interval [s0, s1) a rank needs for windows [lo, hi) (window i starts at i*(n-noverlap))."""
    if hi <= lo:
        return 0, 0
    step = n - noverlap
    return lo * step, (hi - 1) * step + n


def gather_units(local, n_units: int, dist=None, device=None):
    """All ranks contribute ``local`` (array [hi-lo, m], possibly complex) and receive the full [n_units, m]
    array in unit order.  ``dist`` is ``torch.distributed`` (initialised) or None for a single process."""
    local = np.asarray(local)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert local.shape[0] == n_units
        return local
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    is_c = np.iscomplexobj(local)
    m = local.shape[1]
    width = m * (2 if is_c else 1)
    cap = -(-n_units // world)                    # equal-size slots keep it a single all_gather
    buf = torch.zeros((cap, width), dtype=torch.float64, device=device)
    flat = local.view(np.float64).reshape(local.shape[0], width) if is_c else local.astype(np.float64)
    buf[: local.shape[0]] = torch.as_tensor(np.ascontiguousarray(flat), device=device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_units, world, r)
        rows.append(out[r][: hi - lo].cpu().numpy())
    full = np.concatenate(rows, axis=0)
    return full.view(np.complex128).reshape(n_units, m) if is_c else full


def reduce_psd_in_order(x_units):
    """S = sum_i |x_i|^2 accumulated in unit order, then / k^2 (src/lsfft.jl:122,125)."""
    x_units = np.asarray(x_units)
    S = np.zeros(x_units.shape[1])
    for i in range(x_units.shape[0]):
        S += np.abs(x_units[i]) ** 2
    return S / x_units.shape[0] ** 2


# ---- one exchange step for a single large problem: sample rows sharded, partial Grams summed ----
def allreduce_ranges(r4, dist=None):
    """Combine per-shard ``[min V, max V, max|V|, max|X|]`` into the global ranges (min for entry 0, max for the rest)."""
    r4 = np.asarray(r4, dtype=np.float64).copy()
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return r4
    import torch
    dev = _collective_device(dist)
    t = torch.tensor([-r4[0], r4[1], r4[2], r4[3]], dtype=torch.float64, device=dev)   # one MAX reduction: min = -max(-.)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out = t.cpu().numpy()
    out[0] = -out[0]
    return out


def allreduce_sum_(t, dist=None):
    """In-place sum of a tensor over the ranks.  Device tensors go straight into RCCL (backend nccl); with a
    host-only backend (gloo rehearsal) they are staged through host memory."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if t.is_cuda:   # the library reads the result on its own streams: the collective must be complete
            import torch
            torch.cuda.current_stream(t.device).synchronize()
    return t


def _collective_device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
