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
