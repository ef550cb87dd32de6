"""Data parallelism over spatial tiles: one process per GPU, one flat fp32
gradient bucket all-reduced per optimizer step (RCCL over xGMI on MI355X;
``gloo`` in the CPU tests).

The reference has no multi-GPU path (``Trainer()`` defaults, no
collectives).  Tiles are independent units there (training drops inter-tile
edges, ``data/partition/dataset.py:480-494``), so the natural sharding is
tiles -> ranks with replicated parameters (~0.45 M values, 1.8 MB): the
all-reduce is latency-bound, so it is ONE collective on one contiguous buffer.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


class FlatGradBucket:
    """One flat fp32 gradient exchange per optimizer step.

    Autograd leaves every parameter its own ``.grad`` tensor (with ``zero_grad(set_to_none=True)`` the
    accumulation step is a pointer hand-over, not an add kernel per parameter).  ``all_reduce_mean`` packs
    those ~50 tensors into ONE contiguous buffer (a single ``cat``), runs ONE collective on it, and scatters
    the averaged slices back with one multi-tensor copy -- three launches instead of one collective per
    parameter.  With a single rank it does nothing at all."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("parameters are kept in fp32")
        self.numel = sum(p.numel() for p in self.params)
        self.flat: torch.Tensor = torch.empty(0)

    def all_reduce_mean(self, group=None) -> None:
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        grads = []
        for p in self.params:                      # a rank whose tile never touched a parameter sends zeros
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
        self.flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(world)
        views, off = [], 0
        for g in grads:
            views.append(self.flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        torch._foreach_copy_(grads, views)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s weights (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    ps = [p for p in module.parameters()]
    flat = torch.cat([p.detach().reshape(-1).float() for p in ps])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for p in ps:
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()


def assign_tiles(weights: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of tiles (by edge count) to ranks
    so per-step work is balanced; deterministic (ties by tile id)."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    loads = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += weights[i]
    return out


def rank_schedule(batch_weights: Sequence[float], world_size: int) -> List[List[Optional[int]]]:
    """Per-rank list of batch ids for one epoch: batches go to ranks by :func:`assign_tiles` (balanced edge
    counts), each rank runs its heaviest batch first, and shorter lists are padded with ``None`` so every rank
    takes the same number of optimizer steps.  On a ``None`` step a rank skips forward / backward and still
    calls ``FlatGradBucket.all_reduce_mean`` (it contributes zeros; the mean is over the world size, as DDP's
    join does), so the collective never deadlocks on uneven batch counts."""
    per_rank = assign_tiles(batch_weights, world_size)
    steps = max((len(r) for r in per_rank), default=0)
    return [list(r) + [None] * (steps - len(r)) for r in per_rank]


def seed_rank(base_seed: int, rank: int, encoder=None) -> None:
    """Rank-offset RNG streams: torch's generators (negative sampling, triplet sampling) and, when the encoder
    is given, its device-side attention-dropout counter -- replicas must not draw identical masks."""
    torch.manual_seed(int(base_seed) + 7919 * int(rank))
    if encoder is not None and hasattr(encoder, "_step_dev"):
        with torch.no_grad():
            encoder._step_dev.fill_(int(rank) << 40)


def rank_census(device=None, group=None) -> dict:
    """Evidence that the collective really spans ``world_size`` distinct ranks: every rank adds a one-hot of its own
    rank; the all-reduced vector must be all ones.  -> ``{"world_size", "n_ranks_seen", "census"}``."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"world_size": 1, "n_ranks_seen": 1, "census": [1]}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    v = torch.zeros(world, dtype=torch.int32, device=device)
    v[rank] = 1
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    census = [int(x) for x in v.tolist()]
    return {"world_size": world, "n_ranks_seen": sum(1 for c in census if c == 1), "census": census}


def strong_scaling_epoch(batch_weights: Sequence[float], step, units=None, *, sync=None, device=None,
                         warmup: int = 0, group=None) -> dict:
    """One data-parallel epoch over a FIXED list of packed batches (BASELINE config 4: total work does not grow
    with the number of ranks).  ``step(k, i)`` runs optimizer step ``i`` on batch ``k`` (``None``: this rank ran out
    of batches -- it must still take part in the gradient all-reduce, see :func:`rank_schedule`); ``units(k)`` returns
    the work units of batch ``k`` (a sequence of numbers, summed over all ranks); ``sync()`` drains the device;
    ``warmup`` untimed steps come first (negative: this rank's whole schedule once, i.e. the timed epoch is a second
    epoch and finds every per-tile cache warm).
    The epoch is bracketed by barrier + sync on both sides and the reported time is the MAX over ranks."""
    import time
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    sched = rank_schedule(batch_weights, world)[rank]
    sync = sync or (lambda: None)
    for k in (sched if warmup < 0 else sched[:warmup]):     # warmup < 0: one whole untimed epoch first
        step(k, 0)
    sync()
    if on:
        dist.barrier(group)
    sync()
    t0 = time.perf_counter()
    done = None
    for i, k in enumerate(sched):
        step(k, i)
        if k is not None and units is not None:
            u = [float(x) for x in units(k)]
            done = u if done is None else [a + b for a, b in zip(done, u)]
    sync()
    if on:
        dist.barrier(group)
    sync()
    dt = time.perf_counter() - t0
    n_units = len(done) if done is not None else (len(units(0)) if (units is not None and len(batch_weights)) else 0)
    stats = torch.tensor([dt] + (done if done is not None else [0.0] * n_units), dtype=torch.float64, device=device)
    if on:
        tmax = stats[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
        sums = stats[1:].clone(); dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dt, total = float(tmax[0]), [float(x) for x in sums.tolist()]
    else:
        total = [float(x) for x in stats[1:].tolist()]
    out = {"scaling": "strong", "batches": len(batch_weights), "steps_per_rank": len(sched),
           "own_batches": sum(1 for k in sched if k is not None), "epoch_s": dt, "units_total": total}
    out.update(rank_census(device, group))
    return out
