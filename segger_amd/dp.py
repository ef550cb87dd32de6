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

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


class FlatGradBucket:
    """Owns one contiguous fp32 buffer that every parameter's ``.grad`` is a view of,
    so ``all_reduce`` needs no pack/unpack copies."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("parameters are kept in fp32")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self) -> None:
        self.flat.zero_()

    def reattach(self) -> None:
        """Re-point ``.grad`` at the bucket (after ``zero_grad(set_to_none=True)``)."""
        off = 0
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
            off += p.numel()

    def all_reduce_mean(self, group=None) -> None:
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(world)


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
