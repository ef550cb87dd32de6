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
    """One flat fp32 gradient exchange per optimizer step on a PERSISTENT buffer.

    Autograd leaves every parameter its own ``.grad`` tensor (with ``zero_grad(set_to_none=True)`` the accumulation
    step is a pointer hand-over, not an add kernel per parameter).  ``pack`` gathers those ~50 tensors into the flat
    buffer with ONE multi-tensor copy and re-points every ``.grad`` at its slice of it; ``all_reduce_mean`` then runs
    ONE collective on the buffer and divides by the world size in place -- three launches, no allocation, and no copy
    back: the optimizer reads the averaged slices.  The buffer is allocated once (same address every step; a captured
    optimizer graph may read it).  With a single rank ``all_reduce_mean`` does nothing at all -- unless
    ``force_collective`` is set: then the one-rank communicator is exercised too (a world-1 mean is the identity; how the
    RCCL path is tested on a one-GPU box)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], force_collective: bool = False):
        self.force_collective = bool(force_collective)
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("parameters are kept in fp32")
        self.numel = sum(p.numel() for p in self.params)
        self.flat: torch.Tensor = torch.empty(0)
        self.views: List[torch.Tensor] = []

    def _ensure(self) -> None:
        dev = self.params[0].device
        if self.flat.numel() != self.numel or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
            self.views, off = [], 0
            for p in self.params:
                self.views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()

    def pack(self) -> None:
        """``.grad`` tensors -> slices of the flat buffer (one multi-tensor copy), then ``.grad`` = those slices.  A
        parameter without a gradient (a rank whose tile never touched it, an empty step) contributes zeros."""
        self._ensure()
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                src.append(g); dst.append(v)
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)

    def zero(self) -> None:
        """An empty step: zeros into the exchange."""
        self._ensure()
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce_mean(self, group=None, packed: bool = False, force: Optional[bool] = None) -> None:
        """``packed``: the caller has already run :meth:`pack` / :meth:`zero` (e.g. inside a captured graph).
        ``force`` (default: ``self.force_collective``): run pack + collective + divide on a one-rank group too."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1 and not (self.force_collective if force is None else force):
            return
        if not packed:
            self.pack()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(world)


def all_agree(ok: bool, device=None, group=None) -> bool:
    """True on every rank iff ``ok`` is true on every rank (one MIN all-reduce): how ranks decide TOGETHER whether to
    enter a phase that contains collectives -- a rank that failed its local pre-flight must not leave the others
    waiting inside an all-reduce."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(ok)
    v = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
    return bool(int(v.item()))


@torch.no_grad()
def snapshot_training_state(module: torch.nn.Module, optimizer=None) -> dict:
    """Copies of the parameters (and the optimizer's tensors) for :func:`restore_training_state`: a pre-flight that
    takes real optimizer steps on rank-local data must not leave the replicas different."""
    snap = {"params": [p.detach().clone() for p in module.parameters()], "opt": None}
    if optimizer is not None:
        snap["opt"] = {id(p): {k: (v.clone() if isinstance(v, Tensor) else v) for k, v in st.items()}
                       for p, st in optimizer.state.items()}
    return snap


@torch.no_grad()
def restore_training_state(module: torch.nn.Module, snap: dict, optimizer=None) -> None:
    """In place (captured graphs keep their pointers): parameters back to the snapshot; optimizer tensors back to the
    snapshot, or zeroed where the snapshot had none (Adam's initial state)."""
    torch._foreach_copy_([p for p in module.parameters()], snap["params"])
    if optimizer is not None:
        old = snap["opt"] or {}
        for p, st in optimizer.state.items():
            for k, v in st.items():
                if isinstance(v, Tensor):
                    o = old.get(id(p), {}).get(k)
                    v.zero_() if o is None else v.copy_(o)
    try:
        from . import ops
        ops.invalidate_weights(module.parameters())
    except Exception:  # noqa: BLE001  (CPU-only use)
        pass


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


def schedule_fingerprint(batch_weights: Sequence[float]) -> int:
    """A 62-bit hash of the batch list's weights (order included), the same on every platform and process (Python's own
    ``hash`` of floats is, but not salted strings'; this one is explicit FNV-1a over the IEEE-754 bytes)."""
    import struct
    h = 0xcbf29ce484222325
    for w in batch_weights:
        for b in struct.pack("<d", float(w)):
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h >> 2


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
    if on:
        # every rank must take the same number of steps (each one is a collective: a mismatch would deadlock, not fail) over
        # the SAME batch list: ranks that derive the weights themselves (rank-local FOV generation) could agree on the count
        # and still train overlapping or missing batches -- one MAX all-reduce of (x, -x) pairs compares the step count and
        # a 62-bit fingerprint of the weights
        fp = schedule_fingerprint(batch_weights)
        n = torch.tensor([len(sched), -len(sched), fp, -fp], dtype=torch.int64, device=device)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
        if int(n[0]) != len(sched) or int(-n[1]) != len(sched):
            raise RuntimeError(f"strong_scaling_epoch: rank {rank} has {len(sched)} steps, others between {int(-n[1])} and "
                               f"{int(n[0])}: the ranks do not agree on the batch list")
        if int(n[2]) != fp or int(-n[3]) != fp:
            raise RuntimeError(f"strong_scaling_epoch: rank {rank}'s batch weights have fingerprint {fp:016x}, another rank's "
                               f"differ: the ranks do not agree on the batch list")
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
