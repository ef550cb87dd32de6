"""Host-side operators over the C ABI: raw launch helpers + autograd Functions.

Everything here enqueues HIP kernels from ``libsegger_amd.so`` on torch's
current stream; torch is used for device memory and autograd plumbing only.
There is no CPU path (see ``_lib.require_cuda``).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import DTYPE_CODE
from . import graph as _graph
from .graph import EdgeCSR, EdgeGraph


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _rows(t: Tensor, width: int, name: str) -> Tuple[int, int]:
    """(data_ptr, row stride in elements) of a [n, width] tensor whose last dim is dense."""
    sh = t.shape
    if len(sh) != 2 or sh[1] != width:
        raise ValueError(f"{name}: expected [n, {width}], got {tuple(sh)}")
    st = t.stride()
    if sh[0] > 1:
        if st[1] != 1:
            raise ValueError(f"{name}: last dimension must be contiguous")
        return t.data_ptr(), st[0]
    return t.data_ptr(), max(width, st[0])


import functools
import os


@functools.lru_cache(maxsize=None)
def _has_specialised(heads: int, channels: int) -> bool:
    return bool(_lib.load().segger_gatv2_has_specialised(heads, channels))


@functools.lru_cache(maxsize=4096)
def _gat_bwd_ws_bytes(n_dst: int, heads: int, channels: int) -> int:
    return int(_lib.load().segger_gatv2_bwd_workspace_bytes(n_dst, heads, channels))


@functools.lru_cache(maxsize=4096)
def _wgrad_ws_bytes(n: int, m: int, k: int) -> int:
    return int(_lib.load().segger_linear_wgrad_workspace_bytes(n, m, k))


def _seed_parts(seed):
    """``seed`` is an int, or ``(int, device int64 tensor)``: the kernels then use ``int + *tensor`` read at
    run time, so a captured hipGraph draws a new dropout mask on every replay."""
    if isinstance(seed, tuple):
        value, dev = seed
        return int(value) & 0xFFFFFFFFFFFFFFFF, (None if dev is None else dev.data_ptr())
    return int(seed) & 0xFFFFFFFFFFFFFFFF, None


def _bits_ptr(bits: Tensor, n_edges: int) -> int:
    if bits.dtype != torch.uint8 or bits.numel() != n_edges or not bits.is_contiguous():
        raise ValueError("keep_bits must be a contiguous uint8 tensor with one entry per edge slot")
    return bits.data_ptr()


@torch.no_grad()
def dropout_bits(csr: EdgeCSR, heads: int, dropout_p: float, seeds, seed_dev: Optional[Tensor] = None) -> Tensor:
    """uint8 [len(seeds), n_edges]: the attention-dropout keep bits (bit h = head h) of ``len(seeds)`` layers over the
    slots of one CSR view (``segger_dropout_bits``) -- generated once per training step per view, then every forward /
    backward pass of every layer tests a bit instead of hashing per (edge, head)."""
    _lib.require_cuda(csr.col)
    lib = _lib.load()
    seeds = [int(v) & 0xFFFFFFFFFFFFFFFF for v in seeds]
    stride = (csr.n_edges + 15) // 16 * 16                 # planes start 16-byte aligned: four slots are stored as one word
    out = torch.empty((len(seeds), stride), dtype=torch.uint8, device=csr.col.device)
    arr = (C.c_uint64 * len(seeds))(*seeds)
    with _lib.on_device(out.device):
        rc = lib.segger_dropout_bits(csr.eid.data_ptr() if csr.n_edges else None, csr.n_edges, heads, dropout_p, arr,
                                     len(seeds), _lib.ptr(seed_dev), out.data_ptr(), stride, _lib.stream_ptr(out.device))
    _lib.check(rc, "segger_dropout_bits")
    return out[:, :csr.n_edges]


def dropout_bits_many(views, heads: int, dropout_p: float, seed_dev: Optional[Tensor] = None) -> list:
    """:func:`dropout_bits` for up to four ``(csr, seeds)`` views in ONE launch (``segger_dropout_bits_many``)."""
    views = [(c, [int(v) & 0xFFFFFFFFFFFFFFFF for v in sd]) for c, sd in views]
    if not views:
        return []
    if len(views) > 4:
        return dropout_bits_many(views[:4], heads, dropout_p, seed_dev) + dropout_bits_many(views[4:], heads, dropout_p, seed_dev)
    dev = views[0][0].col.device
    _lib.require_cuda(views[0][0].col)
    lib = _lib.load()
    jobs = (_lib.BitsJob * len(views))()
    outs, keep = [], []
    for i, (csr, seeds) in enumerate(views):
        stride = (csr.n_edges + 15) // 16 * 16
        out = torch.empty((len(seeds), stride), dtype=torch.uint8, device=dev)
        arr = (C.c_uint64 * len(seeds))(*seeds)
        keep.append(arr)
        jobs[i].eid = csr.eid.data_ptr() if csr.n_edges else None
        jobs[i].n_edges, jobs[i].n_seeds = csr.n_edges, len(seeds)
        jobs[i].seeds = C.cast(arr, C.c_void_p)
        jobs[i].bits, jobs[i].plane_stride = out.data_ptr(), stride
        outs.append(out[:, :csr.n_edges])
    with _lib.on_device(dev):
        rc = lib.segger_dropout_bits_many(jobs, len(views), heads, dropout_p, _lib.ptr(seed_dev), _lib.stream_ptr(dev))
    _lib.check(rc, "segger_dropout_bits_many")
    return outs


@torch.no_grad()
def step_advance(step: Tensor, inc: int) -> Tensor:
    """``step += inc`` in place and a snapshot of the new value, one launch (``segger_step_advance``); int64[1] on the GPU."""
    _lib.require_cuda(step)
    if step.dtype != torch.int64 or step.numel() != 1:
        raise ValueError("step_advance: int64[1]")
    snap = torch.empty_like(step)
    with _lib.on_device(step.device):
        rc = _lib.load().segger_step_advance(step.data_ptr(), int(inc), snap.data_ptr(), _lib.stream_ptr(step.device))
    _lib.check(rc, "segger_step_advance")
    return snap


@torch.no_grad()
def step_draws(views, heads: int, dropout_p: float, samplers, negatives, seed_dev: Tensor, advance=None):
    """Every random draw of one training step in ONE launch (``segger_step_draws``): ``views`` = up to four ``(csr, seeds)``
    of :func:`dropout_bits_many` (or []), ``samplers`` = up to two ``(index, seed)`` of :func:`triplet_sample`, ``negatives``
    = ``(pos, n_b, n_b_dev, seed)`` of :func:`sample_negatives` or None; all streams read the device word ``seed_dev``;
    ``advance``: up to 64 fp32 device scalars incremented by one on the way (:func:`adam_step_counters`).
    -> (planes, [(pos, neg, d_pos, d_neg), ...], negatives | None): what the separate calls return for the same seeds."""
    if len(views) > 4 or len(samplers) > 2:
        raise ValueError("step_draws: at most four views and two samplers")
    _lib.require_cuda(seed_dev)
    lib = _lib.load()
    dev = seed_dev.device
    a = _lib.StepDrawsArgs()
    keep, planes, draws = [], [], []
    if views:
        jobs = (_lib.BitsJob * len(views))()
        for i, (csr, seeds) in enumerate(views):
            seeds = [int(v) & 0xFFFFFFFFFFFFFFFF for v in seeds]
            stride = (csr.n_edges + 15) // 16 * 16
            out = torch.empty((len(seeds), stride), dtype=torch.uint8, device=dev)
            arr = (C.c_uint64 * len(seeds))(*seeds)
            keep.append(arr)
            jobs[i].eid = csr.eid.data_ptr() if csr.n_edges else None
            jobs[i].n_edges, jobs[i].n_seeds = csr.n_edges, len(seeds)
            jobs[i].seeds = C.cast(arr, C.c_void_p)
            jobs[i].bits, jobs[i].plane_stride = out.data_ptr(), stride
            planes.append(out[:, :csr.n_edges])
        keep.append(jobs)
        a.bits, a.n_bits, a.heads, a.dropout_p = C.cast(jobs, C.c_void_p), len(views), int(heads), float(dropout_p)
    a.n_samplers = len(samplers)
    for i, (index, seed) in enumerate(samplers):
        lab = index["lab"]
        n = int(lab.numel())
        pos = torch.empty(n, dtype=torch.int64, device=dev)
        neg = torch.empty(n, dtype=torch.int64, device=dev)
        dd = torch.empty((2, n), dtype=torch.float32, device=dev)
        j = a.samplers[i]
        j.lab, j.n, j.n_clusters = lab.data_ptr(), n, int(index["n_clusters"])
        j.cdf_pos, j.cdf_neg = index["cdf_pos_t"].data_ptr(), index["cdf_neg_t"].data_ptr()
        j.counts, j.offsets, j.members = index["counts"].data_ptr(), index["offsets"].data_ptr(), index["members"].data_ptr()
        j.seed, j.dists = int(seed) & 0xFFFFFFFFFFFFFFFF, index["dists"].data_ptr()
        j.pos, j.neg, j.d_pos, j.d_neg = pos.data_ptr(), neg.data_ptr(), dd[0].data_ptr(), dd[1].data_ptr()
        draws.append((pos, neg, dd[0], dd[1]))
    out_neg = None
    if negatives is not None:
        npos, n_b, n_b_dev, nseed = negatives
        npos = npos.to(torch.int64).contiguous()
        out_neg = torch.empty_like(npos)
        keep.append(npos)
        a.neg_pos, a.neg_n, a.neg_n_b, a.neg_n_b_dev = npos.data_ptr(), int(npos.numel()), int(n_b), _lib.ptr(n_b_dev)
        a.neg_seed, a.neg_out = int(nseed) & 0xFFFFFFFFFFFFFFFF, out_neg.data_ptr()
    a.seed_dev = seed_dev.data_ptr()
    if advance:
        ptrs = (C.c_void_p * len(advance))(*[t.data_ptr() for t in advance])
        keep.append(ptrs)
        a.advance, a.n_advance = C.cast(ptrs, C.c_void_p), len(advance)
    with _lib.on_device(dev):
        rc = lib.segger_step_draws(C.byref(a), _lib.stream_ptr(dev))
    _lib.check(rc, "segger_step_draws")
    return planes, draws, out_neg


# ---- deferred partial sums (csrc/reduce.hip) ----------------------------------------------------------------
_DEFER_KEEP: Optional[list] = None


class deferred_reductions:
    """Inside this context the final sums of per-workgroup partials (weight / bias gradients of every projection,
    grad_att / grad_bias of every conv) are queued instead of launched; leaving it runs them all as ONE kernel
    (``segger_reductions_flush``).  The gradient TENSORS handed out meanwhile are placeholders: nothing may read them
    before the context exits, so this is for callers that hold the gradients themselves (``torch.autograd.grad`` in
    ``train_step_graph``), not for ``loss.backward()`` into ``.grad`` accumulators.  Workspaces are kept alive here."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        global _DEFER_KEEP
        if _DEFER_KEEP is not None:
            raise RuntimeError("deferred_reductions does not nest")
        with _lib.on_device(self.device):        # the table is pinned to the device current HERE and to the stream of the
            rc = _lib.load().segger_reductions_defer_begin()     # first producer: a backward elsewhere sums for itself
        _lib.check(rc, "segger_reductions_defer_begin")
        _DEFER_KEEP = []
        return self

    def __exit__(self, *exc):
        global _DEFER_KEEP
        try:
            with _lib.on_device(self.device):
                rc = _lib.load().segger_reductions_flush(_lib.stream_ptr(self.device))
            _lib.check(rc, "segger_reductions_flush")
        finally:
            _DEFER_KEEP = None
        return False


def _defer_keep(*tensors) -> None:
    if _DEFER_KEEP is not None:
        _DEFER_KEEP.append(tensors)


def _f32_vec(t: Optional[Tensor], n: int, name: str) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype == torch.float32 and t.numel() == n and t.is_contiguous():
        return t                                         # (only its data_ptr is used: the common case costs no torch call)
    t = t.detach().reshape(-1)
    if t.numel() != n:
        raise ValueError(f"{name}: expected {n} elements, got {t.numel()}")
    return t.to(torch.float32).contiguous()


# --------------------------------------------------------------------------
# GATv2 aggregation: raw launches
# --------------------------------------------------------------------------
def gatv2_fwd_launch(by_dst: EdgeCSR, xl: Tensor, xr: Tensor, att: Tensor, bias: Optional[Tensor],
                     heads: int, channels: int, out: Tensor, *, pre: Optional[Tensor] = None,
                     lse: Optional[Tensor] = None, alpha: Optional[Tensor] = None,
                     apply_gelu: bool = False, negative_slope: float = 0.2,
                     dropout_p: float = 0.0, seed: int = 0, keep_bits: Optional[Tensor] = None) -> None:
    a, keep = _gat_fwd_args(by_dst, xl, xr, att, bias, heads, channels, out, pre=pre, lse=lse, alpha=alpha,
                            apply_gelu=apply_gelu, negative_slope=negative_slope, dropout_p=dropout_p, seed=seed,
                            keep_bits=keep_bits)
    with _lib.on_device(xl.device):
        rc = _lib.load().segger_gatv2_fwd(C.byref(a), _lib.stream_ptr(xl.device))
    _lib.check(rc, "segger_gatv2_fwd")


_FWD_PAIR = 1                   # (0: always two segger_gatv2_fwd calls -- A/B switch of tools/bench_step.py)


def gatv2_fwd_pair_launch(first: dict, second: dict) -> None:
    """Two forwards of one hetero layer in ONE launch (``segger_gatv2_fwd_pair``): ``first`` the low-degree edge type
    (tx-neighbors-tx), ``second`` the high-degree one (tx-belongs-bd); each a dict of :func:`gatv2_fwd_launch`'s
    arguments.  Falls back to two launches inside the library when the pair does not qualify."""
    a, keep_a = _gat_fwd_args(**first)
    b, keep_b = _gat_fwd_args(**second)
    dev = first["xl"].device
    lib = _lib.load()
    with _lib.on_device(dev):
        if _FWD_PAIR:
            rc = lib.segger_gatv2_fwd_pair(C.byref(a), C.byref(b), _lib.stream_ptr(dev))
        else:
            rc = lib.segger_gatv2_fwd(C.byref(a), _lib.stream_ptr(dev)) or lib.segger_gatv2_fwd(C.byref(b), _lib.stream_ptr(dev))
    _lib.check(rc, "segger_gatv2_fwd_pair")


def _gat_fwd_args(by_dst: EdgeCSR, xl: Tensor, xr: Tensor, att: Tensor, bias: Optional[Tensor],
                  heads: int, channels: int, out: Tensor, *, pre: Optional[Tensor] = None,
                  lse: Optional[Tensor] = None, alpha: Optional[Tensor] = None,
                  apply_gelu: bool = False, negative_slope: float = 0.2,
                  dropout_p: float = 0.0, seed: int = 0, keep_bits: Optional[Tensor] = None):
    _lib.require_cuda(xl, xr, att, out)
    hc = heads * channels
    if not (xl.dtype == xr.dtype == out.dtype) or xl.dtype not in DTYPE_CODE:
        raise TypeError(f"gatv2: x_l/x_r/out must share a dtype in {list(DTYPE_CODE)}")
    a = _lib.GatFwdArgs()
    a.by_dst = by_dst.c_struct(ordered=_graph.ROW_ORDER_FORWARD)
    a.x_l, a.ld_xl = _rows(xl, hc, "x_l")
    a.x_r, a.ld_xr = _rows(xr, hc, "x_r")
    if xl.shape[0] != by_dst.n_cols or xr.shape[0] != by_dst.n_rows or out.shape[0] != by_dst.n_rows:
        raise ValueError("gatv2: feature row counts do not match the graph")
    vecs = (_f32_vec(att, hc, "att"), _f32_vec(bias, hc, "bias"))
    a.att, a.bias = vecs[0].data_ptr(), _lib.ptr(vecs[1])
    a.heads, a.channels, a.dtype, a.apply_gelu = heads, channels, DTYPE_CODE[xl.dtype], int(apply_gelu)
    a.negative_slope, a.dropout_p = negative_slope, dropout_p
    a.seed, a.seed_dev = _seed_parts(seed)
    a.out, a.ld_out = _rows(out, hc, "out")
    if pre is not None:
        a.pre, a.ld_pre = _rows(pre, hc, "pre")
    a.lse, a.alpha = _lib.ptr(lse), _lib.ptr(alpha)
    if keep_bits is not None and dropout_p > 0.0:
        a.keep_bits = _bits_ptr(keep_bits, by_dst.n_edges)
    return a, vecs                                       # (vecs: the fp32 copies the struct points at)


def gatv2_bwd_launch(g: EdgeGraph, xl: Tensor, xr: Tensor, att: Tensor, bias: Optional[Tensor],
                     heads: int, channels: int, grad_out: Tensor, pre: Tensor, lse: Tensor,
                     grad_xl: Tensor, grad_xr: Tensor, **kw) -> Tuple[Tensor, Tensor]:
    """Writes grad_xl / grad_xr (views allowed); returns (grad_att[HC], grad_bias[HC]) fp32.  ``zero_rows_out``: a
    second [n_src, HC] matrix the source pass zero-fills on its way (ignored -> ``False`` comes back in
    ``gatv2_bwd_launch.zero_filled`` when this edge type runs the one-pass form or the generic kernels);
    ``grad_xl_zeroed``: the one-pass form may skip its own zero fill."""
    a, gparams, keep = _gat_bwd_args(g, xl, xr, att, bias, heads, channels, grad_out, pre, lse, grad_xl, grad_xr, **kw)
    with _lib.on_device(xl.device):
        rc = _lib.load().segger_gatv2_bwd(C.byref(a), _lib.stream_ptr(xl.device))
    _lib.check(rc, "segger_gatv2_bwd")
    return gparams[0], gparams[1]


_BWD_PAIR = 1                   # (0: always two segger_gatv2_bwd calls -- A/B switch of tools/ab_graphed.py)


def gatv2_bwd_pair_launch(first: tuple, first_kw: dict, second: tuple, second_kw: dict):
    """The backward of both edge types of one hetero layer through ``segger_gatv2_bwd_pair``: ``first`` the two-pass
    edge type (tx-neighbors-tx), ``second`` the one-pass one (tx-belongs-bd), each the positional / keyword arguments
    of :func:`gatv2_bwd_launch`.  ``second``'s ``grad_xl`` is the matrix ``first`` zero-fills (``zero_rows_out``); the
    library merges the source pass of ``first`` with the destination pass of ``second`` for small batches and runs the
    two backward passes one after the other otherwise.  -> ((grad_att, grad_bias) of first, of second)."""
    a, gp_a, keep_a = _gat_bwd_args(*first, **first_kw)
    zeroed = gatv2_bwd_launch.zero_filled
    b, gp_b, keep_b = _gat_bwd_args(*second, grad_xl_zeroed=zeroed, **second_kw)
    dev = first[1].device
    lib = _lib.load()
    with _lib.on_device(dev):
        if _BWD_PAIR:
            rc = lib.segger_gatv2_bwd_pair(C.byref(a), C.byref(b), _lib.stream_ptr(dev))
        else:
            rc = lib.segger_gatv2_bwd(C.byref(a), _lib.stream_ptr(dev)) or lib.segger_gatv2_bwd(C.byref(b), _lib.stream_ptr(dev))
    _lib.check(rc, "segger_gatv2_bwd_pair")
    return (gp_a[0], gp_a[1]), (gp_b[0], gp_b[1])


def _gat_bwd_args(g: EdgeGraph, xl: Tensor, xr: Tensor, att: Tensor, bias: Optional[Tensor],
                  heads: int, channels: int, grad_out: Tensor, pre: Tensor, lse: Tensor,
                  grad_xl: Tensor, grad_xr: Tensor, *, apply_gelu: bool, negative_slope: float = 0.2,
                  dropout_p: float = 0.0, seed: int = 0, keep_bits: Optional[Tuple] = None,
                  zero_rows_out: Optional[Tensor] = None, grad_xl_zeroed: bool = False, passes: int = 0,
                  scratch: Optional[Tuple[Tensor, Tensor]] = None):
    """-> (segger_gatv2_bwd_args, gparams [2, HC] fp32, the tensors the struct points at).  ``passes`` (1: destination
    pass alone, 2: source pass alone over the ``scratch`` = (grad_pre, dsum) a passes = 1 call returned in
    ``gatv2_bwd_launch.scratch``): timing hooks of bench.py, see include/segger_amd.h."""
    _lib.require_cuda(xl, xr, grad_out)
    lib = _lib.load()
    hc = heads * channels
    dev, dt = xl.device, xl.dtype
    n_dst = g.n_dst
    a = _lib.GatBwdArgs()
    a.by_dst = g.by_dst.c_struct()
    gatv2_bwd_launch.zero_filled = False
    if g.by_src is None and g.src_unique() and _has_specialised(heads, channels):
        a.src_unique = 1          # every source has at most one out-edge: the destination pass stores grad_xl itself
        a.grad_xl_zeroed = int(bool(grad_xl_zeroed))
    else:
        a.by_src = g.require_by_src().c_struct()
        if zero_rows_out is not None and _has_specialised(heads, channels) and g.n_src > 0:
            a.zero_rows_out, a.ld_zero = _rows(zero_rows_out, hc, "zero_rows_out")
            gatv2_bwd_launch.zero_filled = True
    a.x_l, a.ld_xl = _rows(xl, hc, "x_l")
    a.x_r, a.ld_xr = _rows(xr, hc, "x_r")
    vecs = (_f32_vec(att, hc, "att"), _f32_vec(bias, hc, "bias"))
    a.att, a.bias = vecs[0].data_ptr(), _lib.ptr(vecs[1])
    a.heads, a.channels, a.dtype, a.apply_gelu = heads, channels, DTYPE_CODE[dt], int(apply_gelu)
    a.negative_slope, a.dropout_p = negative_slope, dropout_p
    a.seed, a.seed_dev = _seed_parts(seed)
    if keep_bits is not None and dropout_p > 0.0:
        if keep_bits[0] is not None:
            a.keep_bits_dst = _bits_ptr(keep_bits[0], g.n_edges)
        if keep_bits[1] is not None:
            a.keep_bits_src = _bits_ptr(keep_bits[1], g.n_edges)
    if grad_out.dtype != dt:
        grad_out = grad_out.to(dt)
    if grad_out.dim() == 2 and grad_out.shape[0] > 1 and grad_out.stride(1) != 1:
        grad_out = grad_out.contiguous()
    a.grad_out, a.ld_go = _rows(grad_out, hc, "grad_out")
    a.pre, a.ld_pre = _rows(pre, hc, "pre")
    a.lse = lse.data_ptr()
    if scratch is not None:
        grad_pre, dsum = scratch
    else:
        grad_pre = torch.empty((n_dst, hc), dtype=dt, device=dev)
        dsum = torch.empty((n_dst, heads, 2), dtype=torch.float32, device=dev)     # (lse, D) pairs for the source pass
    a.passes = int(passes)
    gatv2_bwd_launch.scratch = (grad_pre, dsum)
    a.grad_pre, a.ld_gp = _rows(grad_pre, hc, "grad_pre")
    a.dsum = dsum.data_ptr()
    a.grad_xl, a.ld_gxl = _rows(grad_xl, hc, "grad_xl")
    a.grad_xr, a.ld_gxr = _rows(grad_xr, hc, "grad_xr")
    gparams = torch.empty((2, hc), dtype=torch.float32, device=dev)
    a.grad_att, a.grad_bias = gparams[0].data_ptr(), gparams[1].data_ptr()
    ws_bytes = _gat_bwd_ws_bytes(n_dst, heads, channels)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws_bytes
    _defer_keep(ws, gparams)
    return a, gparams, (vecs, grad_out, grad_pre, dsum, ws)


class _GatV2Aggregate(torch.autograd.Function):
    """One edge type: (x_l, x_r, att, bias) -> out [n_dst, H*C] (GELU optionally fused)."""

    @staticmethod
    def forward(ctx, xl, xr, att, bias, graph: EdgeGraph, heads, channels, apply_gelu,
                negative_slope, dropout_p, seed, want_alpha, keep_bits=None):
        hc = heads * channels
        dev, dt = xl.device, xl.dtype
        n_dst = graph.n_dst
        need_grad = any(ctx.needs_input_grad[:4])
        out = torch.empty((n_dst, hc), dtype=dt, device=dev)
        pre = torch.empty((n_dst, hc), dtype=dt, device=dev) if (need_grad and apply_gelu) else None
        lse = torch.empty((n_dst, heads), dtype=torch.float32, device=dev) if need_grad else None
        alpha = torch.empty((graph.n_edges, heads), dtype=torch.float32, device=dev) if want_alpha else None
        gatv2_fwd_launch(graph.by_dst, xl, xr, att, bias, heads, channels, out, pre=pre, lse=lse, alpha=alpha,
                         apply_gelu=apply_gelu, negative_slope=negative_slope, dropout_p=dropout_p, seed=seed,
                         keep_bits=None if keep_bits is None else keep_bits[0])
        if need_grad:
            ctx.save_for_backward(xl, xr, att, bias, pre if apply_gelu else out, lse)
            ctx.graph, ctx.cfg = graph, (heads, channels, apply_gelu, negative_slope, dropout_p, seed)
            ctx.keep_bits = keep_bits
        if want_alpha:
            ctx.mark_non_differentiable(alpha)
            return out, alpha
        return out, None

    @staticmethod
    def backward(ctx, grad_out, _grad_alpha):
        xl, xr, att, bias, pre, lse = ctx.saved_tensors
        heads, channels, apply_gelu, slope, p, seed = ctx.cfg
        g = ctx.graph
        hc = heads * channels
        gxl = torch.empty((g.n_src, hc), dtype=xl.dtype, device=xl.device)
        gxr = torch.empty((g.n_dst, hc), dtype=xl.dtype, device=xl.device)
        gatt, gbias = gatv2_bwd_launch(g, xl, xr, att, bias, heads, channels, grad_out, pre, lse, gxl, gxr,
                                       apply_gelu=apply_gelu, negative_slope=slope, dropout_p=p, seed=seed,
                                       keep_bits=ctx.keep_bits)
        gatt = gatt.reshape(att.shape).to(att.dtype)
        gbias = gbias.reshape(bias.shape).to(bias.dtype) if bias is not None else None
        return gxl, gxr, gatt, gbias, None, None, None, None, None, None, None, None, None


def gatv2_aggregate(xl: Tensor, xr: Tensor, att: Tensor, bias: Optional[Tensor], graph: EdgeGraph,
                    heads: int, channels: int, *, apply_gelu: bool = False, negative_slope: float = 0.2,
                    dropout_p: float = 0.0, seed: int = 0, return_alpha: bool = False, keep_bits=None):
    """``keep_bits`` = (by_dst bits, by_src bits or None): one layer's planes of :func:`dropout_bits` for ``seed``."""
    out, alpha = _GatV2Aggregate.apply(xl, xr, att, bias, graph, heads, channels, apply_gelu,
                                       negative_slope, dropout_p, seed, return_alpha, keep_bits)
    return (out, alpha) if return_alpha else out


class _HeteroGatLayer(torch.autograd.Function):
    """segger's HeteroConv layer as one autograd node.

    ``xp_tx`` = [lin_l(tx-tx) | lin_r(tx-tx) | lin_l(tx-bd)] (x_tx), one fused
    projection [Nt, 3*HC]; ``xp_bd`` = lin_r(tx-bd)(x_bd) [Nb, HC].  The three
    backward passes write their slices of one [Nt, 3*HC] gradient, so the
    projection's backward is a single GEMM (no slice-zero-add chains).
    """

    @staticmethod
    def forward(ctx, xp_tx, xp_bd, att_tt, bias_tt, att_tb, bias_tb, g_tt: EdgeGraph, g_tb: EdgeGraph,
                heads, channels, apply_gelu, slope, dropout_p, seed_tt, seed_tb, want_alpha, bits_tt=None, bits_tb=None):
        hc = heads * channels
        dev, dt = xp_tx.device, xp_tx.dtype
        need_grad = any(ctx.needs_input_grad[:6])
        xl_tt, xr_tt, xl_tb = xp_tx[:, :hc], xp_tx[:, hc:2 * hc], xp_tx[:, 2 * hc:]
        nt, nb = xp_tx.shape[0], xp_bd.shape[0]
        y_tx = torch.empty((nt, hc), dtype=dt, device=dev)
        y_bd = torch.empty((nb, hc), dtype=dt, device=dev)
        mk = lambda n: torch.empty((n, hc), dtype=dt, device=dev) if (need_grad and apply_gelu) else None
        pre_tx, pre_bd = mk(nt), mk(nb)
        lse_tx = torch.empty((nt, heads), dtype=torch.float32, device=dev) if need_grad else None
        lse_bd = torch.empty((nb, heads), dtype=torch.float32, device=dev) if need_grad else None
        alpha = torch.empty((g_tt.n_edges, heads), dtype=torch.float32, device=dev) if want_alpha else None
        # both edge types in ONE launch (segger_gatv2_fwd_pair): at segger's default batch size the tx-belongs-bd blocks
        # disappear inside the tx-neighbors-tx launch (46 -> 40 us per layer); at C2 it measures neutral
        gatv2_fwd_pair_launch(
            dict(by_dst=g_tt.by_dst, xl=xl_tt, xr=xr_tt, att=att_tt, bias=bias_tt, heads=heads, channels=channels, out=y_tx,
                 pre=pre_tx, lse=lse_tx, alpha=alpha, apply_gelu=apply_gelu, negative_slope=slope, dropout_p=dropout_p,
                 seed=seed_tt, keep_bits=None if bits_tt is None else bits_tt[0]),
            dict(by_dst=g_tb.by_dst, xl=xl_tb, xr=xp_bd, att=att_tb, bias=bias_tb, heads=heads, channels=channels, out=y_bd,
                 pre=pre_bd, lse=lse_bd, apply_gelu=apply_gelu, negative_slope=slope, dropout_p=dropout_p, seed=seed_tb,
                 keep_bits=None if bits_tb is None else bits_tb[0]))
        if need_grad:
            ctx.save_for_backward(xp_tx, xp_bd, att_tt, bias_tt, att_tb, bias_tb,
                                  pre_tx if apply_gelu else y_tx, pre_bd if apply_gelu else y_bd, lse_tx, lse_bd)
            ctx.graphs = (g_tt, g_tb)
            ctx.cfg = (heads, channels, apply_gelu, slope, dropout_p, seed_tt, seed_tb)
            ctx.bits = (bits_tt, bits_tb)
        if want_alpha:
            ctx.mark_non_differentiable(alpha)
        return y_tx, y_bd, alpha

    @staticmethod
    def backward(ctx, gy_tx, gy_bd, _ga):
        xp_tx, xp_bd, att_tt, bias_tt, att_tb, bias_tb, pre_tx, pre_bd, lse_tx, lse_bd = ctx.saved_tensors
        heads, channels, apply_gelu, slope, p, seed_tt, seed_tb = ctx.cfg
        g_tt, g_tb = ctx.graphs
        hc = heads * channels
        gxp_tx = torch.empty_like(xp_tx)
        gxp_bd = torch.empty_like(xp_bd)
        if gy_tx is None:
            gy_tx = torch.zeros_like(pre_tx)
        if gy_bd is None:
            gy_bd = torch.zeros_like(pre_bd)
        # tx-neighbors-tx first: its source pass visits every transcript row and zero-fills the tx-belongs-bd window of
        # the stacked projection gradient on the way, so the one-pass tx-belongs-bd backward needs no fill of its own.
        # (Running tx-belongs-bd on a second stream beside it was measured in round 3: the kernels do overlap, but the
        # small one then takes 10x longer and the big ones 3-7 % longer -- same total)
        # For small batches the library goes one step further (segger_gatv2_bwd_pair): the tx-neighbors-tx DESTINATION
        # pass does the zero fill and its source pass shares a launch with the tx-belongs-bd pass.
        (gatt_tt, gbias_tt), (gatt_tb, gbias_tb) = gatv2_bwd_pair_launch(
            (g_tt, xp_tx[:, :hc], xp_tx[:, hc:2 * hc], att_tt, bias_tt, heads, channels, gy_tx, pre_tx, lse_tx,
             gxp_tx[:, :hc], gxp_tx[:, hc:2 * hc]),
            dict(apply_gelu=apply_gelu, negative_slope=slope, dropout_p=p, seed=seed_tt, keep_bits=ctx.bits[0],
                 zero_rows_out=gxp_tx[:, 2 * hc:]),
            (g_tb, xp_tx[:, 2 * hc:], xp_bd, att_tb, bias_tb, heads, channels, gy_bd, pre_bd, lse_bd,
             gxp_tx[:, 2 * hc:], gxp_bd),
            dict(apply_gelu=apply_gelu, negative_slope=slope, dropout_p=p, seed=seed_tb, keep_bits=ctx.bits[1]))
        r = lambda gt, ref: gt.reshape(ref.shape).to(ref.dtype) if ref is not None else None
        return (gxp_tx, gxp_bd, r(gatt_tt, att_tt), r(gbias_tt, bias_tt), r(gatt_tb, att_tb), r(gbias_tb, bias_tb),
                None, None, None, None, None, None, None, None, None, None, None, None)


def hetero_gat_layer(xp_tx, xp_bd, att_tt, bias_tt, att_tb, bias_tb, g_tt, g_tb, heads, channels, *,
                     apply_gelu=True, negative_slope=0.2, dropout_p=0.0, seed_tt=0, seed_tb=0, return_alpha=False,
                     bits_tt=None, bits_tb=None):
    """``bits_tt`` / ``bits_tb`` = (by_dst plane, by_src plane or None) of :func:`dropout_bits` for this layer."""
    return _HeteroGatLayer.apply(xp_tx, xp_bd, att_tt, bias_tt, att_tb, bias_tb, g_tt, g_tb, heads, channels,
                                 apply_gelu, negative_slope, dropout_p, seed_tt, seed_tb, return_alpha, bits_tt, bits_tb)


# --------------------------------------------------------------------------
# Prediction head
# --------------------------------------------------------------------------
@torch.no_grad()
def edge_cos_argmax(by_src: EdgeCSR, z_src: Tensor, z_dst: Tensor, *, dst_index: Optional[Tensor] = None,
                    min_similarity: Optional[float] = None, eps: float = 1e-8, return_sim: bool = False):
    """-> (max_sim[f32 Ns], max_eid[i64 Ns], seg_idx[i64 Ns], sim[f32 E] | None)."""
    _lib.require_cuda(z_src, z_dst)
    lib = _lib.load()
    dev = z_src.device
    if z_src.dtype != z_dst.dtype or z_src.dtype not in DTYPE_CODE:
        raise TypeError("edge_cos_argmax: z_src / z_dst must share a supported dtype")
    c = int(z_src.shape[1])
    n = by_src.n_rows
    if z_src.shape[0] != n or z_dst.shape[0] != by_src.n_cols or z_dst.shape[1] != c:
        raise ValueError("edge_cos_argmax: embedding shapes do not match the graph")
    a = _lib.EdgeArgmaxArgs()
    a.by_src = by_src.c_struct()
    a.z_src, a.ld_zs = _rows(z_src, c, "z_src")
    a.z_dst, a.ld_zd = _rows(z_dst, c, "z_dst")
    a.channels, a.dtype, a.eps = c, DTYPE_CODE[z_src.dtype], eps
    a.use_min_similarity = int(min_similarity is not None)
    a.min_similarity = float(min_similarity) if min_similarity is not None else 0.0
    di = None
    if dst_index is not None:
        di = dst_index.to(device=dev, dtype=torch.int64).contiguous()
        if di.numel() != by_src.n_cols:
            raise ValueError("edge_cos_argmax: dst_index has the wrong length")
        a.dst_index = di.data_ptr()
    max_sim = torch.empty(n, dtype=torch.float32, device=dev)
    max_eid = torch.empty(n, dtype=torch.int64, device=dev)
    seg = torch.empty(n, dtype=torch.int64, device=dev)
    sim = torch.empty(by_src.n_edges, dtype=torch.float32, device=dev) if return_sim else None
    a.max_sim, a.max_eid, a.seg_idx, a.sim = max_sim.data_ptr(), max_eid.data_ptr(), seg.data_ptr(), _lib.ptr(sim)
    with _lib.on_device(dev):
        rc = lib.segger_edge_cos_argmax(C.byref(a), _lib.stream_ptr(dev))
    _lib.check(rc, "segger_edge_cos_argmax")
    return max_sim, max_eid, seg, sim


# --------------------------------------------------------------------------
# Triplet margin loss over edges
# --------------------------------------------------------------------------
def _triplet_args(src, pos, neg, za, zb, margin, eps, kind: str = "triplet"):
    a = _lib.TripletArgs()
    a.loss_kind = {"triplet": 0, "bce": 1}[kind]
    a.src, a.pos, a.neg, a.n_edges = src.data_ptr(), pos.data_ptr(), neg.data_ptr(), int(src.numel())
    c = int(za.shape[1])
    a.z_a, a.ld_za = _rows(za, c, "z_a")
    a.z_b, a.ld_zb = _rows(zb, c, "z_b")
    a.n_a, a.n_b = int(za.shape[0]), int(zb.shape[0])
    a.channels, a.dtype, a.margin, a.eps = c, DTYPE_CODE[za.dtype], margin, eps
    return a


_CONTRIB_MIN_EDGES = 16384      # (packed 16-bit atomics into a gradient of the embeddings' dtype from this many triplets on: no fp32 staging + cast)


class _TripletEdgeLoss(torch.autograd.Function):
    """``zb is None``: anchors, positives and negatives are rows of the same matrix ``za``
    (loss_tx); one gradient buffer receives all three contributions."""

    @staticmethod
    def forward(ctx, za, zb, src, pos, neg, margin, eps, pos_groups, anchors_unique=False, kind="triplet"):
        same = zb is None
        zb_ = za if same else zb
        _lib.require_cuda(za, zb_, src)
        lib = _lib.load()
        dev = za.device
        if za.dtype != zb_.dtype or za.dtype not in DTYPE_CODE:
            raise TypeError("triplet_edge_loss: z_a / z_b must share a supported dtype")
        src, pos, neg = (t.to(torch.int64).contiguous() for t in (src, pos, neg))
        a = _triplet_args(src, pos, neg, za, zb_, margin, eps, kind)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ws_bytes = lib.segger_triplet_workspace_bytes(a.n_edges)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        a.loss, a.workspace, a.workspace_bytes = loss.data_ptr(), ws.data_ptr(), ws_bytes
        with _lib.on_device(dev):
            rc = lib.segger_triplet_fwd(C.byref(a), _lib.stream_ptr(dev))
        _lib.check(rc, "segger_triplet_fwd")
        ctx.save_for_backward(za, zb_, src, pos, neg)
        ctx.cfg = (margin, eps, same)
        ctx.pos_groups, ctx.anchors_unique, ctx.kind = pos_groups, anchors_unique, kind
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        za, zb, src, pos, neg = ctx.saved_tensors
        margin, eps, same = ctx.cfg
        lib = _lib.load()
        dev = za.device
        a = _triplet_args(src, pos, neg, za, zb, margin, eps, ctx.kind)
        # anchor-matrix rows collect a handful of terms: packed 16-bit atomics straight into a gradient of the
        # embeddings' dtype; a separate (boundary) matrix sums dozens of terms per row and stays fp32
        # (small batches measured no gain from the packed variant: they keep fp32 atomics)
        packed = (za.dtype in (torch.bfloat16, torch.float16) and za.shape[1] % 2 == 0
                  and src.numel() >= _CONTRIB_MIN_EDGES)
        ga = torch.zeros(za.shape, dtype=za.dtype if packed else torch.float32, device=dev)
        a.grad_a, a.grad_a_packed, a.grad_b_packed = ga.data_ptr(), int(packed), int(packed and same)
        pg = ctx.pos_groups
        if same:
            gb = ga
            a.grad_b = ga.data_ptr()
        elif pg is not None:
            # boundary side: the positives -- a boundary's ~40 edges sit next to each other and would hammer one row
            # (0.58 ms of contended fp32 atomics at C2) -- are a segmented sum over the triplets grouped by positive
            # row (for tx-belongs-bd edges: the by-destination view the encoder already built), which also WRITES
            # every row of grad_b; the uniformly sampled negatives add themselves with (uncontended) fp32 atomics
            uniq = ctx.anchors_unique
            uniq = bool(uniq() if callable(uniq) else uniq) and za.shape[1] % 32 == 0 and za.shape[1] <= 128
            gb = (torch.zeros if (uniq or ctx.kind == "bce") else torch.empty)(zb.shape, dtype=torch.float32, device=dev)
            a.grad_b = gb.data_ptr()
            a.pos_indptr, a.pos_eid = pg.indptr.data_ptr(), (pg.eid.data_ptr() if pg.n_edges else None)
            a.anchor_unique = int(uniq)
        else:
            gb = torch.zeros(zb.shape, dtype=torch.float32, device=dev)
            a.grad_b = gb.data_ptr()
        gs = g.detach().to(torch.float32).reshape(1).contiguous()   # upstream scalar stays on the device
        a.grad_scale, a.grad_scale_dev = 1.0, gs.data_ptr()
        with _lib.on_device(dev):
            rc = lib.segger_triplet_bwd(C.byref(a), _lib.stream_ptr(dev))
        _lib.check(rc, "segger_triplet_bwd")
        return ga.to(za.dtype), (None if same else gb.to(zb.dtype)), None, None, None, None, None, None, None, None


def triplet_edge_loss(za: Tensor, zb: Optional[Tensor], src: Tensor, pos: Tensor, neg: Tensor,
                      margin: float, eps: float = 1e-6, pos_groups: Optional[EdgeCSR] = None,
                      anchors_unique=False) -> Tensor:
    """mean_e max(||za[src]-zb[pos]+eps|| - ||za[src]-zb[neg]+eps|| + margin, 0).
    Pass ``zb=None`` (or ``zb is za``) when positives / negatives index the anchor matrix itself.
    ``pos_groups``: the triplets grouped by positive row (``indptr`` over rows of ``zb``, ``eid`` = triplet ids), e.g.
    the by-destination view of the edge store the triplets come from: the backward then needs no atomics for them.
    ``anchors_unique`` (bool, or a callable asked in the backward, e.g. ``EdgeGraph.src_unique``; with ``pos_groups``):
    no row of ``za`` anchors two triplets -- the backward is then one walk over the groups storing the anchors' rows."""
    if zb is za:
        zb = None
    if pos_groups is not None and (zb is None or pos_groups.n_rows != zb.shape[0] or pos_groups.n_edges != src.numel()):
        raise ValueError("triplet_edge_loss: pos_groups does not describe these triplets")
    return _TripletEdgeLoss.apply(za, zb, src, pos, neg, float(margin), float(eps), pos_groups, anchors_unique, "triplet")


def bce_edge_loss(za: Tensor, zb: Tensor, src: Tensor, pos: Tensor, neg: Tensor, pos_groups: Optional[EdgeCSR] = None,
                  anchors_unique=False) -> Tensor:
    """The BCE variant of the segmentation loss (reference lightning_model.py:190-207): ``BCEWithLogitsLoss`` over
    ``cat(<za[src], zb[pos]>, <za[src], zb[neg]>)`` with labels ``cat(1, 0)``, one kernel forward and one backward
    (``segger_triplet_fwd / _bwd`` with ``loss_kind = SEGGER_LOSS_BCE``).  ``pos_groups`` / ``anchors_unique`` as in
    :func:`triplet_edge_loss`."""
    if za.shape[1] % 2:
        raise ValueError("bce_edge_loss: even channel count")
    if pos_groups is not None and (pos_groups.n_rows != zb.shape[0] or pos_groups.n_edges != src.numel()):
        raise ValueError("bce_edge_loss: pos_groups does not describe these edges")
    return _TripletEdgeLoss.apply(za, zb, src, pos, neg, 0.0, 0.0, pos_groups, anchors_unique, "bce")


class _MetricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, pos, neg, d_pos, d_neg, w, eps):
        _lib.require_cuda(z, pos)
        lib = _lib.load()
        dev = z.device
        if z.dtype not in DTYPE_CODE:
            raise TypeError("metric_loss: unsupported embedding dtype")
        n, c = z.shape
        zp, ld = _rows(z, c, "z")
        pos, neg = pos.to(torch.int64).contiguous(), neg.to(torch.int64).contiguous()
        d_pos, d_neg, w = (t.to(torch.float32).contiguous() for t in (d_pos, d_neg, w))
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ws_bytes = lib.segger_triplet_workspace_bytes(n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = lib.segger_metric_fwd(zp, ld, n, c, DTYPE_CODE[z.dtype], pos.data_ptr(), neg.data_ptr(), d_pos.data_ptr(),
                                       d_neg.data_ptr(), w.data_ptr(), eps, loss.data_ptr(), ws.data_ptr(), ws_bytes,
                                       _lib.stream_ptr(dev))
        _lib.check(rc, "segger_metric_fwd")
        ctx.save_for_backward(z, pos, neg, d_pos, d_neg, w)
        ctx.eps = eps
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        z, pos, neg, d_pos, d_neg, w = ctx.saved_tensors
        lib = _lib.load()
        dev = z.device
        n, c = z.shape
        zp, ld = _rows(z, c, "z")
        gz = torch.zeros((n, c), dtype=torch.float32, device=dev)
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        with _lib.on_device(dev):
            rc = lib.segger_metric_bwd(zp, ld, n, c, DTYPE_CODE[z.dtype], pos.data_ptr(), neg.data_ptr(), d_pos.data_ptr(),
                                       d_neg.data_ptr(), w.data_ptr(), ctx.eps, gs.data_ptr(), gz.data_ptr(),
                                       _lib.stream_ptr(dev))
        _lib.check(rc, "segger_metric_bwd")
        return gz.to(z.dtype), None, None, None, None, None, None


def metric_loss(z: Tensor, pos: Tensor, neg: Tensor, d_pos: Tensor, d_neg: Tensor, w: Tensor, eps: float = 1e-8) -> Tensor:
    """sum_i w_i [(cos(z_i, z_pos_i) - (1 - d_pos_i))^2 + (cos(z_i, z_neg_i) - (1 - d_neg_i))^2]: MetricLoss
    (triplet_loss.py:176-204) on sampled triplets, one kernel forward and one backward; ``pos_i < 0`` skips node i."""
    return _MetricLoss.apply(z, pos, neg, d_pos, d_neg, w, float(eps))


class LossHeadSpec:
    """What :func:`loss_head` needs besides the embeddings.  ``tx`` = (anchors, positives, negatives, margin, eps) of
    loss_tx (rows of z_tx; ``-1`` = skip); ``bd`` = (positives, negatives, d_pos, d_neg, weights, eps) of loss_bd (rows
    of z_bd); ``sg`` = (src, pos, neg, margin, eps, pos_groups | None[, anchors_unique]) of the segmentation triplets, or
    None when the batch has at most one boundary (lightning_model.py:173-175: that loss is then 0).  ``anchors_unique``
    (bool, or a callable asked in the backward, e.g. ``EdgeGraph.src_unique``): no transcript is the anchor of two
    triplets -- the backward then walks the groups once and stores the anchors' gradient rows."""

    def __init__(self, tx, bd, sg, sg_kind: str = "triplet", tx_anchors_are_rows: bool = False, sg_of_tx=None,
                 tx_state: Optional[Tensor] = None, grad_out_hint: Optional[Tensor] = None):
        self.tx, self.bd, self.sg, self.sg_kind = tx, bd, sg, sg_kind      # sg_kind "bce": margin / eps unused
        # loss_tx's anchors are arange(n_tx) (possibly with -1 positives = skipped): lets the backward STORE the anchors'
        # gradient rows (loss_head, when z_tx comes straight out of ops.l2_normalize)
        self.tx_anchors_are_rows = bool(tx_anchors_are_rows)
        # one-launch loss head (segger_loss_head_fwd / _bwd): ``sg_of_tx`` int32 [n_tx] = the segmentation triplet anchored
        # at each transcript row (-1: none; :func:`anchor_index`), or a callable returning it; ``tx_state`` int32
        # [2 n_tx + 4] a buffer the caller has ALREADY zero-filled (a captured step does that in its staging launch);
        # ``grad_out_hint`` float32 [4]: the gradient the backward will receive (a training step: e_3) -- the forward then
        # leaves the three scale factors behind and the backward skips their launch when it is handed that very tensor
        self.sg_of_tx, self.tx_state, self.grad_out_hint = sg_of_tx, tx_state, grad_out_hint
        # one-launch head, with grad_out_hint: leave the finishing launch (the three means, the total) to the BACKWARD launch --
        # the returned loss tensor is then only valid once the backward has run (a captured training step reads it after the
        # replay); the caller promises that a backward with exactly the hinted gradient follows
        self.defer_finish = False


class _LossHead(torch.autograd.Function):
    """``LitISTEncoder.get_losses`` (lightning_model.py:151-213) after the sampling, as ONE autograd node:
    out = [a0 * loss_tx, a1 * loss_bd, a2 * loss_sg, sum_i b_i * out_i].  Forward: the three loss kernels write their
    means side by side and one launch combines them.  Backward: one launch turns the incoming gradient into the three
    scale factors (device scalars), then the three backward kernels accumulate into ONE gradient buffer per embedding
    matrix -- no per-loss zero fill, cast and add, and no chain of 0-dim torch ops around the weighted sum."""

    @staticmethod
    def forward(ctx, z_tx, z_bd, a, b, spec: LossHeadSpec, y_tx=None, eps_tx=0.0):
        # y_tx (optional): z_tx == l2_normalize(y_tx, eps_tx) is then a CONSTANT here and the gradient is returned for y_tx
        # (anchor rows stored + normalisation backward inside this node)
        _lib.require_cuda(z_tx, z_bd, a, b)
        lib = _lib.load()
        dev, dt = z_tx.device, z_tx.dtype
        if z_bd.dtype != dt or dt not in DTYPE_CODE:
            raise TypeError("loss_head: z_tx / z_bd must share a supported dtype")
        c = int(z_tx.shape[1])
        if z_bd.shape[1] != c:
            raise ValueError("loss_head: z_tx / z_bd must have the same width")
        a = a.detach().to(torch.float32).contiguous()
        b = b.detach().to(torch.float32).contiguous()
        i64 = lambda t: t.to(torch.int64).contiguous()
        keep = []
        stream = _lib.stream_ptr(dev)
        parts = (C.c_void_p * 3)()
        counts = (C.c_int64 * 3)(0, 0, 0)
        scales = (C.c_float * 3)(0.0, 0.0, 0.0)
        with _lib.on_device(dev):
            anchors, pos, neg, margin, eps = spec.tx
            tx = tuple(i64(t) for t in (anchors, pos, neg))
            ta = _triplet_args(*tx, z_tx, z_tx, float(margin), float(eps))
            ws = torch.empty(lib.segger_triplet_workspace_bytes(ta.n_edges), dtype=torch.uint8, device=dev)
            ta.loss, ta.workspace, ta.workspace_bytes = None, ws.data_ptr(), ws.numel()     # partial sums only
            _lib.check(lib.segger_triplet_fwd(C.byref(ta), stream), "segger_triplet_fwd")
            keep.append(ws)
            if ta.n_edges:
                parts[0], counts[0], scales[0] = ws.data_ptr(), lib.segger_triplet_partial_count(ta.n_edges), 1.0 / ta.n_edges
            bpos, bneg, dp, dn, w, beps = spec.bd
            bd = (i64(bpos), i64(bneg)) + tuple(t.to(torch.float32).contiguous() for t in (dp, dn, w))
            nb = int(z_bd.shape[0])
            zp, ld = _rows(z_bd, c, "z_bd")
            ws = torch.empty(lib.segger_triplet_workspace_bytes(nb), dtype=torch.uint8, device=dev)
            _lib.check(lib.segger_metric_fwd(zp, ld, nb, c, DTYPE_CODE[dt], bd[0].data_ptr(), bd[1].data_ptr(),
                                             bd[2].data_ptr(), bd[3].data_ptr(), bd[4].data_ptr(), float(beps),
                                             None, ws.data_ptr(), ws.numel(), stream), "segger_metric_fwd")
            keep.append(ws)
            if nb:
                parts[1], counts[1], scales[1] = ws.data_ptr(), lib.segger_triplet_partial_count(nb), 1.0
            sg = None
            if spec.sg is not None:
                src, spos, sneg, smargin, seps, pg = spec.sg[:6]
                sg = tuple(i64(t) for t in (src, spos, sneg))
                if pg is not None and (pg.n_rows != nb or pg.n_edges != sg[0].numel()):
                    raise ValueError("loss_head: pos_groups does not describe the segmentation triplets")
                sa = _triplet_args(*sg, z_tx, z_bd, float(smargin), float(seps), spec.sg_kind)
                ws = torch.empty(lib.segger_triplet_workspace_bytes(sa.n_edges), dtype=torch.uint8, device=dev)
                sa.loss, sa.workspace, sa.workspace_bytes = None, ws.data_ptr(), ws.numel()
                _lib.check(lib.segger_triplet_fwd(C.byref(sa), stream), "segger_triplet_fwd")
                keep.append(ws)
                if sa.n_edges:
                    parts[2], counts[2] = ws.data_ptr(), lib.segger_triplet_partial_count(sa.n_edges)
                    scales[2] = (0.5 if spec.sg_kind == "bce" else 1.0) / sa.n_edges
            out = torch.empty(4, dtype=torch.float32, device=dev)
            # the three means from their per-block partial sums and the weighted total: one launch
            _lib.check(lib.segger_loss_combine_partials_fwd(parts, counts, scales, a.data_ptr(), b.data_ptr(), 3,
                                                            out.data_ptr(), stream), "segger_loss_combine_partials_fwd")
        ctx.keep = keep                                      # (the partial sums are read by the launch above)
        ctx.save_for_backward(z_tx, z_bd, a, b, *tx, *bd, *(sg or ()))
        ctx.spec, ctx.y_tx, ctx.eps_tx = spec, y_tx, float(eps_tx)
        return out

    @staticmethod
    def backward(ctx, g_out):
        z_tx, z_bd, a, b = ctx.saved_tensors[:4]
        tx = ctx.saved_tensors[4:7]
        bd = ctx.saved_tensors[7:12]
        sg = ctx.saved_tensors[12:15] if ctx.spec.sg is not None else None
        spec = ctx.spec
        lib = _lib.load()
        dev, dt = z_tx.device, z_tx.dtype
        c = int(z_tx.shape[1])
        nb = int(z_bd.shape[0])
        g_out = g_out.detach().to(torch.float32).contiguous()
        graw = torch.empty(3, dtype=torch.float32, device=dev)
        # transcript rows collect a handful of terms (once as anchor of either loss, ~2 as positive / negative): packed
        # 16-bit atomics straight into a gradient of the embeddings' dtype on large batches; boundary rows sum dozens
        # of terms and stay fp32 (+ one cast)
        packed = dt in (torch.bfloat16, torch.float16) and c % 2 == 0 and tx[0].numel() >= _CONTRIB_MIN_EDGES
        y_tx = ctx.y_tx
        pg = spec.sg[5] if spec.sg is not None else None
        uniq = False
        if pg is not None and len(spec.sg) > 6 and c % 32 == 0 and c <= 128:
            uniq = spec.sg[6]
            uniq = bool(uniq() if callable(uniq) else uniq)
        gb_written = pg is not None and not uniq and spec.sg_kind != "bce"       # (the two-kernel route writes every row)
        # both gradient matrices out of ONE zero-filled buffer (one fill launch instead of two)
        ga_dt = dt if packed else torch.float32
        na = z_tx.numel() * (2 if packed else 4)
        na = (na + 255) // 256 * 256
        nbb = 0 if gb_written else z_bd.numel() * 4
        zbuf = torch.zeros(na + nbb, dtype=torch.uint8, device=dev)
        ga = zbuf[:z_tx.numel() * (2 if packed else 4)].view(ga_dt).view(z_tx.shape)
        gb = (torch.empty(z_bd.shape, dtype=torch.float32, device=dev) if gb_written
              else zbuf[na:].view(torch.float32).view(z_bd.shape))
        ga_rows = torch.empty_like(ga) if y_tx is not None else None      # every row written by loss_tx's kernel
        stream = _lib.stream_ptr(dev)
        with _lib.on_device(dev):
            _lib.check(lib.segger_loss_combine_bwd(g_out.data_ptr(), a.data_ptr(), b.data_ptr(), 3, graw.data_ptr(), stream),
                       "segger_loss_combine_bwd")
            if sg is not None:          # first: with pos_groups its positive side WRITES every row of gb
                sa = _triplet_args(*sg, z_tx, z_bd, float(spec.sg[3]), float(spec.sg[4]), spec.sg_kind)
                sa.grad_a, sa.grad_a_packed, sa.grad_b, sa.grad_b_packed = ga.data_ptr(), int(packed), gb.data_ptr(), 0
                if pg is not None:
                    sa.pos_indptr, sa.pos_eid = pg.indptr.data_ptr(), (pg.eid.data_ptr() if pg.n_edges else None)
                    sa.anchor_unique = int(uniq)      # one walk over the groups: anchor rows stored, not added
                sa.grad_scale, sa.grad_scale_dev = 1.0, graw[2:3].data_ptr()
                _lib.check(lib.segger_triplet_bwd(C.byref(sa), stream), "segger_triplet_bwd")
            zp, ld = _rows(z_bd, c, "z_bd")
            _lib.check(lib.segger_metric_bwd(zp, ld, nb, c, DTYPE_CODE[dt], bd[0].data_ptr(), bd[1].data_ptr(),
                                             bd[2].data_ptr(), bd[3].data_ptr(), bd[4].data_ptr(), float(spec.bd[5]),
                                             graw[1:2].data_ptr(), gb.data_ptr(), stream), "segger_metric_bwd")
            ta = _triplet_args(*tx, z_tx, z_tx, float(spec.tx[3]), float(spec.tx[4]))
            ta.grad_a = ta.grad_b = ga.data_ptr()
            ta.grad_a_packed = ta.grad_b_packed = int(packed)
            ta.grad_scale, ta.grad_scale_dev = 1.0, graw[0:1].data_ptr()
            if ga_rows is not None:
                ta.grad_a_rows = ga_rows.data_ptr()
            _lib.check(lib.segger_triplet_bwd(C.byref(ta), stream), "segger_triplet_bwd")
            if ga_rows is not None:
                # d loss / d y_tx = normalisation backward of (ga + ga_rows), read as two matrices
                g1, g2 = (ga, ga_rows) if ga.dtype == dt else (ga.to(dt), ga_rows.to(dt))
                gy = torch.empty_like(y_tx)
                yp, ldy = _rows(y_tx, c, "y_tx")
                _lib.check(lib.segger_l2norm_bwd2(yp, ldy, g1.data_ptr(), c, g2.data_ptr(), c, int(y_tx.shape[0]), c, ctx.eps_tx,
                                                  gy.data_ptr(), c, DTYPE_CODE[dt], stream), "segger_l2norm_bwd2")
                return None, gb.to(dt), None, None, None, gy, None
        return (ga if packed else ga.to(dt)), gb.to(dt), None, None, None, None, None


_TICKETS: dict = {}


def _ticket(dev) -> Tensor:
    """The zero-initialised int32 the one-launch loss head counts finished blocks in (every launch leaves it zero);
    one per device, never freed: captured graphs hold its address."""
    t = _TICKETS.get(dev)
    if t is None:
        t = _TICKETS[dev] = torch.zeros(16, dtype=torch.int32, device=dev)
    return t


@torch.no_grad()
def anchor_index(src: Tensor, n_rows: int) -> Tensor:
    """int32 [n_rows]: position in ``src`` of each row (-1: the row is not in ``src``).  For the tx-belongs-bd edge list
    (every transcript at most once, heterodata.py:147) this is "the segmentation triplet anchored at transcript r"."""
    inv = torch.full((int(n_rows),), -1, dtype=torch.int32, device=src.device)
    if src.numel():
        inv[src.long()] = torch.arange(src.numel(), dtype=torch.int32, device=src.device)
    return inv


ONE_LAUNCH_LOSS_HEAD = True      # tools flip it: False = the round-3 loss head (three kernels + combination each way)
# The one-launch head for batches up to this many transcript rows.  Round 4 kept large batches on the kernel-by-kernel head: at
# 10^6 rows its forward (chains threaded with 4 returning atomics per triplet: 250 us against 150 us) and its gathered backward
# (-70..130 us) tied in bf16 (tools/bench_loss_head.py, profiles/r04_loss_head_modes_c2.txt).  Round 5 measured the whole C2
# step (tools/bench_step.py, alternating variants in one process): bf16 12.93-13.08 -> 12.88-12.93 ms, and at fp32 storage --
# where the kernel-by-kernel backward adds 128 M fp32 atomics -- 27.4-27.6 -> 26.5 ms: the one-launch head at every size the
# kernel's 30-bit row ids allow.
LOSS_HEAD_ONE_LAUNCH_MAX_ROWS = (1 << 30) - 2


def loss_head_fused_supported(z_tx: Tensor, z_bd: Tensor, spec: "LossHeadSpec") -> bool:
    c = int(z_tx.shape[1])
    n = int(z_tx.shape[0])
    return (ONE_LAUNCH_LOSS_HEAD and spec.tx_anchors_are_rows and z_tx.dtype in DTYPE_CODE and z_tx.dtype == z_bd.dtype
            and bool(_lib.load().segger_loss_head_supported(c, DTYPE_CODE[z_tx.dtype])) and int(z_bd.shape[1]) == c
            and 0 < n <= LOSS_HEAD_ONE_LAUNCH_MAX_ROWS and spec.tx[0].numel() == n and z_bd.shape[0] > 0
            and (spec.sg is None or (len(spec.sg) > 5 and spec.sg[5] is not None)))


class _LossHeadFused(torch.autograd.Function):
    """``LitISTEncoder.get_losses`` after the sampling as ONE launch forward and ONE backward (``segger_loss_head_fwd`` /
    ``_bwd``, csrc/loss_head.hip).  ``t_tx`` / ``t_bd`` carry the gradient: the embeddings themselves, or -- with
    ``z_tx`` / ``z_bd`` given as constants -- the matrices they were normalised from (``z = t / max(|t|, eps)``): the
    transcript side's normalisation backward then happens inside the launch, the boundary side's in a second, tiny one
    that reads the fp32 gradient directly.  No float atomic touches the transcript matrix (rows are gathered, not
    scattered), nothing is zero-filled by a launch of its own."""

    @staticmethod
    def forward(ctx, t_tx, t_bd, a, b, spec: LossHeadSpec, z_tx=None, z_bd=None, eps_tx=0.0, eps_bd=0.0):
        prenorm = z_tx is not None
        if not prenorm:
            z_tx, z_bd = t_tx, t_bd
        _lib.require_cuda(z_tx, z_bd, a, b)
        lib = _lib.load()
        dev, dt = z_tx.device, z_tx.dtype
        n_tx, c = int(z_tx.shape[0]), int(z_tx.shape[1])
        n_bd = int(z_bd.shape[0])
        need = any(ctx.needs_input_grad[:2])
        a = a.detach().to(torch.float32).contiguous()
        b = b.detach().to(torch.float32).contiguous()
        i64 = lambda t: t.to(torch.int64).contiguous()
        f32 = lambda t: t.to(torch.float32).contiguous()
        _, pos, neg, margin, eps = spec.tx
        tx = (i64(pos), i64(neg))
        bpos, bneg, dp, dn, w, beps = spec.bd
        bd = (i64(bpos), i64(bneg), f32(dp), f32(dn), f32(w))
        sg = None
        if spec.sg is not None:
            sg = tuple(i64(t) for t in spec.sg[:3])
        g = _lib.LossHeadArgs()
        g.z_tx, g.ld_ztx = _rows(z_tx, c, "z_tx")
        g.z_bd, g.ld_zbd = _rows(z_bd, c, "z_bd")
        g.n_tx, g.n_bd, g.channels, g.dtype = n_tx, n_bd, c, DTYPE_CODE[dt]
        g.tx_pos, g.tx_neg, g.tx_margin, g.tx_eps = tx[0].data_ptr(), tx[1].data_ptr(), float(margin), float(eps)
        g.bd_pos, g.bd_neg, g.bd_dpos, g.bd_dneg, g.bd_w = (t.data_ptr() for t in bd)
        g.bd_eps = float(beps)
        g.sg_kind = {"triplet": 0, "bce": 1}[spec.sg_kind]
        if sg is not None and sg[0].numel():
            g.sg_src, g.sg_pos, g.sg_neg, g.n_sg = sg[0].data_ptr(), sg[1].data_ptr(), sg[2].data_ptr(), int(sg[0].numel())
            g.sg_margin, g.sg_eps = float(spec.sg[3]), float(spec.sg[4])
        g.a, g.b = a.data_ptr(), b.data_ptr()
        out = torch.empty(4, dtype=torch.float32, device=dev)
        graw = torch.empty(3, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.segger_loss_head_workspace_bytes(n_tx, n_bd, int(g.n_sg)), dtype=torch.uint8, device=dev)
        g.out, g.grad_raw, g.workspace, g.workspace_bytes = out.data_ptr(), graw.data_ptr(), ws.data_ptr(), ws.numel()
        g.ticket = _ticket(dev).data_ptr()
        hint = spec.grad_out_hint
        state = None
        if need:
            tx_w = torch.empty((n_tx, 2), dtype=torch.float32, device=dev)
            head = spec.tx_state
            if head is None:
                head = torch.zeros(2 * n_tx + 4, dtype=torch.int32, device=dev)
            elif head.dtype != torch.int32 or head.numel() < 2 * n_tx + 4 or not head.is_contiguous():
                raise ValueError("loss_head: tx_state must be a contiguous int32 [2 n_tx + 4] tensor (zero-filled)")
            n_hot = int(lib.segger_loss_head_max_hot_rows(n_tx))
            nxt = torch.empty(2 * n_tx + n_tx + n_hot, dtype=torch.int32, device=dev)        # chain links | hot ids + arrivals
            hot_acc = torch.empty((n_hot, c), dtype=torch.float32, device=dev)
            gbd = torch.empty((n_bd, c), dtype=torch.float32, device=dev)
            g.tx_w, g.tx_state, g.tx_next, g.grad_bd = tx_w.data_ptr(), head.data_ptr(), nxt.data_ptr(), gbd.data_ptr()
            g.tx_hot_id, g.tx_hot_acc = nxt[2 * n_tx:].data_ptr(), hot_acc.data_ptr()
            if hint is not None:
                hint = hint.detach()
                if hint.dtype != torch.float32 or hint.numel() != 4 or not hint.is_contiguous() or hint.device != dev:
                    raise ValueError("loss_head: grad_out_hint must be a contiguous float32 [4] tensor on the embeddings' device")
                g.grad_out = hint.data_ptr()
                if getattr(spec, "defer_finish", False):
                    g.reserved_ = 1                      # SEGGER_LOSS_HEAD_DEFER_FINISH
            state = (tx_w, head, nxt, gbd, graw, hot_acc, out, ws, int(g.reserved_))
        with _lib.on_device(dev):
            rc = lib.segger_loss_head_fwd(C.byref(g), _lib.stream_ptr(dev))
        _lib.check(rc, "segger_loss_head_fwd")
        if need:
            ctx.save_for_backward(t_tx, t_bd, z_tx, z_bd, a, b, *tx, *bd, *(sg or ()))
            ctx.spec, ctx.state, ctx.prenorm, ctx.eps = spec, state, prenorm, (float(eps_tx), float(eps_bd))
            ctx.hint = hint if need else None
        return out

    @staticmethod
    def backward(ctx, g_out):
        t_tx, t_bd, z_tx, z_bd, a, b = ctx.saved_tensors[:6]
        tx = ctx.saved_tensors[6:8]
        bd = ctx.saved_tensors[8:13]
        spec = ctx.spec
        sg = ctx.saved_tensors[13:16] if spec.sg is not None else None
        tx_w, head, nxt, gbd, graw, hot_acc, out_fwd, ws_fwd, deferred = ctx.state
        lib = _lib.load()
        dev, dt = z_tx.device, z_tx.dtype
        n_tx, c = int(z_tx.shape[0]), int(z_tx.shape[1])
        n_bd = int(z_bd.shape[0])
        if getattr(ctx, "_ran", False):
            # a second backward over the same graph (retain_graph=True): the forward launch zero-filled the boundary
            # gradient and armed the hot rows' accumulators / arrival counters ONCE -- re-arm them, or this pass would add
            # on top of the first one's sums and never finish a hot row
            gbd.zero_()
            hot_acc.zero_()
            nxt[3 * n_tx:].zero_()
        ctx._ran = True
        stream = _lib.stream_ptr(dev)
        g_out = g_out.detach().to(torch.float32).contiguous()
        # the segmentation triplets' anchor terms ride in the row walk when no transcript anchors two of them
        pg = spec.sg[5] if spec.sg is not None else None
        uniq, of_tx = False, None
        if sg is not None and sg[0].numel():
            uniq = spec.sg[6] if len(spec.sg) > 6 else False
            uniq = bool(uniq() if callable(uniq) else uniq)
            if uniq:
                of_tx = spec.sg_of_tx() if callable(spec.sg_of_tx) else spec.sg_of_tx
                if of_tx is None:
                    of_tx = anchor_index(sg[0], n_tx)
                if of_tx.dtype != torch.int32 or of_tx.numel() < n_tx or not of_tx.is_contiguous():
                    raise ValueError("loss_head: sg_of_tx must be a contiguous int32 [n_tx] tensor")
        fuse_norm = ctx.prenorm and (sg is None or not sg[0].numel() or uniq)
        g = _lib.LossHeadArgs()
        g.z_tx, g.ld_ztx = _rows(z_tx, c, "z_tx")
        g.z_bd, g.ld_zbd = _rows(z_bd, c, "z_bd")
        g.n_tx, g.n_bd, g.channels, g.dtype = n_tx, n_bd, c, DTYPE_CODE[dt]
        g.tx_pos, g.tx_neg, g.tx_margin, g.tx_eps = tx[0].data_ptr(), tx[1].data_ptr(), float(spec.tx[3]), float(spec.tx[4])
        g.bd_pos, g.bd_neg, g.bd_dpos, g.bd_dneg, g.bd_w = (t.data_ptr() for t in bd)
        g.bd_eps = float(spec.bd[5])
        g.sg_kind = {"triplet": 0, "bce": 1}[spec.sg_kind]
        if sg is not None and sg[0].numel():
            if pg.n_rows != n_bd or pg.n_edges != sg[0].numel():
                raise ValueError("loss_head: pos_groups does not describe the segmentation triplets")
            g.sg_src, g.sg_pos, g.sg_neg, g.n_sg = sg[0].data_ptr(), sg[1].data_ptr(), sg[2].data_ptr(), int(sg[0].numel())
            g.sg_margin, g.sg_eps = float(spec.sg[3]), float(spec.sg[4])
            g.sg_pos_indptr, g.sg_pos_eid = pg.indptr.data_ptr(), (pg.eid.data_ptr() if pg.n_edges else None)
            g.sg_of_tx = _lib.ptr(of_tx)
        g.a, g.b, g.grad_raw = a.data_ptr(), b.data_ptr(), graw.data_ptr()
        out_dummy = torch.empty(4, dtype=torch.float32, device=dev)
        g.out = out_dummy.data_ptr()
        if deferred:
            if ctx.hint is None or g_out.data_ptr() != ctx.hint.data_ptr():
                raise RuntimeError("loss_head: defer_finish promised a backward with the hinted gradient")
            # the forward left only its per-block partial sums: this launch finishes the losses (into the forward's output)
            g.out, g.grad_out, g.reserved_ = out_fwd.data_ptr(), ctx.hint.data_ptr(), 1
        g.tx_w, g.tx_state, g.tx_next, g.grad_bd = tx_w.data_ptr(), head.data_ptr(), nxt.data_ptr(), gbd.data_ptr()
        g.tx_hot_id, g.tx_hot_acc = nxt[2 * n_tx:].data_ptr(), hot_acc.data_ptr()
        ws = ws_fwd if deferred else torch.empty(lib.segger_loss_head_workspace_bytes(n_tx, n_bd, int(g.n_sg)), dtype=torch.uint8,
                                                 device=dev)
        g.workspace, g.workspace_bytes, g.ticket = ws.data_ptr(), ws.numel(), _ticket(dev).data_ptr()
        gtx = torch.empty((n_tx, c), dtype=dt, device=dev)
        g.grad_tx, g.ld_gtx = gtx.data_ptr(), c
        if fuse_norm:
            g.y_tx, g.ld_ytx = _rows(t_tx, c, "y_tx")
            g.norm_eps = ctx.eps[0]
        with _lib.on_device(dev):
            if ctx.hint is None or g_out.data_ptr() != ctx.hint.data_ptr():      # (else the forward left the factors behind)
                _lib.check(lib.segger_loss_combine_bwd(g_out.data_ptr(), a.data_ptr(), b.data_ptr(), 3, graw.data_ptr(), stream),
                           "segger_loss_combine_bwd")
            _lib.check(lib.segger_loss_head_bwd(C.byref(g), stream), "segger_loss_head_bwd")
            if sg is not None and sg[0].numel() and not uniq:
                # a transcript anchors two segmentation triplets (never in segger's data): their anchor terms by atomics
                # on top of the rows just written, the boundary side having been done by the launch above
                sa = _triplet_args(*sg, z_tx, z_bd, float(spec.sg[3]), float(spec.sg[4]), spec.sg_kind)
                scratch = torch.zeros((n_bd, c), dtype=torch.float32, device=dev)      # (its boundary side is discarded)
                sa.grad_a, sa.grad_a_packed = gtx.data_ptr(), int(dt != torch.float32)
                sa.grad_b, sa.grad_b_packed = scratch.data_ptr(), 0
                sa.grad_scale, sa.grad_scale_dev = 1.0, graw[2:3].data_ptr()
                _lib.check(lib.segger_triplet_bwd(C.byref(sa), stream), "segger_triplet_bwd")
            if ctx.prenorm:
                # boundary side (and, in the rare case above, the transcript side) through the normalisation's backward
                gy_bd = torch.empty((n_bd, c), dtype=dt, device=dev)
                segs = (_lib.L2NormSeg * 2)()
                n_seg = 0
                yb, ldb = _rows(t_bd, c, "y_bd")
                segs[0].y, segs[0].ld_y, segs[0].n, segs[0].out, segs[0].ld_out = yb, ldb, n_bd, gy_bd.data_ptr(), c
                segs[0].gz, segs[0].ld_gz, segs[0].gz_f32 = gbd.data_ptr(), c, 1
                n_seg = 1
                if not fuse_norm:
                    gy_tx = torch.empty((n_tx, c), dtype=dt, device=dev)
                    yt, ldt = _rows(t_tx, c, "y_tx")
                    segs[1].y, segs[1].ld_y, segs[1].n, segs[1].out, segs[1].ld_out = yt, ldt, n_tx, gy_tx.data_ptr(), c
                    segs[1].gz, segs[1].ld_gz, segs[1].gz_f32 = gtx.data_ptr(), c, 0
                    n_seg = 2
                    gtx = gy_tx
                # (the two eps are the same number in the encoder; the launch takes one)
                _lib.check(lib.segger_l2norm_many(segs, n_seg, c, ctx.eps[1], DTYPE_CODE[dt], stream), "segger_l2norm_many")
                return gtx, gy_bd, None, None, None, None, None, None, None
        return gtx, gbd.to(dt), None, None, None, None, None, None, None


USE_ANCHOR_ROWS = True       # tools flip it: False = loss_tx's anchor terms by atomics at fp32 storage as well


def loss_head(z_tx: Tensor, z_bd: Tensor, a: Tensor, b: Tensor, spec: LossHeadSpec) -> Tensor:
    """-> float32[4] = (a0 * loss_tx, a1 * loss_bd, a2 * loss_sg, sum_i b_i * (the three)): the three losses of
    ``LitISTEncoder.get_losses`` and their weighted sum as one autograd node (see :class:`_LossHead`).  ``a`` / ``b``:
    float32[3] on the device."""
    if loss_head_fused_supported(z_tx, z_bd, spec):
        pt, pb = getattr(z_tx, "_segger_prenorm", None), getattr(z_bd, "_segger_prenorm", None)
        ok = lambda z, p: (p is not None and p[0].requires_grad and p[0].shape == z.shape and p[0].dtype == z.dtype)
        if torch.is_grad_enabled() and ok(z_tx, pt) and ok(z_bd, pb) and pt[1] == pb[1]:
            # both embeddings come straight out of the row normalisation: they are constants here and the gradient goes
            # to its inputs (the transcript side's normalisation backward inside the launch)
            return _LossHeadFused.apply(pt[0], pb[0], a, b, spec, z_tx.detach(), z_bd.detach(), pt[1], pb[1])
        return _LossHeadFused.apply(z_tx, z_bd, a, b, spec)
    pre = getattr(z_tx, "_segger_prenorm", None)
    # (fp32 storage only: there loss_tx's backward is bound by 12 fp32 atomic instructions per triplet, a third of them the
    # anchor's -- 1.19 -> 0.8 ms at C2; with 16-bit embeddings the packed atomics are cheap enough that the second matrix
    # the normalisation backward then reads costs what the stores save: 13.40 vs 13.42 ms per step)
    if (USE_ANCHOR_ROWS and z_tx.dtype == torch.float32
            and spec.tx_anchors_are_rows and pre is not None and pre[0].requires_grad and torch.is_grad_enabled()
            and z_tx.shape[1] % 8 == 0 and pre[0].shape == z_tx.shape and pre[0].dtype == z_tx.dtype
            and spec.tx[0].numel() == z_tx.shape[0]):
        # z_tx is the output of ops.l2_normalize: treat it as a constant and send the gradient to its input -- the anchors'
        # rows are then stored, not added atomically, and the normalisation backward reads the two matrices
        return _LossHead.apply(z_tx.detach(), z_bd, a, b, spec, pre[0], pre[1])
    return _LossHead.apply(z_tx, z_bd, a, b, spec)


@torch.no_grad()
def triplet_sample(index: dict, uniforms=None, seed_dev: Optional[Tensor] = None, seed: Optional[int] = None):
    """``FastTripletSelector.sample_triplets`` in one launch (``segger_triplet_sample``) from the selector's index
    (``triplet_loss.FastTripletSelector.build_index``).  ``uniforms``: four [n] tensors, or None for the kernel's own
    counter-based U[0,1) stream, seeded from torch's CPU generator (so ``torch.manual_seed`` still fixes a run) or from
    ``seed`` + the device word ``seed_dev`` (read at run time: a captured hipGraph draws afresh on every replay)."""
    lab = index["lab"]
    _lib.require_cuda(lab)
    lib = _lib.load()
    dev = lab.device
    n = int(lab.numel())
    u = None
    if uniforms is not None:
        u = torch.stack([t.to(device=dev, dtype=torch.float32) for t in uniforms]).contiguous()
        seed = 0
    elif seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # CPU generator: no kernel, no sync
    pos = torch.empty(n, dtype=torch.int64, device=dev)
    neg = torch.empty(n, dtype=torch.int64, device=dev)
    dd = torch.empty((2, n), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = lib.segger_triplet_sample(lab.data_ptr(), n, int(index["n_clusters"]), index["cdf_pos_t"].data_ptr(),
                                       index["cdf_neg_t"].data_ptr(), index["counts"].data_ptr(),
                                       index["offsets"].data_ptr(), index["members"].data_ptr(), _lib.ptr(u),
                                       int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(seed_dev),
                                       index["dists"].data_ptr(), pos.data_ptr(), neg.data_ptr(), dd[0].data_ptr(),
                                       dd[1].data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "segger_triplet_sample")
    return pos, neg, dd[0], dd[1]


@torch.no_grad()
def sample_negatives(pos: Tensor, n_b: int, n_b_dev: Optional[Tensor] = None, seed: Optional[int] = None,
                     seed_dev: Optional[Tensor] = None) -> Tensor:
    """``(pos + randint(1, n_b)) % n_b`` (the segmentation loss's negatives, lightning_model.py:178-180) in one launch
    (``segger_sample_negatives``); entries with ``pos < 0`` stay ``-1``.  The stream is seeded like
    :func:`triplet_sample`; ``n_b_dev`` (int64[1] on the device) overrides ``n_b`` at run time."""
    _lib.require_cuda(pos)
    lib = _lib.load()
    dev = pos.device
    pos = pos.to(torch.int64).contiguous()
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    neg = torch.empty_like(pos)
    with _lib.on_device(dev):
        rc = lib.segger_sample_negatives(pos.data_ptr(), int(pos.numel()), int(n_b), _lib.ptr(n_b_dev),
                                         int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(seed_dev), neg.data_ptr(),
                                         _lib.stream_ptr(dev))
    _lib.check(rc, "segger_sample_negatives")
    return neg


_FILLS = {"const": _lib.FILL_CONST, "tile": _lib.FILL_TILE, "div": _lib.FILL_DIV, "mod": _lib.FILL_MOD,
          "ramp": _lib.FILL_RAMP}


def _adam_jobs(opt):
    """[(param group, [(param, grad, exp_avg, exp_avg_sq, step)])] when ``segger_adam_step`` covers ``opt``, else None."""
    from .optim import Adam as _Adam
    if type(opt) not in (torch.optim.Adam, _Adam):   # (not subclasses in general: torch's AdamW is one)
        return None
    # torch's AMP contract for fused optimizers (``_step_supports_amp_scaling``): ``GradScaler.step`` skips its own unscale /
    # inf check and hands both to the optimizer as ``grad_scale`` / ``found_inf``.  The kernel reads neither: torch's fused
    # step does the scaled, skippable update.
    if getattr(opt, "grad_scale", None) is not None or getattr(opt, "found_inf", None) is not None:
        return None
    jobs = []
    for g in opt.param_groups:
        if (g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") or g.get("differentiable")
                or not g.get("capturable") or isinstance(g["lr"], Tensor)):
            return None
        rows = []
        for p in g["params"]:
            if p.grad is None:
                continue
            st = opt.state.get(p)
            if not st or "exp_avg" not in st:
                return None
            ts = (p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"])
            if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts) or p.grad.is_sparse:
                return None
            rows.append(ts)
        jobs.append((g, rows))
    return jobs


def adam_step_counters(opt, params=None) -> Optional[list]:
    """The step counters ``segger_adam_step`` would advance (one fp32 scalar per parameter ``params`` -- default: those with
    a gradient -- of a covered optimizer whose state exists), at most 64 in ONE parameter group; None otherwise.  A captured
    step hands them to :func:`step_draws` and then calls ``adam_step(opt, steps_advanced=True)``."""
    from .optim import Adam as _Adam
    if type(opt) not in (torch.optim.Adam, _Adam) or len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    if (g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") or g.get("differentiable")
            or not g.get("capturable") or isinstance(g["lr"], Tensor)):
        return None
    out = []
    for p in (g["params"] if params is None else params):
        st = opt.state.get(p)
        if not st or "step" not in st or not st["step"].is_cuda or st["step"].dtype != torch.float32:
            return None
        out.append(st["step"])
    return out if 0 < len(out) <= 64 else None


def double_bits(x: float) -> int:
    """The int64 whose bits are the fp64 pattern of ``x`` (a ``const`` fill of a float64 buffer by :func:`stage`)."""
    import struct
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def adam_hyper(opt) -> Optional[tuple]:
    """(lr, beta1, beta2, eps) of a one-group optimizer of the kind :func:`adam_step` covers (plain capturable Adam, Python
    float learning rate; the state need not exist yet), else None."""
    from .optim import Adam as _Adam
    if type(opt) not in (torch.optim.Adam, _Adam) or len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    if (g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") or g.get("differentiable")
            or not g.get("capturable") or isinstance(g["lr"], Tensor)):
        return None
    return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))


def adam_step(opt, steps_advanced: bool = False, counter: Optional[Tensor] = None, counter_inc: int = 0,
              hyper_dev: Optional[Tensor] = None) -> bool:
    """``optimizer.step()`` of a plain capturable ``torch.optim.Adam`` through ``segger_adam_step``: every parameter tensor
    in two launches, on the optimizer's own state tensors (checkpoints and eager ``optimizer.step()`` calls stay
    interchangeable).  -> False, nothing done, when the optimizer is anything else (amsgrad, weight decay, maximize, a
    tensor learning rate, non-fp32 or non-contiguous parameters, state not created yet): the caller then runs
    ``optimizer.step()`` itself.  ``steps_advanced``: the step counters were advanced already (:func:`adam_step_counters`);
    ``counter`` (int64[1] on the device): ``counter += counter_inc`` rides in the update launch.  ``hyper_dev`` (float64[4]
    on the device = lr, beta1, beta2, eps; one parameter group): the kernel reads them from there when it RUNS
    (``segger_adam_step_dev``) -- a captured step follows a learning-rate schedule without a new capture.
    (Host cost ~135 us per call for 60 tensors, nearly all of it the per-tensor attribute reads; a cached launch table
    that re-checked pointers and state identity per step measured the same -- tools/host_phases.py.)"""
    # torch's AMP contract for fused optimizers (``_step_supports_amp_scaling``): ``GradScaler.step`` skips its own unscale /
    # inf check and hands both to the optimizer as ``grad_scale`` / ``found_inf``.  The kernel reads neither: torch's fused
    # step does the scaled, skippable update.
    jobs = _adam_jobs(opt)
    if jobs is None:
        return False
    jobs = [(g, rows) for g, rows in jobs if rows]
    if (steps_advanced or counter is not None or hyper_dev is not None) and len(jobs) != 1:
        raise RuntimeError("adam_step: advanced counters / device hyper-parameters need exactly one parameter group with gradients")
    if hyper_dev is not None and (hyper_dev.dtype != torch.float64 or hyper_dev.numel() != 4 or not hyper_dev.is_contiguous()
                                  or not hyper_dev.is_cuda):
        raise ValueError("adam_step: hyper_dev must be a contiguous float64[4] tensor on the device")
    lib = _lib.load()
    for g, rows in jobs:
        arr = (_lib.AdamTensor * len(rows))()
        for a, (p, gr, m, v, st) in zip(arr, rows):
            a.param, a.grad, a.exp_avg, a.exp_avg_sq, a.step, a.numel = (p.data_ptr(), gr.data_ptr(), m.data_ptr(),
                                                                         v.data_ptr(), st.data_ptr(), p.numel())
        dev = rows[0][0].device
        b1, b2 = g["betas"]
        with _lib.on_device(dev):
            if hyper_dev is not None:
                rc = lib.segger_adam_step_dev(arr, len(rows), hyper_dev.data_ptr(), 1 if steps_advanced else 0,
                                              _lib.ptr(counter), int(counter_inc), _lib.stream_ptr(dev))
            else:
                rc = lib.segger_adam_step_ex(arr, len(rows), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                             1 if steps_advanced else 0, _lib.ptr(counter), int(counter_inc), _lib.stream_ptr(dev))
        _lib.check(rc, "segger_adam_step")
    return True


def float_bits(x: float) -> int:
    """The int whose low 32 bits are the fp32 pattern of ``x`` (a ``const`` fill of a float buffer)."""
    import struct
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


@torch.no_grad()
def stage(segments, device) -> None:
    """``segger_stage``: all of ``segments`` in one launch.  A segment is ``(dst, src, fill, a, b, c[, add])``: ``dst`` a
    contiguous tensor written in full; ``src`` a contiguous tensor (or None) copied to its front; the rest filled by
    ``fill`` in ("const", "tile", "div", "mod", "ramp") with integer parameters a, b, c (see include/segger_amd.h);
    ``add`` (integer segments only) is added to every copied / tile-replicated element."""
    lib = _lib.load()
    n = len(segments)
    arr = (_lib.StageSeg * n)()
    for i, seg in enumerate(segments):
        dst, src, fill, a, b, c = seg[:6]
        add = int(seg[6]) if len(seg) > 6 else 0
        if add and dst.is_floating_point():
            raise TypeError(f"stage: segment {i}: `add` is for integer segments")
        if not dst.is_contiguous() or (src is not None and not src.is_contiguous()):
            raise ValueError("stage: tensors must be contiguous")
        n_copy = 0 if src is None else int(src.numel())
        g = arr[i]
        g.dst, g.src = dst.data_ptr(), (src.data_ptr() if src is not None and src.numel() else None)
        g.n_copy, g.n_total = n_copy, int(dst.numel())
        g.a, g.b, g.c = int(a), int(b), int(c)
        g.dst_bytes, g.src_bytes = dst.element_size(), (src.element_size() if src is not None else dst.element_size())
        g.fill = _FILLS[fill]
        g.copy_add = add
        if n_copy > g.n_total:
            raise ValueError(f"stage: segment {i}: source longer than destination")
        if src is not None and src.is_floating_point() != dst.is_floating_point():
            raise TypeError(f"stage: segment {i}: no conversion between float and integer")
        if dst.is_floating_point() and src is not None and src.dtype != dst.dtype:
            raise TypeError(f"stage: segment {i}: float segments copy bit patterns, dtypes must agree")
    with _lib.on_device(device):
        rc = lib.segger_stage(arr, n, _lib.stream_ptr(device))
    _lib.check(rc, "segger_stage")


# --------------------------------------------------------------------------
# Positional embedder: per-graph min / max
# --------------------------------------------------------------------------
@torch.no_grad()
def segment_minmax(pos: Tensor, batch: Optional[Tensor], num_graphs: int, keep_empty: bool = False,
                   out: Optional[Tuple[Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """-> (mins[num_graphs, 2], maxs[num_graphs, 2]) fp32 of ``pos`` grouped by ``batch``.  ``keep_empty``: graphs
    without nodes keep (+inf, -inf) instead of the reference's (0, 0) -- for consumers that only look up the graphs of
    existing nodes (one launch less).  ``out``: buffers the caller has ALREADY filled with +inf / -inf (a captured step
    does that in its staging launch): no initialising launch either."""
    _lib.require_cuda(pos)
    lib = _lib.load()
    dev = pos.device
    pos = pos.to(torch.float32).contiguous()
    if pos.dim() != 2 or pos.shape[1] != 2:
        raise ValueError("segment_minmax: pos must be [n, 2]")
    if batch is not None:
        batch = batch.to(device=dev, dtype=torch.int64).contiguous()
        if batch.numel() != pos.shape[0]:
            raise ValueError("segment_minmax: batch / pos length mismatch")
    flags = 2 if keep_empty else 0
    if out is not None:
        mins, maxs = out
        for t in (mins, maxs):
            if t.dtype != torch.float32 or t.numel() < 2 * num_graphs or not t.is_contiguous() or t.device != dev:
                raise ValueError("segment_minmax: out must be two contiguous float32 [num_graphs, 2] tensors on pos's device")
        flags |= 1
    else:
        mins = torch.empty((num_graphs, 2), dtype=torch.float32, device=dev)
        maxs = torch.empty((num_graphs, 2), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = lib.segger_segment_minmax_ex(pos.data_ptr(), _lib.ptr(batch), int(pos.shape[0]), int(num_graphs),
                                          mins.data_ptr(), maxs.data_ptr(), flags, _lib.stream_ptr(dev))
    _lib.check(rc, "segger_segment_minmax_ex")
    return mins, maxs


# --------------------------------------------------------------------------
# Tall-skinny projection GEMM (MFMA) with autograd
# --------------------------------------------------------------------------
def linear_supported(k_in: int, m_out: int, dtype: torch.dtype) -> bool:
    return dtype in DTYPE_CODE and bool(_lib.load().segger_linear_supported(int(k_in), int(m_out), DTYPE_CODE[dtype]))


def linear_fwd_launch(x: Tensor, w: Tensor, bias: Optional[Tensor], out: Optional[Tensor] = None) -> Tensor:
    """y = x @ w.T + bias for a [n, K] activation (row stride allowed) and contiguous [M, K] weight."""
    _lib.require_cuda(x, w)
    lib = _lib.load()
    n, k = x.shape
    m = w.shape[0]
    if w.dtype != x.dtype or not w.is_contiguous() or w.shape[1] != k:
        raise ValueError("linear: weight must be contiguous [M, K] in the activation dtype")
    xp, ldx = _rows(x, k, "x")
    y = out if out is not None else torch.empty((n, m), dtype=x.dtype, device=x.device)
    yp, ldy = _rows(y, m, "y")
    b = _f32_vec(bias, m, "bias")
    with _lib.on_device(x.device):
        rc = lib.segger_linear_fwd(xp, ldx, w.data_ptr(), _lib.ptr(b), yp, ldy, n, k, m, DTYPE_CODE[x.dtype],
                                   _lib.stream_ptr(x.device))
    _lib.check(rc, "segger_linear_fwd")
    return y


def f32_split_planes(w: Tensor, transposed: bool = False) -> Tensor:
    """bf16 [3, M, K]: hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid) of an fp32 matrix (the weight operand of
    :func:`linear_f32_split_launch`); ``transposed``: the planes of w^T, [3, K, M].  One launch (``segger_f32_split_planes``)."""
    w = w.detach()
    if not w.is_cuda:                     # (host tensors: plain torch, for tests of the split's algebra)
        w = w.float().t().contiguous() if transposed else w.float()
        hi = w.bfloat16()
        r = w - hi.float()
        mid = r.bfloat16()
        return torch.stack((hi, mid, (r - mid.float()).bfloat16())).contiguous()
    if w.dtype != torch.float32 or w.dim() != 2 or not w.is_contiguous():
        w = w.float().contiguous()
    m, k = int(w.shape[0]), int(w.shape[1])
    out = torch.empty((3, k, m) if transposed else (3, m, k), dtype=torch.bfloat16, device=w.device)
    with _lib.on_device(w.device):
        rc = _lib.load().segger_f32_split_planes(w.data_ptr(), m, k, int(transposed), out.data_ptr(), _lib.stream_ptr(w.device))
    _lib.check(rc, "segger_f32_split_planes")
    return out


def linear_f32_gate_launch(x: Tensor, w: Tensor, gate: Tensor, kind: str) -> Tensor:
    """``(x @ w.T) * act'(gate)`` at fp32 storage, ``kind`` in ("gelu", "silu") (``segger_linear_fwd_f32_gate``): the data
    gradient through an activation in one kernel; on the bf16x3 split when :data:`F32_SPLIT` covers the shape."""
    _lib.require_cuda(x, w, gate)
    n, k = x.shape
    m = int(w.shape[0])
    if x.dtype != torch.float32 or gate.dtype != torch.float32 or tuple(gate.shape) != (n, m) or w.shape[1] != k:
        raise ValueError("linear_f32_gate: x fp32 [n, K], w [M, K], gate fp32 [n, M]")
    xp, ldx = _rows(x, k, "x")
    gp, ldg = _rows(gate, m, "gate")
    y = torch.empty((n, m), dtype=torch.float32, device=x.device)
    split = F32_SPLIT and ldx % 4 == 0 and linear_f32_split_supported(k, m)
    wq = f32_split_planes(w) if split else w.detach().float().contiguous()
    with _lib.on_device(x.device):
        rc = _lib.load().segger_linear_fwd_f32_gate(xp, ldx, wq.data_ptr(), int(split), gp, ldg, {"gelu": 1, "silu": 2}[kind],
                                                    y.data_ptr(), m, n, k, m, _lib.stream_ptr(x.device))
    _lib.check(rc, "segger_linear_fwd_f32_gate")
    return y


def linear_f32_act_launch(x: Tensor, w: Tensor, bias: Optional[Tensor], kind: str) -> Tuple[Tensor, Tensor]:
    """``(y, act(y))`` with ``y = x @ w.T + bias`` at fp32 storage from ONE kernel (``segger_linear_fwd_f32_act``; exact-fp32
    MFMA, K in 64 / 128 / 256), ``kind`` in ("gelu", "silu")."""
    _lib.require_cuda(x, w)
    n, k = x.shape
    m = int(w.shape[0])
    if x.dtype != torch.float32 or w.dtype != torch.float32 or w.shape[1] != k or not w.is_contiguous():
        raise ValueError("linear_f32_act: x fp32 [n, K], w contiguous fp32 [M, K]")
    xp, ldx = _rows(x, k, "x")
    y = torch.empty((n, m), dtype=torch.float32, device=x.device)
    ya = torch.empty((n, m), dtype=torch.float32, device=x.device)
    b = None if bias is None else bias.detach().float().contiguous()
    with _lib.on_device(x.device):
        rc = _lib.load().segger_linear_fwd_f32_act(xp, ldx, w.data_ptr(), _lib.ptr(b), y.data_ptr(), m, ya.data_ptr(), m,
                                                   {"gelu": 1, "silu": 2}[kind], n, k, m, _lib.stream_ptr(x.device))
    _lib.check(rc, "segger_linear_fwd_f32_act")
    return y, ya


def linear_f32_gate_supported(k_in: int, m_out: int) -> bool:
    return (m_out % 64 == 0 and k_in in (64, 128, 256)) or (F32_SPLIT and linear_f32_split_supported(k_in, m_out))


class _MlpSiluF32(torch.autograd.Function):
    """``linear(silu(linear(x, w0, b0)), w2, b2)`` at fp32 storage as ONE autograd node (the positional embedder's shared MLP,
    ist_encoder.py:43-49, on its un-fused route): the backward's SiLU derivative rides in the epilogue of the data-gradient
    GEMM (``segger_linear_fwd_f32_gate``) instead of torch's silu_backward pass.  ``x`` receives no gradient (the sinusoid
    features are constants)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w2, b2, gelu_out=False):
        w0d, w2d = w0.detach().contiguous(), w2.detach().contiguous()
        z1, h1 = linear_f32_act_launch(x, w0d, b0, "silu")           # pre-activation and SiLU from one kernel
        ctx.save_for_backward(x, z1, h1, w2)
        if gelu_out:                                                  # (y, gelu(y)): the second a constant for autograd
            y, gy_ = linear_f32_act_launch(h1, w2d, b2, "gelu")
            ctx.mark_non_differentiable(gy_)
            return y, gy_
        return linear_fwd_launch(h1, w2d, b2.detach())

    @staticmethod
    def backward(ctx, gy, _unused=None):
        x, z1, h1, w2 = ctx.saved_tensors
        gy = gy.contiguous()
        gw2, gb2 = linear_wgrad_launch(gy, h1)
        dz1 = linear_f32_gate_launch(gy, w2.detach().t().contiguous(), z1, "silu")
        gw0, gb0 = linear_wgrad_launch(dz1, x)
        return None, gw0, gb0, gw2, gb2, None


def mlp_silu_f32_supported(x: Tensor, w0: Tensor, w2: Tensor) -> bool:
    d_in, d_h, d_out = int(w0.shape[1]), int(w0.shape[0]), int(w2.shape[0])
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] > 0 and not x.requires_grad
            and linear_supported(d_in, d_h, torch.float32) and linear_supported(d_h, d_out, torch.float32)
            and linear_wgrad_supported(d_h, d_in, torch.float32) and linear_wgrad_supported(d_out, d_h, torch.float32)
            and linear_f32_gate_supported(d_out, d_h) and d_in in (64, 128, 256) and d_h in (64, 128, 256)
            and d_h % 64 == 0 and d_out % 64 == 0)


def mlp_silu_f32(x: Tensor, w0, b0, w2, b2, gelu_out: bool = False):
    """``linear(silu(linear(x)))``; ``gelu_out``: ``(y, gelu(y))`` with the GELU a constant for autograd (the gradient is
    expected for ``y``: the consumer applies gelu')."""
    return _MlpSiluF32.apply(x, w0, b0, w2, b2, gelu_out)


# fp32 storage: the positional embedder's first Linear as a degree-12 polynomial of the normalised coordinate
# (csrc/posenc_poly.hip) -- no [2n, 256] feature matrix (2 GB at C2: segger_posfreq wrote it, two exact-fp32 GEMMs read it)
# and no K = 256 GEMM, forward or weight gradient.  False: posfreq + _MlpSiluF32 (round 5's route).
POS_POLY_F32 = True


def pos_poly_mlp_f32_supported(pos: Tensor, w0: Tensor, w2: Tensor) -> bool:
    dim, fd = int(w0.shape[0]), int(w0.shape[1])
    return (POS_POLY_F32 and pos.is_cuda and pos.dim() == 2 and pos.shape[1] == 2 and pos.shape[0] > 0 and not pos.requires_grad
            and w0.dtype == torch.float32 and w2.dtype == torch.float32 and tuple(w2.shape) == (dim, dim)
            and bool(_lib.load().segger_posenc_poly_supported(fd, dim))
            and linear_supported(dim, dim, torch.float32) and linear_wgrad_supported(dim, dim, torch.float32)
            and linear_f32_gate_supported(dim, dim) and dim % 64 == 0)


class _PosPolyMlpF32(torch.autograd.Function):
    """``Linear -> SiLU -> Linear`` of ``Positional2dEmbedder`` on the sinusoid of the normalised coordinates at fp32 storage,
    from the POSITIONS (ist_encoder.py:57-79 in one autograd node): the first layer by ``segger_posenc_poly_fwd`` (13
    coefficients per channel, refreshed from W0 / b0 by ``segger_posenc_poly_coef``), its weight gradient by
    ``segger_posenc_poly_wgrad`` (13 moments per channel); the 64-wide second layer on the exact-fp32 kernels as before."""

    @staticmethod
    def forward(ctx, pos, batch, mins, maxs, eps, max_period, w0, b0, w2, b2, gelu_out=False):
        lib = _lib.load()
        dev = pos.device
        pos = pos.detach().to(torch.float32).contiguous()
        if batch is not None:
            batch = batch.to(device=dev, dtype=torch.int64).contiguous()
        n, dim, fd = int(pos.shape[0]), int(w0.shape[0]), int(w0.shape[1])
        w0d, w2d, b0d = w0.detach().contiguous(), w2.detach().contiguous(), b0.detach().float().contiguous()
        coef = torch.empty((dim, 16), dtype=torch.float32, device=dev)
        z1 = torch.empty((2 * n, dim), dtype=torch.float32, device=dev)
        h1 = torch.empty_like(z1)
        pn = torch.empty(2 * n, dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            st = _lib.stream_ptr(dev)
            rc = lib.segger_posenc_poly_coef(w0d.data_ptr(), b0d.data_ptr(), fd, dim, float(max_period), coef.data_ptr(), st)
            _lib.check(rc, "segger_posenc_poly_coef")
            rc = lib.segger_posenc_poly_fwd(pos.data_ptr(), _lib.ptr(batch), mins.data_ptr(), maxs.data_ptr(), n, float(eps),
                                            coef.data_ptr(), dim, z1.data_ptr(), h1.data_ptr(), pn.data_ptr(), st)
            _lib.check(rc, "segger_posenc_poly_fwd")
        ctx.save_for_backward(pn, z1, h1, w2)
        ctx.cfg = (fd, dim, float(max_period))
        if gelu_out:                                                  # (y, gelu(y)): the second a constant for autograd
            y, gy_ = linear_f32_act_launch(h1, w2d, b2, "gelu")
            ctx.mark_non_differentiable(gy_)
            return y, gy_
        return linear_fwd_launch(h1, w2d, b2.detach())

    @staticmethod
    def backward(ctx, gy, _unused=None):
        pn, z1, h1, w2 = ctx.saved_tensors
        fd, dim, max_period = ctx.cfg
        lib = _lib.load()
        dev = gy.device
        gy = gy.contiguous()
        gw2, gb2 = linear_wgrad_launch(gy, h1)
        dz1 = linear_f32_gate_launch(gy, w2.detach().t().contiguous(), z1, "silu")
        gw0 = torch.empty((dim, fd), dtype=torch.float32, device=dev)
        gb0 = torch.empty(dim, dtype=torch.float32, device=dev)
        rows = int(dz1.shape[0])
        ws_bytes = int(lib.segger_posenc_poly_wgrad_workspace_bytes(rows, dim))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = lib.segger_posenc_poly_wgrad(dz1.data_ptr(), dim, pn.data_ptr(), rows, fd, dim, max_period, gw0.data_ptr(),
                                              gb0.data_ptr(), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev))
        _lib.check(rc, "segger_posenc_poly_wgrad")
        _defer_keep(ws)
        return None, None, None, None, None, None, gw0, gb0, gw2, gb2, None


def pos_poly_mlp_f32(pos: Tensor, batch: Optional[Tensor], mins: Tensor, maxs: Tensor, w0, b0, w2, b2, *, eps: float = 1e-8,
                     max_period: float = 10000.0, gelu_out: bool = False):
    """[n, 2] positions -> the embedder's MLP output as coordinate rows [2n, dim] (``gelu_out``: ``(y, gelu(y))``, the GELU a
    constant for autograd as in :func:`mlp_silu_f32`)."""
    return _PosPolyMlpF32.apply(pos, batch, mins, maxs, eps, max_period, w0, b0, w2, b2, gelu_out)


def linear_f32_split_supported(k_in: int, m_out: int) -> bool:
    return bool(_lib.load().segger_linear_fwd_f32_split_supported(int(k_in), int(m_out)))


def linear_f32_split_launch(x: Tensor, w3: Tensor, bias: Optional[Tensor]) -> Tensor:
    """y = x @ w.T + bias for fp32 ``x`` [n, K] on the bf16 matrix pipe (``segger_linear_fwd_f32_split``: three-way bf16
    split of both operands, six partial products, fp32 accumulation); ``w3`` = :func:`f32_split_planes` of w."""
    _lib.require_cuda(x, w3)
    n, k = x.shape
    m = int(w3.shape[1])
    if x.dtype != torch.float32 or w3.dtype != torch.bfloat16 or tuple(w3.shape) != (3, m, k) or not w3.is_contiguous():
        raise ValueError("linear_f32_split: x fp32 [n, K], w3 contiguous bf16 [3, M, K]")
    xp, ldx = _rows(x, k, "x")
    y = torch.empty((n, m), dtype=torch.float32, device=x.device)
    b = _f32_vec(bias, m, "bias")
    with _lib.on_device(x.device):
        rc = _lib.load().segger_linear_fwd_f32_split(xp, ldx, w3.data_ptr(), _lib.ptr(b), y.data_ptr(), m, n, k, m,
                                                     _lib.stream_ptr(x.device))
    _lib.check(rc, "segger_linear_fwd_f32_split")
    return y


def colsum(x: Tensor) -> Tensor:
    """fp32 column sums of a [n, cols] matrix (row stride allowed): the bias gradient ``grad_out.sum(0)``."""
    _lib.require_cuda(x)
    lib = _lib.load()
    n, cols = x.shape
    if x.dtype not in DTYPE_CODE or cols % 8 != 0 or cols > 2048:
        return x.sum(0, dtype=torch.float32)
    xp, ld = _rows(x, cols, "x")
    out = torch.empty(cols, dtype=torch.float32, device=x.device)
    ws_bytes = lib.segger_colsum_workspace_bytes(n, cols)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        rc = lib.segger_colsum(xp, ld, n, cols, DTYPE_CODE[x.dtype], out.data_ptr(), ws.data_ptr(), ws_bytes,
                               _lib.stream_ptr(x.device))
    _lib.check(rc, "segger_colsum")
    return out


def segment_rowsum(x: Tensor, by_id: EdgeCSR) -> Tensor:
    """fp32 [n_ids, D]: sum of the rows of ``x`` grouped by id (``by_id`` = :func:`rows_by_id` of the row ids)."""
    _lib.require_cuda(x)
    lib = _lib.load()
    n, d = x.shape
    n_seg = by_id.n_rows
    xp, ld = _rows(x, d, "x")
    out = torch.empty((n_seg, d), dtype=torch.float32, device=x.device)
    ws_bytes = lib.segger_segment_rowsum_workspace_bytes(n, n_seg, d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        rc = lib.segger_segment_rowsum(xp, ld, n, d, DTYPE_CODE[x.dtype], by_id.indptr.data_ptr(),
                                       by_id.col.data_ptr() if n else None, n_seg, out.data_ptr(), ws.data_ptr(),
                                       ws_bytes, _lib.stream_ptr(x.device))
    _lib.check(rc, "segger_segment_rowsum")
    return out


def linear_wgrad_supported(m_out: int, k_in: int, dtype: torch.dtype) -> bool:
    return dtype in DTYPE_CODE and bool(_lib.load().segger_linear_wgrad_supported(int(m_out), int(k_in), DTYPE_CODE[dtype]))


def linear_wgrad_launch(gy: Tensor, x: Tensor, want_bias: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """(dW[M, K], db[M]) fp32 of ``y = x @ W.T + b`` from ``gy`` [n, M] and ``x`` [n, K] (row strides allowed):
    one pass over both matrices on the MFMA weight-gradient kernel (``segger_linear_wgrad``)."""
    _lib.require_cuda(gy, x)
    lib = _lib.load()
    n, m = gy.shape
    k = x.shape[1]
    if x.shape[0] != n or gy.dtype != x.dtype:
        raise ValueError("linear_wgrad: gy / x must share the row count and the dtype")
    gp, ldg = _rows(gy, m, "gy")
    xp, ldx = _rows(x, k, "x")
    gw = torch.empty((m, k), dtype=torch.float32, device=x.device)
    gb = torch.empty(m, dtype=torch.float32, device=x.device) if want_bias else None
    ws_bytes = _wgrad_ws_bytes(n, m, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    split = (F32_SPLIT and F32_SPLIT_WGRAD and x.dtype == torch.float32 and ldg % 4 == 0 and ldx % 4 == 0
             and bool(lib.segger_linear_wgrad_f32_split_supported(m, k)))
    with _lib.on_device(x.device):
        if split:        # fp32 storage: six bf16 partial products instead of the exact-fp32 MFMA (see F32_SPLIT)
            rc = lib.segger_linear_wgrad_f32_split(gp, ldg, xp, ldx, n, m, k, gw.data_ptr(), _lib.ptr(gb), ws.data_ptr(),
                                                   ws_bytes, _lib.stream_ptr(x.device))
        else:
            rc = lib.segger_linear_wgrad(gp, ldg, xp, ldx, n, m, k, DTYPE_CODE[x.dtype], gw.data_ptr(), _lib.ptr(gb),
                                         ws.data_ptr(), ws_bytes, _lib.stream_ptr(x.device))
    _lib.check(rc, "segger_linear_wgrad")
    _defer_keep(ws, gw, gb)
    return gw, gb


def linear_wgrad_dx_supported(m_out: int, k_in: int, dtype: torch.dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and bool(
        _lib.load().segger_linear_wgrad_dx_supported(int(m_out), int(k_in), DTYPE_CODE[dtype]))


def linear_wgrad_dx_gate_supported(m_out: int, k_in: int, dtype: torch.dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and bool(
        _lib.load().segger_linear_wgrad_dx_gate_supported(int(m_out), int(k_in), DTYPE_CODE[dtype]))


FUSED_WGRAD_DX = True        # tools flip it for A/B runs: False = separate data-gradient GEMM + weight-gradient kernel
FUSED_GELU_GATE = True       # tools flip it: False = the first layer's GELU derivative as an elementwise pass of its own


def linear_wgrad_dx_launch(gy: Tensor, x: Tensor, wt: Tensor, want_bias: bool = True, gate: Optional[Tensor] = None
                           ) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """(dX[n, K], dW[M, K], db[M]) of ``y = x @ W.T + b`` in ONE pass over ``gy`` [n, M] (``segger_linear_wgrad_dx``):
    ``wt`` = W^T [K, M] contiguous in the activation dtype; dX in the activation dtype, dW / db fp32.  ``gate`` [n, K]:
    dX comes multiplied by gelu'(gate) (x was gelu(gate); ``linear_wgrad_dx_gate_supported``)."""
    _lib.require_cuda(gy, x, wt)
    lib = _lib.load()
    n, m = gy.shape
    k = x.shape[1]
    if x.shape[0] != n or gy.dtype != x.dtype or wt.dtype != x.dtype or tuple(wt.shape) != (k, m) or not wt.is_contiguous():
        raise ValueError("linear_wgrad_dx: gy [n, M], x [n, K] and a contiguous W^T [K, M] of one dtype")
    gp, ldg = _rows(gy, m, "gy")
    xp, ldx = _rows(x, k, "x")
    qp, ldq = None, 0
    if gate is not None:
        if tuple(gate.shape) != (n, k) or gate.dtype != x.dtype:
            raise ValueError("linear_wgrad_dx: gate must be [n, K] in the activation dtype")
        _lib.require_cuda(gate)
        qp, ldq = _rows(gate, k, "gate")
    gx = torch.empty((n, k), dtype=x.dtype, device=x.device)
    gw = torch.empty((m, k), dtype=torch.float32, device=x.device)
    gb = torch.empty(m, dtype=torch.float32, device=x.device) if want_bias else None
    ws_bytes = _wgrad_ws_bytes(n, m, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        rc = lib.segger_linear_wgrad_dx(gp, ldg, xp, ldx, wt.data_ptr(), n, m, k, DTYPE_CODE[x.dtype], gw.data_ptr(),
                                        _lib.ptr(gb), gx.data_ptr(), k, qp, ldq, ws.data_ptr(), ws_bytes,
                                        _lib.stream_ptr(x.device))
    _lib.check(rc, "segger_linear_wgrad_dx")
    _defer_keep(ws, gw, gb)
    return gx, gw, gb


def linear_fwd_pair_launch(xa: Tensor, wa: Tensor, xb: Tensor, wb: Tensor) -> Tuple[Tensor, Tensor]:
    """``(xa @ wa.T, xb @ wb.T)`` for 16-bit activations whose widths may differ (``segger_linear_fwd_pair_k``: one launch
    for widths (384, 128) or equal widths -- a first layer's two data gradients -- two otherwise)."""
    _lib.require_cuda(xa, wa, xb, wb)
    args, outs = [], []
    for x, w in ((xa, wa), (xb, wb)):
        n, k = x.shape
        m = int(w.shape[0])
        if w.dtype != x.dtype or not w.is_contiguous() or w.shape[1] != k:
            raise ValueError("linear_fwd_pair: weights must be contiguous [M, K] in the activation dtype")
        y = torch.empty((n, m), dtype=x.dtype, device=x.device)
        a = _lib.LinearArgs()
        a.x, a.ldx = _rows(x, k, "x")
        a.w, a.y, a.ldy, a.n_rows, a.m_out = w.data_ptr(), y.data_ptr(), m, n, m
        args.append(a)
        outs.append(y)
    with _lib.on_device(xa.device):
        rc = _lib.load().segger_linear_fwd_pair_k(C.byref(args[0]), int(xa.shape[1]), C.byref(args[1]), int(xb.shape[1]),
                                                  DTYPE_CODE[xa.dtype], _lib.stream_ptr(xa.device))
    _lib.check(rc, "segger_linear_fwd_pair_k")
    return outs[0], outs[1]


def linear_wgrad_pair_launch(sides, dx: bool):
    """``sides`` = two (gy [n, M], x [n, K], wt [K, M] | None, want_bias) of one K and dtype -> [(gx | None, gw, gb | None)] * 2 from
    ONE launch (``segger_linear_wgrad_pair``; ``dx``: the one-pass form with the data gradients, else dW / db only)."""
    lib = _lib.load()
    dev, dt = sides[0][0].device, sides[0][1].dtype
    args, outs, keep = [], [], []
    for gy, x, wt, want_bias in sides:
        _lib.require_cuda(gy, x)
        n, m = gy.shape
        k = x.shape[1]
        if x.shape[0] != n or gy.dtype != dt or x.dtype != dt:
            raise ValueError("linear_wgrad_pair: gy / x must share the row count and the dtype")
        a = _lib.WgradArgs()
        a.dy, a.ld_dy = _rows(gy, m, "gy")
        a.x, a.ld_x = _rows(x, k, "x")
        a.n_rows, a.m_out = n, m
        gx = None
        if dx:
            if wt.dtype != dt or tuple(wt.shape) != (k, m) or not wt.is_contiguous():
                raise ValueError("linear_wgrad_pair: W^T must be contiguous [K, M] in the activation dtype")
            gx = torch.empty((n, k), dtype=dt, device=dev)
            a.w_t, a.dx, a.ld_dx = wt.data_ptr(), gx.data_ptr(), k
        gw = torch.empty((m, k), dtype=torch.float32, device=dev)
        gb = torch.empty(m, dtype=torch.float32, device=dev) if want_bias else None
        ws_bytes = _wgrad_ws_bytes(n, m, k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        a.grad_w, a.grad_b, a.workspace, a.workspace_bytes = gw.data_ptr(), _lib.ptr(gb), ws.data_ptr(), ws_bytes
        args.append(a)
        outs.append((gx, gw, gb))
        keep.append((ws, gw, gb))
    with _lib.on_device(dev):
        rc = lib.segger_linear_wgrad_pair(C.byref(args[0]), C.byref(args[1]), int(sides[0][1].shape[1]), DTYPE_CODE[dt],
                                          _lib.stream_ptr(dev))
    _lib.check(rc, "segger_linear_wgrad_pair")
    for kk in keep:
        _defer_keep(*kk)
    return outs


# ---- "the hand-written path is off": vendor-GEMM use is counted and announced once per (site, K, M, dtype) ------------------
# The MFMA projection kernels cover segger's CLI widths (hidden 64 / 128, heads x channels = 128, 16-bit and fp32).  Another
# width (say hidden_channels=96) still WORKS -- through torch's library GEMM -- but not on this project's kernels.
# ``ops.vendor_gemm_calls`` counts those calls per (site, K, M, dtype); the first one of each kind warns.
vendor_gemm_calls: dict = {}


def _vendor_gemm(site: str, k_in: int, m_out: int, dtype) -> None:
    key = (site, int(k_in), int(m_out), str(dtype).replace("torch.", ""))
    seen = vendor_gemm_calls.get(key, 0)
    vendor_gemm_calls[key] = seen + 1
    if not seen:
        import warnings
        warnings.warn(f"segger_amd: {site} for K={k_in} -> M={m_out} ({key[3]}) is not covered by the hand-written MFMA "
                      f"kernels (segger_linear_supported / segger_linear_wgrad_supported) and runs on the vendor GEMM; "
                      f"covered widths: include/segger_amd.h. Counted in segger_amd.ops.vendor_gemm_calls.",
                      RuntimeWarning, stacklevel=3)


def _weight_grad_gemm(gy: Tensor, x: Tensor) -> Tensor:
    """dW[M, K] = gy^T x as a plain library GEMM: shapes the MFMA weight-gradient kernel does not cover
    (``linear_wgrad_supported``).  fp32 result."""
    _vendor_gemm("weight gradient dW = dY^T X", x.shape[1], gy.shape[1], x.dtype)
    return (gy.t() @ x).float()


# Parameters change between forwards in ways their version counters do not always show: torch's FUSED optimizers
# update them in place without bumping ``_version`` (measured: fused Adam 0 -> 0, foreach Adam 0 -> 1).  Every
# optimizer step therefore advances a generation counter ON THE PARAMETERS IT STEPPED (torch's optimizer post-step
# hook); anything else that writes parameters behind autograd's back (``p.data`` arithmetic, a hipGraph replay) must
# call :func:`invalidate_weights` on them, or :func:`invalidate_weight_cache` (everything).  The generation is per
# parameter so that one model's optimizer step does not re-key another model's packs: a pending backward of the other
# model must not see "weights changed" (and its packs are not re-copied for nothing).
_GLOBAL_GENERATION = [0]
_GEN_ATTR = "_segger_weight_gen"
_BASE_ATTR = "_segger_weight_base"      # an alias (detached view) of a parameter names it here: it ages with its base


def alias_of(p: Tensor) -> Tensor:
    """A distinct autograd leaf over the storage of parameter ``p`` whose cached copies follow ``p``'s generation."""
    a = p.detach().requires_grad_(p.requires_grad)
    setattr(a, _BASE_ATTR, p)
    return a


def _gen_of(p) -> tuple:
    base = getattr(p, _BASE_ATTR, None)
    return (getattr(p, _GEN_ATTR, 0), 0 if base is None else getattr(base, _GEN_ATTR, 0))


def invalidate_weights(params) -> None:
    """The given parameters were written in place: their cached compute-dtype copies are rebuilt on next use."""
    for p in params:
        if p is not None:
            setattr(p, _GEN_ATTR, getattr(p, _GEN_ATTR, 0) + 1)


def invalidate_weight_cache(*_args, **_kwargs) -> None:
    """Forget every cached compute-dtype copy of the projection weights (they are rebuilt on next use)."""
    _GLOBAL_GENERATION[0] += 1


def _optimizer_stepped(optimizer, *_args, **_kwargs) -> None:
    for group in optimizer.param_groups:
        invalidate_weights(group["params"])


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_optimizer_stepped)

# While a hipGraph is being captured, a refresh must only touch buffers the captured step owns: a multi-tensor copy
# over every pack of the process would bake pointers to OTHER models' packs (and their source parameters) into the
# graph, and replays would write through them after those models are gone.  ``pack_scope(params)`` names the
# parameters of the step being captured; outside a scope a capturing refresh is limited to the requesting pack.
_PACK_SCOPE: list = []


class pack_scope:
    def __init__(self, params):
        self.ids = frozenset(id(p) for p in params)

    def __enter__(self):
        _PACK_SCOPE.append(self.ids)
        return self

    def __exit__(self, *exc):
        _PACK_SCOPE.pop()
        return False


class _Pack:
    """Compute-dtype copy of one or several row-stacked fp32 master weights (+ the fp32 stacked bias, + the transposed
    copy the data gradient streams), refreshed only when a parameter changed (optimizer step, load_state_dict, .to()):
    the three projections that read x_tx (lin_l / lin_r of tx-neighbors-tx, lin_l of tx-belongs-bd) are ONE GEMM
    without a per-forward cat + cast + transpose (and without autograd's slice-copies on the way back).  The copies
    live in persistent buffers; after an optimizer step ALL stale packs of the process are refreshed by one multi-tensor
    copy (``torch._foreach_copy_`` casts fp32 -> bf16 into row windows of the stacked buffers in a single launch)."""

    def __init__(self, weights, biases):
        import weakref
        self.params = [weakref.ref(p) for p in tuple(weights) + tuple(b for b in biases if b is not None)]
        self.param_ids = frozenset(id(p) for p in tuple(weights) + tuple(b for b in biases if b is not None))
        self.rows = [int(w.shape[0]) for w in weights]
        self.has_bias = [b is not None for b in biases]
        self.k = int(weights[0].shape[1])
        self.dtype = None
        self.w = self.b = self._wt = None
        self._wt_fresh = False
        self.key = None
        self.views: list = []

    def _alloc(self, dtype, device):
        m = sum(self.rows)
        self.w = torch.empty((m, self.k), dtype=dtype, device=device)
        self.b = torch.zeros(m, dtype=torch.float32, device=device) if any(self.has_bias) else None
        self._wt, self._wt_fresh, self.dtype = None, False, dtype
        self.views, r0 = [], 0
        for r in self.rows:
            self.views.append(self.w[r0:r0 + r]); r0 += r
        r0 = 0
        for r, hb in zip(self.rows, self.has_bias):
            if hb:
                self.views.append(self.b[r0:r0 + r])
            r0 += r

    def _current_key(self):
        ps = [r() for r in self.params]
        if any(p is None for p in ps):
            return None, ps
        return (_GLOBAL_GENERATION[0],) + tuple((p.data_ptr(), p._version) + _gen_of(p) for p in ps), ps

    def get(self, dtype, device):
        key, ps = self._current_key()
        if self.w is None or self.dtype != dtype or self.w.device != device:
            self._alloc(dtype, device)
            self.key = None
        if self.key != key:
            if self.key is not None:
                _refresh_stale_packs(self)                   # one launch for every stale pack (of the scope)
            if self.key != key:                              # (first use, or skipped by the scope)
                with torch.no_grad():
                    _copy_groups(self.views, [p.detach() for p in ps])
                self.key, self._wt_fresh = key, False
        return self

    def planes(self, transposed: bool = False) -> Tensor:
        """fp32 packs: the bf16 [3, M, K] split of the stacked weight (or [3, K, M] of its transpose) for the opt-in
        bf16x3 projections (F32_SPLIT), rebuilt when the pack was refreshed."""
        slot = "_planes_t" if transposed else "_planes"
        hit = self.__dict__.get(slot)
        if hit is None or hit[0] != self.key:
            with torch.no_grad():
                hit = (self.key, f32_split_planes(self.w, transposed=transposed))
            self.__dict__[slot] = hit
        return hit[1]

    @property
    def wt(self) -> Tensor:                              # [K, M]: dX = dY @ W
        if self._wt is None:
            self._wt = torch.empty((self.k, sum(self.rows)), dtype=self.dtype, device=self.w.device)
        if not self._wt_fresh:
            self._wt.copy_(self.w.t())
            self._wt_fresh = True
        return self._wt


_PACKS: dict = {}


def _copy_groups(dsts, srcs) -> None:
    """``torch._foreach_copy_`` per destination dtype: one multi-tensor launch casts all fp32 weights into their bf16 /
    f16 row windows, one copies the fp32 biases.  (A single call over destinations of mixed dtypes mis-copied the fp32
    -> fp32 part on torch 2.10 / ROCm: biases came out wrong while the weights were right.)"""
    by_dtype: dict = {}
    for d, s_ in zip(dsts, srcs):
        g = by_dtype.setdefault(d.dtype, ([], []))
        g[0].append(d); g[1].append(s_)
    for d_list, s_list in by_dtype.values():
        torch._foreach_copy_(d_list, s_list)


@torch.no_grad()
def _refresh_stale_packs(requester: "_Pack") -> None:
    """Refresh every pack whose parameters changed since it was filled: ONE launch for all 16-bit packs on the GPU
    (``segger_pack_refresh``: casts into the stacked buffers, their transposed copies, the bias copies); anything else
    by one multi-tensor copy per dtype + one launch for the transposed copies.  During a hipGraph capture only the
    packs of the active :class:`pack_scope` (or, without one, only ``requester``) are touched: see ``_PACK_SCOPE``."""
    scope = _PACK_SCOPE[-1] if _PACK_SCOPE else None
    capturing = requester.w.is_cuda and torch.cuda.is_current_stream_capturing()
    dsts, srcs, live, one_launch = [], [], [], []
    for pk in _PACKS.values():
        if pk.w is None or pk.w.device != requester.w.device:
            continue
        if scope is not None:
            if not pk.param_ids <= scope:
                continue
        elif capturing and pk is not requester:
            continue
        key, ps = pk._current_key()
        if key is None or key == pk.key:
            continue
        fast = (pk.w.is_cuda and pk.w.element_size() == 2
                and all(p.dtype == torch.float32 and p.is_contiguous() and p.device == pk.w.device for p in ps))
        if fast:
            one_launch.append((pk, ps))
        else:
            dsts += pk.views
            srcs += [p.detach() for p in ps]
        live.append((pk, key))
    if dsts:
        _copy_groups(dsts, srcs)
    # fp32 packs that serve the bf16x3 split kernels: the plane sets they already hold (normal and / or transposed) are
    # rebuilt for ALL of them by ONE launch (segger_f32_split_planes_many) into the same buffers -- lazily, one launch per
    # pack and orientation, they were 16-18 launches of 5 us in a captured 1M-edge step at fp32 storage
    jobs = []
    for pk, key in live:
        if pk.w.is_cuda and pk.w.dtype == torch.float32:
            for slot, tr in (("_planes", 0), ("_planes_t", 1)):
                hit = pk.__dict__.get(slot)
                if hit is not None and hit[1].is_cuda and hit[1].device == pk.w.device:
                    jobs.append((pk, slot, tr, key, hit[1]))
    for j0 in range(0, len(jobs), _lib.PLANES_MAX_JOBS):
        chunk = jobs[j0:j0 + _lib.PLANES_MAX_JOBS]
        arr = (_lib.PlanesJob * len(chunk))()
        for a, (pk, slot, tr, key, buf) in zip(arr, chunk):
            a.w, a.rows, a.cols, a.transpose, a.planes = pk.w.data_ptr(), int(pk.w.shape[0]), int(pk.w.shape[1]), tr, buf.data_ptr()
        dev = chunk[0][0].w.device
        with _lib.on_device(dev):
            rc = _lib.load().segger_f32_split_planes_many(arr, len(chunk), _lib.stream_ptr(dev))
        _lib.check(rc, "segger_f32_split_planes_many")
        for pk, slot, tr, key, buf in chunk:
            pk.__dict__[slot] = (key, buf)
    done_t = set()
    for dt in {pk.w.dtype for pk, _ in one_launch}:
        # casts into the stacked buffers, transposed copies and bias copies of every stale pack: ONE launch
        # (segger_pack_refresh) instead of two multi-tensor copies + the transposing launch below
        group = [(pk, ps) for pk, ps in one_launch if pk.w.dtype == dt]
        n_seg = sum(len(pk.views) for pk, _ in group)
        arr = (_lib.PackSeg * n_seg)()
        i = 0
        for pk, ps in group:
            m_total, r0 = sum(pk.rows), 0
            n_w = len(pk.rows)
            for j, r in enumerate(pk.rows):
                g = arr[i]; i += 1
                g.src, g.dst, g.rows, g.cols = ps[j].data_ptr(), pk.views[j].data_ptr(), r, pk.k
                if pk._wt is not None:
                    g.dst_t, g.ld_t = pk._wt.data_ptr() + r0 * pk._wt.element_size(), m_total
                r0 += r
            for j in range(n_w, len(pk.views)):              # biases, in the order of pk.views
                g = arr[i]; i += 1
                g.src, g.dst, g.rows, g.cols, g.dst_f32 = ps[j].data_ptr(), pk.views[j].data_ptr(), int(pk.views[j].numel()), 1, 1
            if pk._wt is not None:
                done_t.add(id(pk))
        dev = group[0][0].w.device
        with _lib.on_device(dev):
            rc = _lib.load().segger_pack_refresh(arr, n_seg, DTYPE_CODE[dt], _lib.stream_ptr(dev))
        _lib.check(rc, "segger_pack_refresh")
    for pk, key in live:
        pk.key, pk._wt_fresh = key, id(pk) in done_t
    # the transposed copies the data gradients stream (every pack that has been through a backward): one launch
    tr = [pk for pk, _ in live if pk._wt is not None and pk.w.element_size() == 2 and pk.w.is_cuda and not pk._wt_fresh]
    if tr:
        arr = (_lib.TransposeSeg * len(tr))()
        for i, pk in enumerate(tr):
            arr[i].dst, arr[i].src = pk._wt.data_ptr(), pk.w.data_ptr()
            arr[i].rows, arr[i].cols = int(pk.w.shape[0]), int(pk.w.shape[1])
        dev = tr[0].w.device
        with _lib.on_device(dev):
            rc = _lib.load().segger_transpose_many(arr, len(tr), _lib.stream_ptr(dev))
        _lib.check(rc, "segger_transpose_many")
        for pk in tr:
            pk._wt_fresh = True


def packs_of(params) -> list:
    """The live packs built over (a subset of) ``params``: a captured step keeps them -- and so their buffers -- alive
    for as long as its graph may replay."""
    ids = frozenset(id(p) for p in params)
    return [pk for pk in _PACKS.values() if pk.param_ids <= ids]


def _pack_for(weights, biases) -> _Pack:
    """The cache entry of a parameter group, keyed by the tensors' identities (dropped when the first one dies)."""
    import weakref
    ids = tuple(id(w) for w in weights) + tuple(id(b) for b in biases)
    pk = _PACKS.get(ids)
    if pk is None:
        pk = _PACKS[ids] = _Pack(weights, biases)
        weakref.finalize(weights[0], _PACKS.pop, ids, None)
    return pk


class _Linear(torch.autograd.Function):
    """x [n, K] (bf16/f16); ``n_w`` fp32 master weights [M_i, K] stacked by rows, ``n_w`` fp32 biases (or None)
    -> [n, sum M_i].  Forward, the data gradient and the weight / bias gradients run on the hand-written MFMA kernels
    (csrc/linear.hip, csrc/linear_wgrad.hip); each parameter's gradient is a row window of the one fused result."""

    @staticmethod
    def forward(ctx, x, n_w, *params):
        weights, biases = params[:n_w], params[n_w:]
        pk = _pack_for(weights, biases).get(x.dtype, x.device)
        if F32_SPLIT and x.dtype == torch.float32 and linear_f32_split_supported(x.shape[1], pk.w.shape[0]):
            y = linear_f32_split_launch(x, pk.planes(), pk.b)
        else:
            y = linear_fwd_launch(x, pk.w, pk.b)
        ctx.save_for_backward(x)
        ctx.n_w = n_w
        _linear_save(ctx, pk, weights, biases)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return _linear_backward(ctx, x, gy, ctx.needs_input_grad[0], ctx.needs_input_grad[2:])


def _linear_save(ctx, pk, weights, biases) -> None:
    ctx.pack = pk
    ctx.rows = [int(w.shape[0]) for w in weights]
    ctx.has_bias = [b is not None for b in biases]
    # the pack is refreshed IN PLACE after an optimizer step; a backward that runs later than that (not the case in
    # forward -> backward -> step training) would see the new weights: remember which generation this forward used
    ctx.w, ctx.wt_of, ctx.w_key = pk.w, pk, pk.key


def _grad_rows(gy: Tensor, dt) -> Tensor:
    if gy.dtype != dt:
        gy = gy.to(dt)
    if gy.dim() != 2 or (gy.shape[0] > 1 and gy.stride(1) != 1):
        gy = gy.contiguous()
    return gy


def _linear_backward(st, x, gy, need_x: bool, need_params, pre=None) -> tuple:
    """-> (gx, None, *grads_w, *grads_b) of one projection; ``st`` holds what :func:`_linear_save` left, ``need_params`` =
    needs_input_grad of its weights then biases; ``pre`` = (gx | None, gw, gb) already computed by a paired launch."""
    dt = x.dtype
    n_w = len(st.rows)
    gy = _grad_rows(gy, dt)
    w = st.w
    m, k = w.shape
    gx = None
    want_w = any(need_params[:n_w])
    want_b = any(h and g for h, g in zip(st.has_bias, need_params[n_w:]))
    gw = gb = None
    if pre is not None:
        gx, gw, gb = pre
    if need_x and gx is None:
        if st.wt_of.key != st.w_key:
            raise RuntimeError("the projection weights changed between this forward and its backward "
                               "(optimizer step in between?): run backward before stepping")
        # (st.wt_of.wt = W^T [K, M], dX = dY @ W -- asked for only where it is read: at fp32 storage the split kernels take
        #  the transposed PLANES instead, and the property's transposing copy was one stray launch per projection and step)
        if (FUSED_WGRAD_DX and (want_w or want_b) and x.shape[0] > 0 and linear_wgrad_dx_supported(m, k, dt)):
            gx, gw, gb = linear_wgrad_dx_launch(gy, x, st.wt_of.wt, want_bias=want_b)    # dY read ONCE for dX, dW and db
        elif F32_SPLIT and dt == torch.float32 and linear_f32_split_supported(m, k):
            gx = linear_f32_split_launch(gy, st.wt_of.planes(transposed=True), None)
        elif linear_supported(m, k, dt):
            gx = linear_fwd_launch(gy, st.wt_of.wt, None)
        else:
            _vendor_gemm("data gradient dX = dY W", m, k, dt)
            gx = gy @ w
    if gw is not None:
        pass
    elif (want_w or want_b) and x.shape[0] > 0 and linear_wgrad_supported(m, k, dt):
        gw, gb = linear_wgrad_launch(gy, x, want_bias=want_b)       # dY and X read once for both
    else:
        if want_w:
            gw = _weight_grad_gemm(gy, x)
        if want_b:
            gb = colsum(gy)
    grads_w, grads_b, r0 = [], [], 0
    for i, r in enumerate(st.rows):
        grads_w.append(gw[r0:r0 + r] if (gw is not None and need_params[i]) else None)
        grads_b.append(gb[r0:r0 + r] if (gb is not None and st.has_bias[i] and need_params[n_w + i]) else None)
        r0 += r
    return (gx, None) + tuple(grads_w) + tuple(grads_b)


class _State:
    pass


class _LinearPair(torch.autograd.Function):
    """Two projections with the same K as one launch (``segger_linear_fwd_pair``): (xa, xb, n_wa, n_wb, *weights_a,
    *biases_a, *weights_b, *biases_b) -> (ya, yb).  Backward: each side's own :func:`_linear_backward`."""

    @staticmethod
    def forward(ctx, xa, xb, n_wa, n_wb, *params):
        pa, pb = params[:2 * n_wa], params[2 * n_wa:]
        lib = _lib.load()
        sides, args, outs = [], [], []
        for x, n_w, pp in ((xa, n_wa, pa), (xb, n_wb, pb)):
            weights, biases = pp[:n_w], pp[n_w:]
            pk = _pack_for(weights, biases).get(x.dtype, x.device)
            st = _State()
            _linear_save(st, pk, weights, biases)
            sides.append(st)
            n, k = x.shape
            m = int(pk.w.shape[0])
            y = torch.empty((n, m), dtype=x.dtype, device=x.device)
            a = _lib.LinearArgs()
            a.x, a.ldx = _rows(x, k, "x")
            a.w, a.bias = pk.w.data_ptr(), _lib.ptr(_f32_vec(pk.b, m, "bias"))
            a.y, a.ldy = _rows(y, m, "y")
            a.n_rows, a.m_out = n, m
            args.append(a)
            outs.append(y)
        with _lib.on_device(xa.device):
            rc = lib.segger_linear_fwd_pair(C.byref(args[0]), C.byref(args[1]), int(xa.shape[1]), DTYPE_CODE[xa.dtype],
                                            _lib.stream_ptr(xa.device))
        _lib.check(rc, "segger_linear_fwd_pair")
        ctx.save_for_backward(xa, xb)
        ctx.sides, ctx.n_w = sides, (n_wa, n_wb)
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, gya, gyb):
        xa, xb = ctx.saved_tensors
        n_wa, n_wb = ctx.n_w
        need = ctx.needs_input_grad
        na, nb = need[4:4 + 2 * n_wa], need[4 + 2 * n_wa:]
        pre = (None, None)
        sts = ctx.sides
        dt = xa.dtype
        wants = [any(nn[:len(st.rows)]) or any(h and g for h, g in zip(st.has_bias, nn[len(st.rows):]))
                 for st, nn in zip(sts, (na, nb))]
        if (WGRAD_PAIR and all(wants) and gya is not None and gyb is not None and dt in (torch.bfloat16, torch.float16)
                and xa.shape[0] > 0 and xb.shape[0] > 0):
            # both sides' backward passes in one launch: with the data gradients when both want them and the one-pass
            # kernel covers both shapes, else the weight / bias gradients only
            k = int(xa.shape[1])
            ms = [int(st.w.shape[0]) for st in sts]
            dx = (FUSED_WGRAD_DX and need[0] and need[1] and all(linear_wgrad_dx_supported(m, k, dt) for m in ms))
            if dx or all(linear_wgrad_supported(m, k, dt) for m in ms):
                for st in sts:
                    if dx and st.wt_of.key != st.w_key:
                        raise RuntimeError("the projection weights changed between this forward and its backward "
                                           "(optimizer step in between?): run backward before stepping")
                gya, gyb = _grad_rows(gya, dt), _grad_rows(gyb, dt)
                bias = [any(h and g for h, g in zip(st.has_bias, nn[len(st.rows):])) for st, nn in zip(sts, (na, nb))]
                pre = linear_wgrad_pair_launch([(gya, xa, sts[0].wt_of.wt if dx else None, bias[0]),
                                                (gyb, xb, sts[1].wt_of.wt if dx else None, bias[1])], dx)
                if not dx and need[0] and need[1] and all(linear_supported(m, k, dt) for m in ms):
                    # the two data gradients the one-pass kernel does not cover (a first layer reads K = 256): one launch
                    for st in sts:
                        if st.wt_of.key != st.w_key:
                            raise RuntimeError("the projection weights changed between this forward and its backward "
                                               "(optimizer step in between?): run backward before stepping")
                    gxa, gxb = linear_fwd_pair_launch(gya, sts[0].wt_of.wt, gyb, sts[1].wt_of.wt)
                    pre = [(gxa,) + tuple(pre[0][1:]), (gxb,) + tuple(pre[1][1:])]
        ra = _linear_backward(sts[0], xa, gya, need[0], na, pre[0])
        rb = _linear_backward(sts[1], xb, gyb, need[1], nb, pre[1])
        return (ra[0], rb[0], None, None) + ra[2:] + rb[2:]


# fp32 storage: forward projections and their data gradients as three-way bf16 splits on the bf16 matrix pipe
# (segger_linear_fwd_f32_split: 0.77 vs 1.17 ms for 1M x 128 -> 384) instead of the exact-fp32 MFMA.  ON by default since
# round 5: measured against fp64 its error is within the exact kernel's own (3.2e-7 vs 3.5e-7 of sum |x||w|,
# profiles/r04_f32_split.txt -- fp32 accumulation dominates both) and every fp32 parity test holds with it.  It is not
# bit-identical to a chain of fp32 FMAs: SEGGER_AMD_F32_EXACT=1 (or ops.F32_SPLIT = False) selects the exact kernels.
F32_SPLIT = os.environ.get("SEGGER_AMD_F32_EXACT", "0") in ("", "0")
F32_GATE_EPILOGUE = True     # fp32 storage: gelu' / silu' of a data gradient in the GEMM's epilogue (segger_linear_fwd_f32_gate)
F32_SPLIT_WGRAD = True       # (with F32_SPLIT) the weight gradients on the split as well (segger_linear_wgrad_f32_split)
LINEAR_PAIR = True           # tools flip it: False = one launch per projection
WGRAD_PAIR = True            # ... and per projection backward


def linear_pair(xa: Tensor, wa, ba, xb: Tensor, wb, bb) -> Tuple[Tensor, Tensor]:
    """``(linear(xa, wa, ba), linear(xb, wb, bb))`` -- as ONE launch when both are 2-D activations of the same dtype and
    width on the MFMA kernels (a hetero layer's transcript and boundary projections, ``lin_last`` of both node types)."""
    tup = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,)
    wa, ba, wb, bb = tup(wa), tup(ba), tup(wb), tup(bb)
    ma, mb = sum(int(w.shape[0]) for w in wa), sum(int(w.shape[0]) for w in wb)
    ok = (LINEAR_PAIR and xa.dim() == 2 and xb.dim() == 2 and xa.is_cuda and xb.is_cuda and xa.dtype == xb.dtype
          and xa.dtype in (torch.bfloat16, torch.float16)
          and xa.shape[1] == xb.shape[1] and xa.shape[0] > 0 and xb.shape[0] > 0
          and len(ba) == len(wa) and len(bb) == len(wb)
          and linear_supported(xa.shape[1], ma, xa.dtype) and linear_supported(xb.shape[1], mb, xb.dtype))
    if not ok:
        return linear(xa, wa, ba), linear(xb, wb, bb)
    if xa.shape[0] > 1 and xa.stride(1) != 1:
        xa = xa.contiguous()
    if xb.shape[0] > 1 and xb.stride(1) != 1:
        xb = xb.contiguous()
    return _LinearPair.apply(xa, xb, len(wa), len(wb), *wa, *ba, *wb, *bb)


def linear(x: Tensor, weight, bias) -> Tensor:
    """``F.linear`` for node-feature matrices.  ``weight`` / ``bias`` may be sequences of tensors: the maps are
    stacked by rows into one GEMM (``[lin_l | lin_r | ...](x)``).  Activations with a covered (K, M) use the MFMA
    kernels -- bf16 / f16 on v_mfma_f32_32x32x16, fp32 (the reference's arithmetic width) on the exact-fp32
    v_mfma_f32_32x32x2_f32 -- and only uncovered shapes fall to the vendor GEMM."""
    weights = tuple(weight) if isinstance(weight, (list, tuple)) else (weight,)
    biases = tuple(bias) if isinstance(bias, (list, tuple)) else (bias,)
    if len(biases) != len(weights):
        raise ValueError("linear: one bias (or None) per weight")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    m_out = sum(int(w.shape[0]) for w in weights)
    if x2.is_cuda and linear_supported(x2.shape[1], m_out, x2.dtype) and x2.shape[0] > 0:
        if x2.shape[0] > 1 and x2.stride(1) != 1:
            x2 = x2.contiguous()
        y = _Linear.apply(x2, len(weights), *weights, *biases)
    else:
        _lib.require_cuda(x2)
        if x2.shape[0] > 0:
            _vendor_gemm("projection y = x W^T", x2.shape[1], m_out, x2.dtype)
        w = weights[0] if len(weights) == 1 else torch.cat(weights, 0)
        b = None
        if any(bb is not None for bb in biases):
            b = torch.cat([bb if bb is not None else ww.new_zeros(ww.shape[0]) for ww, bb in zip(weights, biases)], 0)
        y = torch.nn.functional.linear(x2, w.to(x2.dtype), None if b is None else b.to(x2.dtype))
    return y.reshape(*lead, m_out)


# --------------------------------------------------------------------------
# Encoder front end / tail (fused row-wise kernels)
# --------------------------------------------------------------------------
@torch.no_grad()
def posfreq(pos: Tensor, batch: Optional[Tensor], mins: Tensor, maxs: Tensor, freq_dim: int, dtype: torch.dtype,
            eps: float = 1e-8, max_period: float = 10000.0) -> Tensor:
    """[n, 2] positions -> [n, 2, freq_dim] sinusoid of the per-graph normalised coordinates."""
    _lib.require_cuda(pos)
    lib = _lib.load()
    dev = pos.device
    pos = pos.to(torch.float32).contiguous()
    n = int(pos.shape[0])
    if batch is not None:
        batch = batch.to(device=dev, dtype=torch.int64).contiguous()
    out = torch.empty((n, 2, freq_dim), dtype=dtype, device=dev)
    with _lib.on_device(dev):
        rc = lib.segger_posfreq(pos.data_ptr(), _lib.ptr(batch), mins.data_ptr(), maxs.data_ptr(), n, freq_dim,
                                eps, max_period, out.data_ptr(), DTYPE_CODE[dtype], _lib.stream_ptr(dev))
    _lib.check(rc, "segger_posfreq")
    return out


def posmlp_supported(freq_dim: int, dim: int, dtype: torch.dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and bool(
        _lib.load().segger_posmlp_supported(int(freq_dim), int(dim), DTYPE_CODE[dtype]))


FUSED_POSMLP_BWD = True      # tools flip it: False = the embedder's backward as three kernels (round 2)


class _PosMlp(torch.autograd.Function):
    """Positional2dEmbedder in one kernel (``segger_posmlp_fwd``): [n, 2] positions -> [n, 128].  With gradients the
    kernel also stores the first layer's pre-activation and the normalised coordinates (4 bytes per row), and the
    backward is assembled from the projection kernels: dW2 / db2 by ``segger_linear_wgrad``, dh1 by
    ``segger_linear_fwd``, dW0 / db0 by ``segger_posmlp_wgrad`` (the sinusoid features regenerated in the kernel)."""

    @staticmethod
    def forward(ctx, pos, batch, mins, maxs, eps, max_period, dtype, train, gelu, w0, b0, w2, b2):
        _lib.require_cuda(pos, w0)
        lib = _lib.load()
        dev = pos.device
        pos = pos.to(torch.float32).contiguous()
        n = int(pos.shape[0])
        if batch is not None:
            batch = batch.to(device=dev, dtype=torch.int64).contiguous()
        pk0 = _pack_for((w0,), (b0,)).get(dtype, dev)
        pk2 = _pack_for((w2,), (b2,)).get(dtype, dev)
        pe = torch.empty((n, 2 * w2.shape[0]), dtype=dtype, device=dev)
        z1 = torch.empty((2 * n, w0.shape[0]), dtype=dtype, device=dev) if train else None
        pn = torch.empty(2 * n, dtype=torch.float32, device=dev) if train else None
        pre = torch.empty_like(pe) if (train and gelu) else None
        h1 = torch.empty_like(z1) if (train and not FUSED_POSMLP_BWD) else None    # (the one-pass backward recomputes it)
        ctx.set_materialize_grads(False)
        with _lib.on_device(dev):
            rc = lib.segger_posmlp_fwd(pos.data_ptr(), _lib.ptr(batch), mins.data_ptr(), maxs.data_ptr(), n, float(eps),
                                       float(max_period), pk0.w.data_ptr(), pk0.b.data_ptr(), pk2.w.data_ptr(),
                                       pk2.b.data_ptr(), pe.data_ptr(), _lib.ptr(z1), _lib.ptr(pn), _lib.ptr(h1), _lib.ptr(pre),
                                       int(bool(gelu)),
                                       DTYPE_CODE[dtype],
                                       _lib.stream_ptr(dev))
        _lib.check(rc, "segger_posmlp_fwd")
        ctx.by_pre = bool(train and int(gelu) == 2)
        if train:
            ctx.save_for_backward(z1, pn, pre, h1)
            ctx.pk2, ctx.key2, ctx.max_period = pk2, pk2.key, float(max_period)
        if ctx.by_pre:
            # the consumer applies gelu' itself (in the epilogue of its data-gradient kernel) and returns d loss / d pre:
            # `pe` = gelu(pre) leaves as a constant, `pre` as the differentiable output
            ctx.mark_non_differentiable(pe)
            return pe, pre
        return pe

    @staticmethod
    def backward(ctx, gpe, gpre=None):
        z1, pn, pre, h1 = ctx.saved_tensors
        dt = z1.dtype
        d = z1.shape[1]
        if ctx.by_pre:
            gpe = gpre
        if gpe is None:                                      # (nothing downstream used the output)
            return (None,) * 13
        if pre is not None and not ctx.by_pre:               # the output was gelu(embedder output)
            gpe = torch.ops.aten.gelu_backward(gpe.to(dt), pre)
        g = gpe.to(dt).reshape(-1, d)
        if g.shape[0] > 1 and g.stride(1) != 1:
            g = g.contiguous()
        if ctx.pk2.key != ctx.key2:
            raise RuntimeError("the positional MLP's weights changed between this forward and its backward")
        need = ctx.needs_input_grad
        if h1 is None:                                       # one pass over g: all four parameter gradients
            lib = _lib.load()
            dev = z1.device
            gw0 = torch.empty((d, 4 * d), dtype=torch.float32, device=dev)
            gb0 = torch.empty(d, dtype=torch.float32, device=dev)
            gw2 = torch.empty((d, d), dtype=torch.float32, device=dev)
            gb2 = torch.empty(d, dtype=torch.float32, device=dev)
            n_rows = int(g.shape[0])
            ws_bytes = lib.segger_posmlp_bwd_workspace_bytes(n_rows)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            gp, ldg = _rows(g, d, "g")
            with _lib.on_device(dev):
                rc = lib.segger_posmlp_bwd(gp, ldg, z1.data_ptr(), pn.data_ptr(), ctx.pk2.wt.data_ptr(), n_rows,
                                           ctx.max_period, DTYPE_CODE[dt], gw0.data_ptr(), gb0.data_ptr(), gw2.data_ptr(),
                                           gb2.data_ptr(), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev))
            _lib.check(rc, "segger_posmlp_bwd")
            _defer_keep(ws, gw0, gb0, gw2, gb2)
            return (None, None, None, None, None, None, None, None, None, gw0 if need[9] else None,
                    gb0 if need[10] else None, gw2 if need[11] else None, gb2 if need[12] else None)
        gw2, gb2 = linear_wgrad_launch(g, h1)
        # dz1 = (g @ W2) * silu'(z1): the SiLU derivative is applied in the GEMM's epilogue
        lib0 = _lib.load()
        dz1 = torch.empty_like(z1)
        wt = ctx.pk2.wt
        gp, ldg = _rows(g, d, "g")
        with _lib.on_device(z1.device):
            rc = lib0.segger_linear_fwd_silu_grad(gp, ldg, wt.data_ptr(), z1.data_ptr(), d, dz1.data_ptr(), d,
                                                  int(g.shape[0]), d, d, DTYPE_CODE[dt], _lib.stream_ptr(z1.device))
        _lib.check(rc, "segger_linear_fwd_silu_grad")
        # dW0 = dz1^T F with the sinusoid features F regenerated from one float per row inside the kernel
        lib = _lib.load()
        dev = z1.device
        gw0 = torch.empty((d, 4 * d), dtype=torch.float32, device=dev)
        gb0 = torch.empty(d, dtype=torch.float32, device=dev)
        ws_bytes = lib.segger_linear_wgrad_workspace_bytes(int(dz1.shape[0]), d, 4 * d)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        dp, ldd = _rows(dz1, d, "dz1")
        with _lib.on_device(dev):
            rc = lib.segger_posmlp_wgrad(dp, ldd, pn.data_ptr(), int(dz1.shape[0]), ctx.max_period, DTYPE_CODE[dt],
                                         gw0.data_ptr(), gb0.data_ptr(), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev))
        _lib.check(rc, "segger_posmlp_wgrad")
        _defer_keep(ws, gw0, gb0)
        return (None, None, None, None, None, None, None, None, None, gw0 if need[9] else None,
                gb0 if need[10] else None, gw2 if need[11] else None, gb2 if need[12] else None)


def _posmlp_fwd_launch(pos, batch, mins, maxs, eps, max_period, dtype, train, gelu, pk0, pk2, d):
    """One ``segger_posmlp_fwd`` launch -> (pe, z1, pn, pre): what ``_PosMlp`` / ``_PosMlpPair`` keep of a row set (the
    one-pass backward recomputes h1)."""
    dev = pos.device
    pos = pos.to(torch.float32).contiguous()
    n = int(pos.shape[0])
    if batch is not None:
        batch = batch.to(device=dev, dtype=torch.int64).contiguous()
    pe = torch.empty((n, 2 * d), dtype=dtype, device=dev)
    z1 = torch.empty((2 * n, d), dtype=dtype, device=dev) if train else None
    pn = torch.empty(2 * n, dtype=torch.float32, device=dev) if train else None
    pre = torch.empty_like(pe) if (train and gelu) else None
    with _lib.on_device(dev):
        rc = _lib.load().segger_posmlp_fwd(pos.data_ptr(), _lib.ptr(batch), mins.data_ptr(), maxs.data_ptr(), n, float(eps),
                                           float(max_period), pk0.w.data_ptr(), pk0.b.data_ptr(), pk2.w.data_ptr(),
                                           pk2.b.data_ptr(), pe.data_ptr(), _lib.ptr(z1), _lib.ptr(pn), None, _lib.ptr(pre),
                                           int(gelu), DTYPE_CODE[dtype], _lib.stream_ptr(dev))
    _lib.check(rc, "segger_posmlp_fwd")
    return pe, z1, pn, pre


class _PosMlpPair(torch.autograd.Function):
    """``Positional2dEmbedder`` of TWO row sets as one autograd node: a = the transcripts (GELU applied in the kernel,
    ``(gelu(pre), pre)`` out as ``posmlp(return_pre=True)``), b = the boundaries (plain).  Two forward launches; the
    backward is ONE ``segger_posmlp_bwd_pair`` launch whose partial sums cover both sets, so the embedder's parameters get a
    single gradient each -- two ``_PosMlp`` nodes cost four accumulation launches behind autograd."""

    @staticmethod
    def forward(ctx, pos_a, batch_a, mins_a, maxs_a, pos_b, batch_b, mins_b, maxs_b, eps, max_period, dtype, w0, b0, w2, b2):
        _lib.require_cuda(pos_a, pos_b, w0)
        dev = pos_a.device
        d = int(w0.shape[0])
        pk0 = _pack_for((w0,), (b0,)).get(dtype, dev)
        pk2 = _pack_for((w2,), (b2,)).get(dtype, dev)
        ctx.set_materialize_grads(False)
        pe_a, z1_a, pn_a, pre_a = _posmlp_fwd_launch(pos_a, batch_a, mins_a, maxs_a, eps, max_period, dtype, True, 2, pk0, pk2, d)
        pe_b, z1_b, pn_b, _ = _posmlp_fwd_launch(pos_b, batch_b, mins_b, maxs_b, eps, max_period, dtype, True, 0, pk0, pk2, d)
        ctx.save_for_backward(z1_a, pn_a, z1_b, pn_b)
        ctx.pk2, ctx.key2, ctx.max_period = pk2, pk2.key, float(max_period)
        ctx.mark_non_differentiable(pe_a)                    # gelu(pre) leaves as a constant; `pre` carries the gradient
        return pe_a, pre_a, pe_b

    @staticmethod
    def backward(ctx, _gpe_a, gpre_a, gpe_b):
        z1_a, pn_a, z1_b, pn_b = ctx.saved_tensors
        if gpre_a is None and gpe_b is None:
            return (None,) * 15
        if ctx.pk2.key != ctx.key2:
            raise RuntimeError("the positional MLP's weights changed between this forward and its backward")
        dt, d, dev = z1_a.dtype, int(z1_a.shape[1]), z1_a.device
        lib = _lib.load()

        def rows(g):
            if g is None:
                return None, d, 0
            g = g.to(dt).reshape(-1, d)
            if g.shape[0] > 1 and g.stride(1) != 1:
                g = g.contiguous()
            gp, ldg = _rows(g, d, "g")
            return (g, gp), ldg, int(g.shape[0])
        ga, lda, na = rows(gpre_a)
        gb, ldb, nb = rows(gpe_b)
        gw0 = torch.empty((d, 4 * d), dtype=torch.float32, device=dev)
        gb0 = torch.empty(d, dtype=torch.float32, device=dev)
        gw2 = torch.empty((d, d), dtype=torch.float32, device=dev)
        gb2 = torch.empty(d, dtype=torch.float32, device=dev)
        ws_bytes = lib.segger_posmlp_bwd_pair_workspace_bytes(na, nb)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = lib.segger_posmlp_bwd_pair(ga[1] if ga else None, lda, z1_a.data_ptr(), pn_a.data_ptr(), na,
                                            gb[1] if gb else None, ldb, z1_b.data_ptr(), pn_b.data_ptr(), nb,
                                            ctx.pk2.wt.data_ptr(), ctx.max_period, DTYPE_CODE[dt], gw0.data_ptr(), gb0.data_ptr(),
                                            gw2.data_ptr(), gb2.data_ptr(), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev))
        _lib.check(rc, "segger_posmlp_bwd_pair")
        _defer_keep(ws, gw0, gb0, gw2, gb2, ga, gb)
        need = ctx.needs_input_grad
        return (None,) * 11 + (gw0 if need[11] else None, gb0 if need[12] else None, gw2 if need[13] else None,
                               gb2 if need[14] else None)


def posmlp_pair_supported(w0: Tensor, b0, w2: Tensor, b2, dtype: torch.dtype) -> bool:
    """The one-node route of :func:`posmlp_pair`: the fused 16-bit embedder with its one-pass backward, training."""
    return (FUSED_POSMLP_BWD and b0 is not None and b2 is not None and posmlp_supported(w0.shape[1], w0.shape[0], dtype)
            and tuple(w2.shape) == (w0.shape[0], w0.shape[0]) and torch.is_grad_enabled()
            and any(t.requires_grad for t in (w0, b0, w2, b2)))


def posmlp_pair(pos_a: Tensor, batch_a, mins_a: Tensor, maxs_a: Tensor, pos_b: Tensor, batch_b, mins_b: Tensor, maxs_b: Tensor,
                w0: Tensor, b0: Tensor, w2: Tensor, b2: Tensor, dtype: torch.dtype, eps: float = 1e-8, max_period: float = 10000.0):
    """``((gelu(pe_a), pe_a), pe_b)``: :func:`posmlp` of two row sets (``a`` as ``gelu=True, return_pre=True``, ``b`` plain)
    behind ONE autograd node (:class:`_PosMlpPair`); see :func:`posmlp_pair_supported`."""
    if not posmlp_pair_supported(w0, b0, w2, b2, dtype):
        raise ValueError("posmlp_pair: unsupported (see posmlp_pair_supported)")
    act_a, pre_a, pe_b = _PosMlpPair.apply(pos_a, batch_a, mins_a, maxs_a, pos_b, batch_b, mins_b, maxs_b, eps, max_period, dtype,
                                           w0, b0, w2, b2)
    return (act_a, pre_a), pe_b


def posmlp(pos: Tensor, batch: Optional[Tensor], mins: Tensor, maxs: Tensor, w0: Tensor, b0: Tensor, w2: Tensor,
           b2: Tensor, dtype: torch.dtype, eps: float = 1e-8, max_period: float = 10000.0, gelu: bool = False,
           return_pre: bool = False):
    """``Positional2dEmbedder.forward`` (reference ist_encoder.py:33-79) for bf16 / f16 activations; ``gelu``: the GELU
    that ISTEncoder applies to the concatenated input (ist_encoder.py:324-325) on top, in the same kernel.
    ``return_pre`` (with ``gelu``): ``(gelu(pre), pre)`` where the first is a constant for autograd and ``pre`` (the
    embedder's output, None when nothing needs a gradient) carries the gradient -- for a consumer that multiplies by
    gelu'(pre) itself (``embed_linear``)."""
    if not posmlp_supported(w0.shape[1], w0.shape[0], dtype) or tuple(w2.shape) != (w0.shape[0], w0.shape[0]):
        raise ValueError("posmlp: unsupported shapes (see segger_posmlp_supported)")
    if b0 is None or b2 is None:
        raise ValueError("posmlp: the embedder's Linear layers carry biases")
    train = torch.is_grad_enabled() and any(t.requires_grad for t in (w0, b0, w2, b2))   # else nothing is stored
    if return_pre and gelu:
        if train:
            return _PosMlp.apply(pos, batch, mins, maxs, eps, max_period, dtype, train, 2, w0, b0, w2, b2)
        return _PosMlp.apply(pos, batch, mins, maxs, eps, max_period, dtype, train, 1, w0, b0, w2, b2), None
    return _PosMlp.apply(pos, batch, mins, maxs, eps, max_period, dtype, train, int(bool(gelu)), w0, b0, w2, b2)


class EmbedInput:
    """The transcripts' first-layer input ``gelu(cat(table[ids], pe))`` (ist_encoder.py:312-325) kept as its parts:
    ``table`` fp32 [G, D] (the gene embedding), ``ids`` int32 [n], ``act_pe`` = gelu(positional embedding) [n, D] in the
    compute dtype, ``by_gene`` = rows grouped by id.  :func:`embed_linear` projects it without materialising the
    concatenation.  ``pre_pe`` (optional): the positional embedding before its GELU when ``act_pe`` is a constant for
    autograd (``posmlp(return_pre=True)``): the projection's backward then returns d / d pre_pe."""

    def __init__(self, table: Tensor, ids: Tensor, act_pe: Tensor, by_gene: Optional[EdgeCSR],
                 pre_pe: Optional[Tensor] = None):
        self.table, self.ids, self.act_pe, self.by_gene, self.pre_pe = table, ids, act_pe, by_gene, pre_pe
        self.shape = (int(act_pe.shape[0]), int(table.shape[1]) + int(act_pe.shape[1]))
        self.dtype, self.device = act_pe.dtype, act_pe.device


class _RowBiasLinear(torch.autograd.Function):
    """y = c @ Wc^T + T[ids]  (``segger_linear_fwd_rowbias``).  Backward: dc = dY Wc, dWc = dY^T c (MFMA kernels),
    dT = rows of dY summed by id (``segger_segment_rowsum`` over the rows-by-gene grouping).  ``pre`` (optional):
    c = gelu(pre) is a constant and the data gradient is returned for ``pre``, gelu'(pre) applied in the kernel."""

    @staticmethod
    def forward(ctx, c, wc, tab, ids, by_gene, pre=None):
        lib = _lib.load()
        dev, dt = c.device, c.dtype
        n, k = c.shape
        m = wc.shape[0]
        w16 = wc.detach().to(dt).contiguous()
        tab16 = tab.detach().to(dt).contiguous()             # the table is added in fp32 but travels in the compute dtype
        y = torch.empty((n, m), dtype=dt, device=dev)
        cp, ldc = _rows(c, k, "c")
        with _lib.on_device(dev):
            rc = lib.segger_linear_fwd_rowbias(cp, ldc, w16.data_ptr(), None, tab16.data_ptr(), m, ids.data_ptr(),
                                               y.data_ptr(), m, n, k, m, DTYPE_CODE[dt], _lib.stream_ptr(dev))
        _lib.check(rc, "segger_linear_fwd_rowbias")
        ctx.save_for_backward(c, w16, pre)
        ctx.by_gene, ctx.ids, ctx.n_ids = by_gene, ids, int(tab.shape[0])
        return y

    @staticmethod
    def backward(ctx, gy):
        c, w16, pre = ctx.saved_tensors
        dt = c.dtype
        gy = gy.to(dt)
        if gy.shape[0] > 1 and gy.stride(1) != 1:
            gy = gy.contiguous()
        gc = gw = gt = None
        m, k = w16.shape
        want_c = ctx.needs_input_grad[0] or (pre is not None and ctx.needs_input_grad[5])
        gated = False
        if (FUSED_WGRAD_DX and want_c and ctx.needs_input_grad[1] and c.shape[0] > 0
                and linear_wgrad_dx_supported(m, k, dt)):
            gated = pre is not None and FUSED_GELU_GATE and linear_wgrad_dx_gate_supported(m, k, dt)
            gc, gw, _ = linear_wgrad_dx_launch(gy, c, w16.t().contiguous(), want_bias=False,     # dY read once for both
                                               gate=pre if gated else None)
        else:
            if want_c:
                gc = linear_fwd_launch(gy, w16.t().contiguous(), None)          # [n, M] @ Wc -> [n, K]
            if ctx.needs_input_grad[1]:
                gw, _ = linear_wgrad_launch(gy, c, want_bias=False)
        if ctx.needs_input_grad[2]:
            by_gene = ctx.by_gene if ctx.by_gene is not None else rows_by_id(ctx.ids, ctx.n_ids)
            gt = segment_rowsum(gy, by_gene)
        if pre is not None:
            if gc is not None and not gated:
                gc = torch.ops.aten.gelu_backward(gc, pre)
            return None, gw, gt, None, None, gc
        return gc, gw, gt, None, None, None


def embed_linear_supported(x: "EmbedInput", m_out: int) -> bool:
    k = int(x.act_pe.shape[1])
    return (x.act_pe.is_cuda and x.table.dtype == torch.float32 and m_out % 4 == 0 and k in (64, 128)
            and linear_supported(k, m_out, x.dtype)
            and linear_supported(m_out, k, x.dtype) and linear_wgrad_supported(m_out, k, x.dtype))


EMBED_LINEAR_ONE_NODE = True     # tools flip it: False = the table and its gradients through torch ops around _RowBiasLinear
EMBED_LINEAR_MAX_GENES = 1024


def _gene_table_args(table, weights, biases, dt):
    a = _lib.GeneTableArgs()
    d = int(table.shape[1])
    a.table, a.n_genes, a.D, a.n_w, a.dtype = table.data_ptr(), int(table.shape[0]), d, len(weights), DTYPE_CODE[dt]
    for i, (w, b) in enumerate(zip(weights, biases)):
        if w.dtype != torch.float32 or w.dim() != 2 or w.shape[1] != 2 * d or w.stride(1) != 1:
            raise ValueError("embed_linear: weights must be fp32 [m, 2 D] with a dense last dimension")
        a.w[i], a.ld_w[i], a.m[i] = w.data_ptr(), int(w.stride(0)) if w.shape[0] > 1 else 2 * d, int(w.shape[0])
        if b is not None:
            if b.dtype != torch.float32 or not b.is_contiguous() or b.numel() != w.shape[0]:
                raise ValueError("embed_linear: biases must be contiguous fp32 [m]")
            a.b[i] = b.data_ptr()
    return a


class _EmbedLinear(torch.autograd.Function):
    """``linear(gelu(cat(table[ids], pe)), cat(weights), cat(biases))`` as ONE autograd node on hand-written kernels only:
    forward = ``segger_gene_table_fwd`` (the per-gene table T = gelu(E) Wa^T + b, and the positional half Wc / Wc^T of the
    weights in the compute dtype) + ``segger_linear_fwd_rowbias``; backward = the one-pass MFMA kernel (dc, dWc from one
    read of dY) + the by-gene row sum of dY + ``segger_gene_table_bwd`` (dE, both halves of every weight's gradient, the
    bias gradients).  Arguments: c = gelu(pre) [n, D] (compute dtype), pre (or None), table, ids, by_gene, n_w, then the n_w
    weights and the n_w biases (None allowed)."""

    @staticmethod
    def forward(ctx, c, pre, table, ids, by_gene, n_w, *wb):
        weights, biases = wb[:n_w], wb[n_w:]
        lib = _lib.load()
        dev, dt = c.device, c.dtype
        n, d = c.shape
        g = int(table.shape[0])
        m = sum(int(w.shape[0]) for w in weights)
        a = _gene_table_args(table, weights, biases, dt)
        tab = torch.empty((g, m), dtype=dt, device=dev)
        wc = torch.empty((m, d), dtype=dt, device=dev)
        wc_t = torch.empty((d, m), dtype=dt, device=dev)
        a.tab, a.ld_tab, a.wc, a.wc_t = tab.data_ptr(), m, wc.data_ptr(), wc_t.data_ptr()
        y = torch.empty((n, m), dtype=dt, device=dev)
        cp, ldc = _rows(c, d, "c")
        split = (F32_SPLIT and dt == torch.float32 and ldc % 4 == 0 and linear_f32_split_supported(d, m))
        with _lib.on_device(dev):
            _lib.check(lib.segger_gene_table_fwd(C.byref(a), _lib.stream_ptr(dev)), "segger_gene_table_fwd")
            if split:      # fp32 storage: the positional GEMM on the bf16x3 split, the table row added in its epilogue
                w3 = f32_split_planes(wc)
                rc = lib.segger_linear_fwd_f32_split_rowbias(cp, ldc, w3.data_ptr(), tab.data_ptr(), m, ids.data_ptr(),
                                                             y.data_ptr(), m, n, d, m, _lib.stream_ptr(dev))
            else:
                rc = lib.segger_linear_fwd_rowbias(cp, ldc, wc.data_ptr(), None, tab.data_ptr(), m, ids.data_ptr(),
                                                   y.data_ptr(), m, n, d, m, DTYPE_CODE[dt], _lib.stream_ptr(dev))
        _lib.check(rc, "segger_linear_fwd_rowbias")
        ctx.save_for_backward(c, pre, table, ids, wc, wc_t, *weights)
        ctx.by_gene, ctx.n_w, ctx.has_bias = by_gene, n_w, tuple(b is not None for b in biases)
        return y

    @staticmethod
    def backward(ctx, gy):
        c, pre, table, ids, wc, wc_t = ctx.saved_tensors[:6]
        weights = ctx.saved_tensors[6:]
        n_w = ctx.n_w
        lib = _lib.load()
        dev, dt = c.device, c.dtype
        d = int(c.shape[1])
        m = int(wc.shape[0])
        gy = gy.to(dt)
        if gy.shape[0] > 1 and gy.stride(1) != 1:
            gy = gy.contiguous()
        need = ctx.needs_input_grad
        want_c = need[0] or (pre is not None and need[1])
        want_w = any(need[6:6 + n_w])
        gc = gw = None
        gated = False
        if FUSED_WGRAD_DX and want_c and want_w and c.shape[0] > 0 and linear_wgrad_dx_supported(m, d, dt):
            gated = pre is not None and FUSED_GELU_GATE and linear_wgrad_dx_gate_supported(m, d, dt)
            gc, gw, _ = linear_wgrad_dx_launch(gy, c, wc_t, want_bias=False, gate=pre if gated else None)
        else:
            if want_c:
                if (dt == torch.float32 and pre is not None and need[1] and gy.stride(0) % 4 == 0 and F32_GATE_EPILOGUE
                        and linear_f32_gate_supported(m, d)):
                    gc = linear_f32_gate_launch(gy, wc_t, pre, "gelu")      # dX * gelu'(pre) in one kernel
                    gated = True
                elif F32_SPLIT and dt == torch.float32 and gy.stride(0) % 4 == 0 and linear_f32_split_supported(m, d):
                    gc = linear_f32_split_launch(gy, f32_split_planes(wc, transposed=True), None)
                else:
                    gc = linear_fwd_launch(gy, wc_t, None)
            if want_w:
                gw, _ = linear_wgrad_launch(gy, c, want_bias=False)
        by_gene = ctx.by_gene if ctx.by_gene is not None else rows_by_id(ids, int(table.shape[0]))
        gt = segment_rowsum(gy, by_gene)
        if gw is not None and lib.segger_reductions_pending() >= 0:
            # inside ops.deferred_reductions the weight gradient above is a placeholder until the flush: run what is queued
            # now (one launch) and go on deferring -- the launch below READS dWc
            with _lib.on_device(dev):
                _lib.check(lib.segger_reductions_flush(_lib.stream_ptr(dev)), "segger_reductions_flush")
                _lib.check(lib.segger_reductions_defer_begin(), "segger_reductions_defer_begin")
        biases = [None] * n_w
        a = _gene_table_args(table, weights, biases, dt)
        a.g_tab, a.g_wc = gt.data_ptr(), _lib.ptr(gw)
        g_table = torch.empty_like(table) if need[2] else None
        a.g_table = _lib.ptr(g_table)
        g_w, g_b = [], []
        for i, w in enumerate(weights):
            gwi = torch.empty((int(w.shape[0]), 2 * d), dtype=torch.float32, device=dev) if need[6 + i] else None
            gbi = (torch.empty(int(w.shape[0]), dtype=torch.float32, device=dev)
                   if ctx.has_bias[i] and need[6 + n_w + i] else None)
            a.g_w[i], a.g_b[i] = _lib.ptr(gwi), _lib.ptr(gbi)
            g_w.append(gwi); g_b.append(gbi)
        with _lib.on_device(dev):
            _lib.check(lib.segger_gene_table_bwd(C.byref(a), _lib.stream_ptr(dev)), "segger_gene_table_bwd")
        if pre is not None:
            if gc is not None and not gated:
                gc = torch.ops.aten.gelu_backward(gc, pre)
            return (None, gc, g_table, None, None, None, *g_w, *g_b)
        return (gc, None, g_table, None, None, None, *g_w, *g_b)


def embed_linear(x: "EmbedInput", weight, bias) -> Tensor:
    """``linear(gelu(cat(table[ids], pe)), W, b)`` for the stacked first-layer projections without the concatenated
    [n, 2D] input: the embedding half depends on a row only through its gene, so it is a per-gene table
    ``T = gelu(table) Wa^T + b`` ([G, M], a tiny GEMM) added in the epilogue of the GEMM over the positional half
    (K: 2D -> D).  Autograd: T's gradient is the by-gene row sum of dY; table, Wa and b receive theirs through T."""
    weights = tuple(weight) if isinstance(weight, (list, tuple)) else (weight,)
    biases = tuple(bias) if isinstance(bias, (list, tuple)) else (bias,)
    d = int(x.table.shape[1])
    # (the table kernels are latency-sized: a few hundred genes.  A 5k-gene panel's dW reduction would take 0.6 ms on their
    #  24 workgroups -- the torch-composed route with its vendor GEMMs below serves those)
    if (EMBED_LINEAR_ONE_NODE and len(weights) <= 4 and x.act_pe.shape[1] == d and x.table.shape[0] <= EMBED_LINEAR_MAX_GENES
            and all(w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] == 2 * d and w.stride(1) == 1 for w in weights)
            and all(b is None or (b.dtype == torch.float32 and b.is_contiguous()) for b in biases)):
        return _EmbedLinear.apply(x.act_pe, x.pre_pe, x.table, x.ids, x.by_gene, len(weights), *weights, *biases)
    w = weights[0] if len(weights) == 1 else torch.cat(weights, 0)                # [M, 2D] fp32 master weights
    _vendor_gemm("per-gene table T = gelu(E) Wa^T (torch-composed first layer)", d, w.shape[0], torch.float32)
    tab = torch.nn.functional.gelu(x.table) @ w[:, :d].t()                         # [G, M]
    if any(b is not None for b in biases):
        tab = tab + torch.cat([b if b is not None else ww.new_zeros(ww.shape[0]) for ww, b in zip(weights, biases)], 0)
    return _RowBiasLinear.apply(x.act_pe, w[:, d:], tab, x.ids, x.by_gene, x.pre_pe)


class _EmbedGelu(torch.autograd.Function):
    """gelu(cat(table[ids], pe)): table fp32 [G, D] (embedding weight), ids int32 [n], pe [n, D] -> [n, 2D]."""

    @staticmethod
    def forward(ctx, table, ids, pe, by_gene):
        _lib.require_cuda(table, ids, pe)
        lib = _lib.load()
        n, d = pe.shape
        g = table.shape[0]
        out = torch.empty((n, 2 * d), dtype=pe.dtype, device=pe.device)
        pp, ldp = _rows(pe, d, "pe")
        with _lib.on_device(pe.device):
            rc = lib.segger_embed_gelu_fwd(table.data_ptr(), ids.data_ptr(), pp, ldp, n, g, d, out.data_ptr(), 2 * d,
                                           DTYPE_CODE[pe.dtype], _lib.stream_ptr(pe.device))
        _lib.check(rc, "segger_embed_gelu_fwd")
        ctx.save_for_backward(table, ids, pe)
        ctx.by_gene = by_gene
        return out

    @staticmethod
    def backward(ctx, gx0):
        table, ids, pe = ctx.saved_tensors
        lib = _lib.load()
        dev = pe.device
        n, d = pe.shape
        g = table.shape[0]
        if gx0.dtype != pe.dtype:
            gx0 = gx0.to(pe.dtype)
        if gx0.stride(-1) != 1:
            gx0 = gx0.contiguous()
        gp, ldg = _rows(gx0, 2 * d, "gx0")
        pp, ldp = _rows(pe, d, "pe")
        gpe = torch.empty_like(pe)
        want_table = ctx.needs_input_grad[0]
        gtable = torch.empty_like(table) if want_table else None
        ws_bytes = lib.segger_embed_gelu_bwd_workspace_bytes(n, g, d) if want_table else 0
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        by_gene = ctx.by_gene
        if want_table and by_gene is None:
            by_gene = rows_by_id(ids, g)
        with _lib.on_device(dev):
            rc = lib.segger_embed_gelu_bwd(gp, ldg, table.data_ptr(), pp, ldp, n, g, d, gpe.data_ptr(), d, _lib.ptr(gtable),
                                           by_gene.indptr.data_ptr() if want_table else None,
                                           (by_gene.col.data_ptr() if n else None) if want_table else None,
                                           ws.data_ptr(), ws_bytes, DTYPE_CODE[pe.dtype], _lib.stream_ptr(dev))
        _lib.check(rc, "segger_embed_gelu_bwd")
        return gtable, None, gpe, None


class _FrontJoin(torch.autograd.Function):
    """(gelu(cat(table[ids], pe[:n_tx])), gelu(cat(xb, pe[n_tx:]))): the encoder's layer-0 input of both node types in one
    launch each way (``segger_front_join_fwd`` / ``_bwd``); ``pe`` = the positional embeddings of the two types back to
    back.  The backward returns the gradient of ``pe`` as one matrix."""

    @staticmethod
    def forward(ctx, table, ids, xb, pe, by_gene):
        _lib.require_cuda(table, ids, xb, pe)
        lib = _lib.load()
        n_tx, n_bd, d = int(ids.shape[0]), int(xb.shape[0]), int(pe.shape[1])
        dev, dt = pe.device, pe.dtype
        out_tx = torch.empty((n_tx, 2 * d), dtype=dt, device=dev)
        out_bd = torch.empty((n_bd, 2 * d), dtype=dt, device=dev)
        a = _lib.FrontJoinArgs()
        a.table, a.ids, a.n_rows_table, a.D, a.dtype = table.data_ptr(), _lib.ptr(ids) if n_tx else None, int(table.shape[0]), d, DTYPE_CODE[dt]
        a.pe, a.ld_pe = _rows(pe, d, "pe")
        a.n_tx, a.n_bd = n_tx, n_bd
        a.xb, a.ld_xb = _rows(xb, d, "xb")
        a.out_tx, a.ld_out_tx, a.out_bd, a.ld_out_bd = out_tx.data_ptr(), 2 * d, out_bd.data_ptr(), 2 * d
        with _lib.on_device(dev):
            rc = lib.segger_front_join_fwd(C.byref(a), _lib.stream_ptr(dev))
        _lib.check(rc, "segger_front_join_fwd")
        ctx.save_for_backward(table, ids, xb, pe)
        ctx.by_gene = by_gene
        return out_tx, out_bd

    @staticmethod
    def backward(ctx, g_tx, g_bd):
        table, ids, xb, pe = ctx.saved_tensors
        lib = _lib.load()
        n_tx, n_bd, d = int(ids.shape[0]), int(xb.shape[0]), int(pe.shape[1])
        dev, dt = pe.device, pe.dtype

        def rows2(g, n):
            if g is None:
                g = torch.zeros((n, 2 * d), dtype=dt, device=dev)
            if g.dtype != dt:
                g = g.to(dt)
            if g.stride(-1) != 1:
                g = g.contiguous()
            return g
        g_tx, g_bd = rows2(g_tx, n_tx), rows2(g_bd, n_bd)
        g_pe = torch.empty((n_tx + n_bd, d), dtype=dt, device=dev)
        g_xb = torch.empty((n_bd, d), dtype=dt, device=dev)
        want_table = ctx.needs_input_grad[0]
        g_table = torch.empty_like(table) if want_table else None
        g = int(table.shape[0])
        ws_bytes = lib.segger_embed_gelu_bwd_workspace_bytes(n_tx, g, d) if want_table else 0
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        by_gene = ctx.by_gene
        if want_table and by_gene is None and n_tx:
            by_gene = rows_by_id(ids, g)
        a = _lib.FrontJoinArgs()
        a.table, a.ids, a.n_rows_table, a.D, a.dtype = table.data_ptr(), _lib.ptr(ids) if n_tx else None, g, d, DTYPE_CODE[dt]
        a.pe, a.ld_pe = _rows(pe, d, "pe")
        a.n_tx, a.n_bd = n_tx, n_bd
        a.xb, a.ld_xb = _rows(xb, d, "xb")
        a.g_tx, a.ld_g_tx = _rows(g_tx, 2 * d, "g_tx")
        a.g_bd, a.ld_g_bd = _rows(g_bd, 2 * d, "g_bd")
        a.g_pe, a.ld_g_pe, a.g_xb, a.ld_g_xb = g_pe.data_ptr(), d, g_xb.data_ptr(), d
        if want_table:
            a.g_table = g_table.data_ptr()
            if n_tx:
                a.gene_ptr, a.gene_rows = by_gene.indptr.data_ptr(), by_gene.col.data_ptr()
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws_bytes
        with _lib.on_device(dev):
            rc = lib.segger_front_join_bwd(C.byref(a), _lib.stream_ptr(dev))
        _lib.check(rc, "segger_front_join_bwd")
        return g_table, None, g_xb, g_pe, None


def front_join(table: Tensor, ids: Tensor, xb: Tensor, pe: Tensor, by_gene: Optional[EdgeCSR] = None):
    """-> (x_tx [n_tx, 2D], x_bd [n_bd, 2D]) = (gelu(cat(table[ids], pe[:n_tx])), gelu(cat(xb, pe[n_tx:]))), reference
    ist_encoder.py:312-320 for both node types.  ``pe``: [n_tx + n_bd, D]; ``by_gene`` as :func:`embed_gelu`."""
    if table.dtype != torch.float32 or not table.is_contiguous():
        raise TypeError("front_join: the embedding table must be contiguous fp32")
    d = int(table.shape[1])
    if pe.shape[1] != d or xb.shape[1] != d or d % 8 or pe.shape[0] != ids.shape[0] + xb.shape[0] or xb.dtype != pe.dtype:
        raise ValueError("front_join: pe [n_tx + n_bd, D] and xb [n_bd, D] must match the table width and each other's dtype")
    return _FrontJoin.apply(table, ids.to(torch.int32).contiguous(), xb, pe, by_gene)


def rows_by_id(ids: Tensor, n_ids: int) -> EdgeCSR:
    """Rows grouped by id (``indptr`` over ids, ``col`` = row numbers, ascending inside an id): what the
    embedding-table gradient sums over.  One radix sort; cache it per batch (``ISTEncoder`` does)."""
    from .graph import csr_from_coo
    n = int(ids.shape[0])
    return csr_from_coo(ids.long(), torch.arange(n, device=ids.device), int(n_ids), max(n, 1), validate=False)


def embed_gelu(table: Tensor, ids: Tensor, pe: Tensor, by_gene: Optional[EdgeCSR] = None) -> Tensor:
    """``by_gene`` = :func:`rows_by_id` of ``ids`` when the caller has it cached (built in backward otherwise)."""
    if table.dtype != torch.float32 or not table.is_contiguous():
        raise TypeError("embed_gelu: the embedding table must be contiguous fp32")
    if pe.shape[1] != table.shape[1] or pe.shape[1] % 32:
        raise ValueError("embed_gelu: pe width must equal the embedding width and be a multiple of 32")
    return _EmbedGelu.apply(table, ids.to(torch.int32).contiguous(), pe.contiguous(), by_gene)


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, eps):
        _lib.require_cuda(y)
        lib = _lib.load()
        n, c = y.shape
        z = torch.empty((n, c), dtype=y.dtype, device=y.device)
        yp, ldy = _rows(y, c, "y")
        with _lib.on_device(y.device):
            rc = lib.segger_l2norm_fwd(yp, ldy, n, c, eps, z.data_ptr(), c, DTYPE_CODE[y.dtype], _lib.stream_ptr(y.device))
        _lib.check(rc, "segger_l2norm_fwd")
        ctx.save_for_backward(y)
        ctx.eps = eps
        return z

    @staticmethod
    def backward(ctx, gz):
        (y,) = ctx.saved_tensors
        lib = _lib.load()
        n, c = y.shape
        if gz.dtype != y.dtype:
            gz = gz.to(y.dtype)
        if gz.stride(-1) != 1 or gz.dim() != 2:
            gz = gz.contiguous()
        gy = torch.empty((n, c), dtype=y.dtype, device=y.device)
        yp, ldy = _rows(y, c, "y")
        gp, ldg = _rows(gz, c, "gz")
        with _lib.on_device(y.device):
            rc = lib.segger_l2norm_bwd(yp, ldy, gp, ldg, n, c, ctx.eps, gy.data_ptr(), c, DTYPE_CODE[y.dtype],
                                       _lib.stream_ptr(y.device))
        _lib.check(rc, "segger_l2norm_bwd")
        return gy, None


class _L2NormMany(torch.autograd.Function):
    """Row normalisation of several matrices in ONE launch each way (``segger_l2norm_many``)."""

    @staticmethod
    def forward(ctx, eps, *ys):
        lib = _lib.load()
        c, dt, dev = int(ys[0].shape[1]), ys[0].dtype, ys[0].device
        zs = [torch.empty((int(y.shape[0]), c), dtype=dt, device=dev) for y in ys]
        segs = (_lib.L2NormSeg * len(ys))()
        for sg, y, z in zip(segs, ys, zs):
            sg.y, sg.ld_y = _rows(y, c, "y")
            sg.n, sg.out, sg.ld_out = int(y.shape[0]), z.data_ptr(), c
        with _lib.on_device(dev):
            rc = lib.segger_l2norm_many(segs, len(ys), c, eps, DTYPE_CODE[dt], _lib.stream_ptr(dev))
        _lib.check(rc, "segger_l2norm_many")
        ctx.save_for_backward(*ys)
        ctx.eps = eps
        return tuple(zs)

    @staticmethod
    def backward(ctx, *gzs):
        ys = ctx.saved_tensors
        lib = _lib.load()
        c, dt, dev = int(ys[0].shape[1]), ys[0].dtype, ys[0].device
        segs = (_lib.L2NormSeg * len(ys))()
        outs, keep, k = [], [], 0
        for y, gz in zip(ys, gzs):
            if gz is None:
                outs.append(None)
                continue
            if gz.dtype not in (dt, torch.float32):
                gz = gz.to(dt)
            if gz.dim() != 2 or (gz.shape[0] > 1 and gz.stride(1) != 1):
                gz = gz.contiguous()
            gy = torch.empty((int(y.shape[0]), c), dtype=dt, device=dev)
            sg = segs[k]; k += 1
            sg.y, sg.ld_y = _rows(y, c, "y")
            sg.n, sg.out, sg.ld_out = int(y.shape[0]), gy.data_ptr(), c
            sg.gz, sg.ld_gz = _rows(gz, c, "gz")
            sg.gz_f32 = int(gz.dtype == torch.float32 and dt != torch.float32)
            outs.append(gy); keep.append(gz)
        if k:
            with _lib.on_device(dev):
                rc = lib.segger_l2norm_many(segs, k, c, ctx.eps, DTYPE_CODE[dt], _lib.stream_ptr(dev))
            _lib.check(rc, "segger_l2norm_many")
        return (None,) + tuple(outs)


def l2_normalize_many(ys: dict, eps: float = 1e-12) -> dict:
    """``{k: F.normalize(v, dim=-1)}`` for up to four [n_k, C] matrices of one width and dtype in ONE launch (both node
    types of the encoder's tail, ist_encoder.py:331-332); anything else goes through :func:`l2_normalize` per entry."""
    vals = list(ys.values())
    if (1 < len(vals) <= 4 and all(v.dim() == 2 and v.is_cuda for v in vals) and vals[0].shape[1] in (8, 16, 32, 64, 128)
            and vals[0].dtype in DTYPE_CODE and all(v.shape[1] == vals[0].shape[1] and v.dtype == vals[0].dtype for v in vals)
            and all(v.shape[0] > 0 for v in vals)):
        zs = _L2NormMany.apply(float(eps), *vals)
        out = {}
        for k, y, z in zip(ys, vals, zs):
            if torch.is_grad_enabled() and y.requires_grad:
                z._segger_prenorm = (y, float(eps))
            out[k] = z
        return out
    return {k: l2_normalize(v, eps) for k, v in ys.items()}


def l2_normalize(y: Tensor, eps: float = 1e-12) -> Tensor:
    """F.normalize(y, dim=-1) for [n, C] with C in {8,16,32,64,128}; other widths use torch."""
    if y.dim() == 2 and y.shape[1] in (8, 16, 32, 64, 128) and y.dtype in DTYPE_CODE:
        z = _L2Norm.apply(y, float(eps))
        if torch.is_grad_enabled() and y.requires_grad:
            z._segger_prenorm = (y, float(eps))      # (read by ops.loss_head, which may differentiate through y directly)
        return z
    _lib.require_cuda(y)
    return torch.nn.functional.normalize(y.float(), dim=-1, eps=eps).to(y.dtype)
