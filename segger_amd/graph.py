"""Device-resident CSR views of a batch's edge stores.

PyG re-derives the grouping of edges by destination inside every conv call
(scatter kernels keyed on ``edge_index[1]``).  Here each edge type is sorted
ONCE per batch by ``segger_csr_from_coo`` (HIP radix sort) into

* ``by_dst``: rows = destination nodes, cols = sources  -> forward + dst-side backward
* ``by_src``: rows = source nodes, cols = destinations  -> src-side backward, prediction head

and cached on the batch object, so the 4 layers x (forward + backward) reuse it.
"""
from __future__ import annotations

import weakref
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .hetero import EdgeType


# Degree-balanced row visiting order (segger_csr_row_order): within windows of 64 consecutive rows the rows are visited by
# descending degree, so the four rows that share a wave have near-equal length.  A wave walks 4-edge batches until its
# LONGEST row is done; on the C2 tile's by-destination view (kNN in-degrees: mean 15, sd 3.3) that costs 12.9 % of the
# forward's / destination pass's batch iterations, the window order leaves 2.1 % (profiles/r06_aggregation_valu_budget.txt:
# SQ_INSTS_VALU -8.7 % / -8.9 %, destination pass 0.871 -> 0.817 ms, forward 0.634 -> 0.618).  Round 2 had measured the
# same order 3-10 % SLOWER (34 % more L2 requests: rows of one wave no longer adjacent) with the kernels of that round; with
# prefetching walks and 4 rows in flight the instruction count decides.  By-source views of a kNN graph have constant
# out-degree (nothing to balance: 0.655 -> 0.650 ms) and stay in natural order; small views (captured 1M-edge batches:
# kernels of 10-40 us) too.
ROW_ORDER_WINDOW = 0         # by-source views: rows per balancing window (power of two <= 64); 0 = natural order
ROW_ORDER_WINDOW_DST = 64    # by-destination views of at least ROW_ORDER_MIN_ROWS rows
ROW_ORDER_MIN_ROWS = 1 << 17
# ... and only the BACKWARD's destination pass follows it: inside the training step the forward (pair launch) measured
# 617 -> 652 us per layer with the order while the destination pass gained 837 -> 807 (rocprofv3 kernel trace of
# tools/bench_step.py, both orders on one box) -- alone, the forward had gained 3 %.
ROW_ORDER_FORWARD = False
WAVE_PER_ROW_DEGREE = 32     # csrc/gatv2.hip kWavePerRowDegree: from this average degree a whole wave walks one row


@dataclass
class EdgeCSR:
    indptr: Tensor   # int64 [n_rows + 1]
    col: Tensor      # int32 [n_edges]
    eid: Tensor      # int32 [n_edges]  original COO position of each slot
    n_rows: int
    n_cols: int
    order: Optional[Tensor] = None   # int32 [n_rows]: degree-balanced visiting order (balanced_order), or None
    # block tables of the NATURAL visiting order (block_tables()): (blk_cnt, blk_src, col_local) or None
    tables: Optional[tuple] = None

    @property
    def n_edges(self) -> int:
        return int(self.col.shape[0])

    def c_struct(self, ordered: bool = True) -> _lib.Csr:
        """``ordered=False``: the same view without its visiting order (the forward walks natural order, see
        ``ROW_ORDER_FORWARD``)."""
        # built once per view (4 layers x 3 passes ask for it every step) and kept while the arrays are the SAME tensor
        # objects at the SAME addresses: the entry holds the tensors themselves (an id() of a freed tensor can be reused by
        # a new one) and their data pointers (resize_ / set_ / `.data =` keep the object and move the storage)
        order = self.order if ordered else None
        tables = self.tables if not ordered else None           # (the tables describe the natural order)
        slot = "_c_struct" if ordered else "_c_struct_natural"
        hit = self.__dict__.get(slot)
        if hit is not None:
            ts, ptrs, dims, c = hit
            if (ts[0] is self.indptr and ts[1] is self.col and ts[2] is self.eid and ts[3] is order and ts[4] is tables
                    and dims == (self.n_rows, self.n_cols)
                    and ptrs == (self.indptr.data_ptr(), self.col.data_ptr(), self.eid.data_ptr(), _lib.ptr(order))):
                return c
        ptrs = (self.indptr.data_ptr(), self.col.data_ptr(), self.eid.data_ptr(), _lib.ptr(order))
        c = _lib.Csr(ptrs[0], ptrs[1] if self.n_edges else None, ptrs[2] if self.n_edges else None,
                     self.n_rows, self.n_cols, self.n_edges, ptrs[3])
        if tables is not None:
            c.blk_cnt, c.blk_src, c.col_local = (t.data_ptr() for t in tables)
        self.__dict__[slot] = ((self.indptr, self.col, self.eid, order, tables), ptrs, (self.n_rows, self.n_cols), c)
        return c

    def block_tables(self) -> "EdgeCSR":
        """Attach the block tables of ``segger_csr_block_tables`` (natural visiting order): per 16-row workgroup the distinct
        column ids its rows gather + the LDS slot of every edge.  One host read of the overflow flag (a block with more than
        128 distinct ids / 1024 edges: the view then keeps the plain kernels)."""
        if self.tables is None and self.n_rows > 0 and self.n_edges > 0:
            dev = self.indptr.device
            nblk = (self.n_rows + 15) // 16
            cnt = torch.empty(nblk, dtype=torch.int32, device=dev)
            src = torch.empty((nblk, 128), dtype=torch.int32, device=dev)
            loc = torch.empty(self.n_edges, dtype=torch.uint8, device=dev)
            over = torch.zeros(1, dtype=torch.int32, device=dev)
            with _lib.on_device(dev):
                rc = _lib.load().segger_csr_block_tables(self.indptr.data_ptr(), self.col.data_ptr(), None, self.n_rows,
                                                         self.n_edges, cnt.data_ptr(), src.data_ptr(), loc.data_ptr(),
                                                         over.data_ptr(), _lib.stream_ptr(dev))
            _lib.check(rc, "segger_csr_block_tables")
            if int(over.item()) == 0:
                self.tables = (cnt, src, loc)
        return self

    def balanced_order(self, window: Optional[int] = None) -> "EdgeCSR":
        """Attach the visiting order of ``segger_csr_row_order`` (rows of near-equal degree share a wave; computed
        once per view, results never depend on it).  Skipped for views the kernels walk one row per wave."""
        if window is None:
            window = ROW_ORDER_WINDOW
        if (window > 0 and self.order is None and self.n_rows > 0
                and self.n_edges < WAVE_PER_ROW_DEGREE * self.n_rows):
            order = torch.empty(self.n_rows, dtype=torch.int32, device=self.indptr.device)
            with _lib.on_device(order.device):
                rc = _lib.load().segger_csr_row_order(self.indptr.data_ptr(), self.n_rows, window, order.data_ptr(),
                                                      _lib.stream_ptr(order.device))
            _lib.check(rc, "segger_csr_row_order")
            self.order = order
        return self


# ---- deferred validation -------------------------------------------------------------------------------
# segger_csr_from_coo counts edges whose node ids fall outside the graph and CLAMPS them (no kernel ever reads out
# of bounds), so the check needs no host sync in the middle of a step: the counter is copied to pinned memory
# asynchronously and examined when a later call (or flush_validation) finds the copy complete.
_PIN_SLOTS = 256
_pin: Optional[Tensor] = None
_pin_next = 0
_pending: list = []            # (event, slot, message)


def _poll_validation(block: bool = False) -> None:
    global _pending
    keep = []
    for ev, slot, msg in _pending:
        if block:
            ev.synchronize()
        if ev.query():
            n_bad = int(_pin[slot])
            if n_bad:
                _pending = []
                raise IndexError(msg.format(n=n_bad))
        else:
            keep.append((ev, slot, msg))
    _pending = keep


def flush_validation() -> None:
    """Wait for every outstanding deferred edge-index check and raise IndexError if one failed."""
    _poll_validation(block=True)


def _defer_validation(bad: Tensor, msg: str) -> None:
    global _pin, _pin_next
    if _pin is None:
        _pin = torch.zeros(_PIN_SLOTS, dtype=torch.int32).pin_memory()
    if len(_pending) >= _PIN_SLOTS - 1:
        _poll_validation(block=True)
    slot = _pin_next
    _pin_next = (_pin_next + 1) % _PIN_SLOTS
    _pin[slot:slot + 1].copy_(bad, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(bad.device))
    _pending.append((ev, slot, msg))
    _poll_validation()


def csr_from_coo(row: Tensor, col: Tensor, n_rows: int, n_cols: int, validate=True) -> EdgeCSR:
    """Stable sort of COO edges by ``row`` on the device (rows/cols as in ``edge_index``).
    ``validate``: True = check the node ids now (one host sync), ``"deferred"`` = check without a sync (raises
    from a later call or from :func:`flush_validation`; out-of-range ids are clamped meanwhile), False = never."""
    _lib.require_cuda(row, col)
    lib = _lib.load()
    dev = row.device
    row = row.to(torch.int64).contiguous()
    col = col.to(torch.int64).contiguous()
    E = int(row.shape[0])
    indptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
    ccol = torch.empty(E, dtype=torch.int32, device=dev)
    eid = torch.empty(E, dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev) if validate else None
    ws_bytes = lib.segger_csr_from_coo_workspace_bytes(E, n_rows)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        rc = lib.segger_csr_from_coo(row.data_ptr(), col.data_ptr(), E, n_rows, n_cols,
                                     indptr.data_ptr(), ccol.data_ptr(), eid.data_ptr(), _lib.ptr(bad),
                                     ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev))
    _lib.check(rc, "segger_csr_from_coo")
    msg = f"edge_index holds {{n}} edge(s) with node ids outside [0,{n_rows}) x [0,{n_cols})"
    if validate == "deferred":
        with _lib.on_device(dev):
            _defer_validation(bad, msg)
    elif validate:
        n_bad = int(bad.item())          # one host sync per edge type per batch
        if n_bad:
            raise IndexError(msg.format(n=n_bad))
    return EdgeCSR(indptr, ccol, eid, n_rows, n_cols)


class DeferredFlag:
    """A device int32 read on the host WITHOUT a device-wide sync: copied to pinned memory when created, examined
    when first asked for -- by then (the backward of the step whose forward created it) the copy has long landed, and
    if not, only the copy's own event is waited for."""

    def __init__(self, dev_value: Tensor):
        self._host = _pinned_slot(self)
        self._host.copy_(dev_value, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record(torch.cuda.current_stream(dev_value.device))
        self._value: Optional[int] = None

    def get(self) -> int:
        if self._value is None:
            self._event.synchronize()
            self._value = int(self._host[0])
        return self._value


_flag_ring: Optional[Tensor] = None
_flag_live: list = []
_flag_next = 0


def _pinned_slot(owner) -> Tensor:
    """One int32 of a pinned ring (pinned allocations are too slow to make per batch)."""
    global _flag_ring, _flag_next, _flag_live
    if _flag_ring is None:
        _flag_ring = torch.zeros(_PIN_SLOTS, dtype=torch.int32).pin_memory()
        _flag_live = [None] * _PIN_SLOTS
    slot = _flag_next
    _flag_next = (_flag_next + 1) % _PIN_SLOTS
    old = _flag_live[slot]() if _flag_live[slot] is not None else None
    if old is not None:
        old.get()                       # the ring came round: settle the flag that still owns this slot
    _flag_live[slot] = weakref.ref(owner)
    return _flag_ring[slot:slot + 1]


@dataclass
class EdgeGraph:
    """Both sorted views of one edge type (src type -> dst type).  ``by_src`` may be deferred
    (``need_by_src="lazy"``): the graph then carries the COO edge list and a flag telling whether every source has
    at most one out-edge -- if so the backward needs no by-source view at all (``segger_gatv2_bwd_args.src_unique``),
    otherwise :meth:`require_by_src` sorts it on first use."""
    by_dst: Optional[EdgeCSR]
    by_src: Optional[EdgeCSR]
    n_src: int
    n_dst: int
    n_edges: int
    edge_index: Optional[Tensor] = None
    unique_flag: Optional[object] = None        # DeferredFlag, or a plain bool when known

    def src_unique(self) -> bool:
        f = self.unique_flag
        if f is None:
            return False
        if isinstance(f, DeferredFlag):
            f = self.unique_flag = bool(f.get())
        return bool(f)

    def require_by_src(self) -> EdgeCSR:
        if self.by_src is None:
            if self.edge_index is None:
                raise RuntimeError("this EdgeGraph was built without its by-source view")
            src, dst = self.edge_index[0], self.edge_index[1]
            self.by_src = csr_from_coo(src, dst, self.n_src, self.n_dst, False).balanced_order()
        return self.by_src


def sources_unique(src: Tensor, n_src: int) -> Tensor:
    """Device int32[1]: 1 when no node id occurs twice in ``src`` (``segger_coo_unique``)."""
    _lib.require_cuda(src)
    lib = _lib.load()
    dev = src.device
    src = src.to(torch.int64).contiguous()
    marks = torch.empty(max(n_src, 1), dtype=torch.int32, device=dev)
    out = torch.empty(1, dtype=torch.int32, device=dev)
    with _lib.on_device(dev):
        rc = lib.segger_coo_unique(src.data_ptr(), int(src.numel()), int(n_src), marks.data_ptr(), out.data_ptr(),
                                   _lib.stream_ptr(dev))
    _lib.check(rc, "segger_coo_unique")
    return out


def build_edge_graph(edge_index: Tensor, n_src: int, n_dst: int, *, need_by_dst: bool = True,
                     need_by_src=True, validate=True, known_unique: bool = False) -> EdgeGraph:
    """``need_by_src``: True, False, or ``"lazy"`` (see :class:`EdgeGraph`).  ``known_unique``: the caller vouches that
    no source has two out-edges (e.g. checked once for the whole slide): "lazy" then needs no check of its own."""
    src, dst = edge_index[0], edge_index[1]
    by_dst = None
    if need_by_dst:
        by_dst = csr_from_coo(dst, src, n_dst, n_src, validate)
        by_dst.balanced_order(ROW_ORDER_WINDOW_DST if n_dst >= ROW_ORDER_MIN_ROWS else ROW_ORDER_WINDOW)
    if need_by_src == "lazy" and known_unique:
        return EdgeGraph(by_dst, None, n_src, n_dst, int(edge_index.shape[1]), edge_index, True)
    if need_by_src == "lazy":
        with _lib.on_device(edge_index.device):
            flag = DeferredFlag(sources_unique(src, n_src))
        return EdgeGraph(by_dst, None, n_src, n_dst, int(edge_index.shape[1]), edge_index, flag)
    # both views hold the same edges: one check is enough
    by_src = (csr_from_coo(src, dst, n_src, n_dst, False if need_by_dst else validate).balanced_order()
              if need_by_src else None)
    return EdgeGraph(by_dst, by_src, n_src, n_dst, int(edge_index.shape[1]))


def padded_view_segments(dst: EdgeCSR, src: EdgeCSR, n_real: int, pad_cols=None) -> list:
    """``ops.stage`` segments that write the CSR view ``src`` (``n_real`` rows) into the larger static view ``dst`` of a
    captured graph: real rows / slots are copied, the padding edges are spread evenly over the dummy rows
    n_real .. dst.n_rows-1 (q each) with columns = the row itself (``pad_cols`` None: self-loops) or given as a fill
    ``(kind, a, b)``, and edge ids continuing after the real ones.  No sort: the padded view is a valid CSR as is."""
    e = src.n_edges
    pad, n_dummy = dst.n_edges - e, dst.n_rows - n_real
    if pad < 0 or n_dummy < 1:
        raise ValueError("padded_view_segments: the static view must hold every edge and at least one dummy row")
    q = max(-(-pad // n_dummy), 1)
    col_fill = ("div", n_real, q) if pad_cols is None else tuple(pad_cols)
    return [(dst.indptr, src.indptr, "ramp", e, q, pad),
            (dst.col, src.col, *col_fill, 0),
            (dst.eid, src.eid, "div", e, 1, 0)]


def batch_cache(batch) -> dict:
    """Per-batch scratch dict; works for HeteroBatch and for a PyG Batch."""
    c = getattr(batch, "_segger_amd_cache", None)
    if c is None:
        c = {}
        try:
            object.__setattr__(batch, "_segger_amd_cache", c)
        except Exception:
            pass
    return c


def edge_graph(cache: Optional[dict], key, edge_index: Tensor, n_src: int, n_dst: int, **kw) -> EdgeGraph:
    if cache is None:
        return build_edge_graph(edge_index, n_src, n_dst, **kw)
    k = ("graph", key, edge_index.data_ptr(), int(edge_index.shape[1]), n_src, n_dst,
         kw.get("need_by_dst", True), kw.get("need_by_src", True))
    g = cache.get(k)
    if g is None:
        factory = cache.get("graph_factory")         # tiles.TilePartition: views sliced from a once-per-slide sort
        if factory is not None:
            g = factory(key, edge_index, n_src, n_dst, kw.get("need_by_dst", True), kw.get("need_by_src", True))
        if g is None:
            unique = bool(cache.get("src_unique", {}).get(key, False))     # tiles.TilePartition: slide-level check
            g = build_edge_graph(edge_index, n_src, n_dst, known_unique=unique, **kw)
        cache[k] = g
    return g
