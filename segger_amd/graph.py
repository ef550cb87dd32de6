"""Device-resident CSR views of a batch's edge stores.

PyG re-derives the grouping of edges by destination inside every conv call
(scatter kernels keyed on ``edge_index[1]``).  Here each edge type is sorted
ONCE per batch by ``segger_csr_from_coo`` (HIP radix sort) into

* ``by_dst``: rows = destination nodes, cols = sources  -> forward + dst-side backward
* ``by_src``: rows = source nodes, cols = destinations  -> src-side backward, prediction head

and cached on the batch object, so the 4 layers x (forward + backward) reuse it.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .hetero import EdgeType


@dataclass
class EdgeCSR:
    indptr: Tensor   # int64 [n_rows + 1]
    col: Tensor      # int32 [n_edges]
    eid: Tensor      # int32 [n_edges]  original COO position of each slot
    n_rows: int
    n_cols: int

    @property
    def n_edges(self) -> int:
        return int(self.col.shape[0])

    def c_struct(self) -> _lib.Csr:
        return _lib.Csr(self.indptr.data_ptr(),
                        self.col.data_ptr() if self.n_edges else None,
                        self.eid.data_ptr() if self.n_edges else None,
                        self.n_rows, self.n_cols, self.n_edges)


def csr_from_coo(row: Tensor, col: Tensor, n_rows: int, n_cols: int, validate: bool = True) -> EdgeCSR:
    """Stable sort of COO edges by ``row`` on the device (rows/cols as in ``edge_index``)."""
    _lib.require_cuda(row, col)
    lib = _lib.load()
    dev = row.device
    row = row.to(torch.int64).contiguous()
    col = col.to(torch.int64).contiguous()
    E = int(row.shape[0])
    indptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
    ccol = torch.empty(E, dtype=torch.int32, device=dev)
    eid = torch.empty(E, dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.segger_csr_from_coo_workspace_bytes(E, n_rows)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.segger_csr_from_coo(row.data_ptr(), col.data_ptr(), E, n_rows, n_cols,
                                     indptr.data_ptr(), ccol.data_ptr(), eid.data_ptr(), bad.data_ptr(),
                                     ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev))
    _lib.check(rc, "segger_csr_from_coo")
    if validate:
        n_bad = int(bad.item())          # one host sync per edge type per batch
        if n_bad:
            raise IndexError(f"edge_index holds {n_bad} edge(s) with node ids outside "
                             f"[0,{n_rows}) x [0,{n_cols})")
    return EdgeCSR(indptr, ccol, eid, n_rows, n_cols)


@dataclass
class EdgeGraph:
    """Both sorted views of one edge type (src type -> dst type)."""
    by_dst: Optional[EdgeCSR]
    by_src: Optional[EdgeCSR]
    n_src: int
    n_dst: int
    n_edges: int


def build_edge_graph(edge_index: Tensor, n_src: int, n_dst: int, *, need_by_dst: bool = True,
                     need_by_src: bool = True, validate: bool = True) -> EdgeGraph:
    src, dst = edge_index[0], edge_index[1]
    by_dst = csr_from_coo(dst, src, n_dst, n_src, validate) if need_by_dst else None
    by_src = csr_from_coo(src, dst, n_src, n_dst, validate and not need_by_dst) if need_by_src else None
    return EdgeGraph(by_dst, by_src, n_src, n_dst, int(edge_index.shape[1]))


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
            g = build_edge_graph(edge_index, n_src, n_dst, **kw)
        cache[k] = g
    return g
