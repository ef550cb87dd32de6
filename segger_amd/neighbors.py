"""Graph construction on the device: kNN edges without leaving HBM.

Restates reference ``src/segger/data/utils/neighbors.py``:

* ``kdtree_neighbors`` (``:122-163``: scipy ``KDTree.query(k, distance_upper_bound)``, chunked, CPU)
  -> :func:`knn_grid` over the HIP kernel ``segger_knn_grid`` (exact, uniform grid);
* ``knn_to_edge_index`` (``:54-92``): dense neighbour table with padding -> COO ``edge_index`` whose
  row 0 is the QUERY point (source) and row 1 the neighbour (target);
* ``setup_transcripts_graph`` (``:166-180``) -> :func:`transcripts_graph`.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib


def knn_grid(points: Tensor, k: int, max_dist: float = math.inf, query: Optional[Tensor] = None,
             return_dist: bool = False, points_per_cell: float = 2.0) -> Tuple[Tensor, Optional[Tensor]]:
    """Neighbour table ``[m, k]`` (int32; ids into ``points`` sorted by distance; ``len(points)`` = padding)."""
    _lib.require_cuda(points)
    lib = _lib.load()
    dev = points.device
    pts = points.to(torch.float32).contiguous()
    q = pts if query is None else query.to(device=dev, dtype=torch.float32).contiguous()
    n, m = int(pts.shape[0]), int(q.shape[0])
    nbr = torch.empty((m, k), dtype=torch.int32, device=dev)
    dist = torch.empty((m, k), dtype=torch.float32, device=dev) if return_dist else None
    if m == 0:
        return nbr, dist
    if n == 0:
        nbr.fill_(0)
        if dist is not None:
            dist.fill_(math.inf)
        return nbr, dist
    both = pts if query is None else torch.cat([pts, q])
    lo, hi = both.min(0).values, both.max(0).values
    x0, y0, x1, y1 = (float(v) for v in torch.stack([lo, hi]).flatten().tolist())    # one host sync per graph
    w, h = max(x1 - x0, 1e-6), max(y1 - y0, 1e-6)
    cell = math.sqrt(w * h * points_per_cell / n)
    if math.isfinite(max_dist):
        cell = min(cell, max_dist)
    cell = max(cell, math.sqrt(w * h / (4.0 * n)), max(w, h) / 30000.0)              # <= 4n cells, < 2^31 cells
    nx, ny = int(w / cell) + 1, int(h / cell) + 1
    ws_bytes = lib.segger_knn_workspace_bytes(n, nx, ny)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        rc = lib.segger_knn_grid(pts.data_ptr(), n, None if query is None else q.data_ptr(), m, k, float(max_dist),
                                 x0, y0, cell, nx, ny, nbr.data_ptr(), _lib.ptr(dist), ws.data_ptr(), ws_bytes,
                                 _lib.stream_ptr(dev))
    _lib.check(rc, "segger_knn_grid")
    return nbr, dist


def knn_to_edge_index(neighbor_table: Tensor, padding_value: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """Dense ``[N, K]`` neighbour table -> ``(edge_index [2, E] int64, index_ptr [N + 1])``; entries equal to the
    padding value (default N) are dropped; edges keep row-major order (query 0's neighbours first)."""
    n, k = neighbor_table.shape
    if padding_value is None:
        padding_value = n
    valid = neighbor_table != padding_value
    flat = valid.reshape(-1).nonzero(as_tuple=False).squeeze(1)
    col = neighbor_table.reshape(-1)[flat].long()
    row = torch.div(flat, k, rounding_mode="floor")
    deg = valid.sum(1)
    indptr = torch.cat([deg.new_zeros(1), deg.cumsum(0)])
    return torch.stack([row, col]), indptr


def transcripts_graph(pos: Tensor, max_k: int, max_dist: float = math.inf) -> Tensor:
    """``tx-neighbors-tx`` edges: every transcript -> its ``max_k`` nearest transcripts (itself included)
    within ``max_dist``."""
    nbr, _ = knn_grid(pos, max_k, max_dist)
    ei, _ = knn_to_edge_index(nbr, padding_value=int(pos.shape[0]))
    return ei


def prediction_graph_uniform(tx_pos: Tensor, bd_pos: Tensor, max_k: int, max_dist: float = math.inf) -> Tensor:
    """``setup_prediction_graph(mode='uniform')`` (``neighbors.py:213-221``): kNN from boundary centroids
    (queries) into the transcripts (points); row 0 = query (boundary) id, row 1 = transcript id."""
    nbr, _ = knn_grid(tx_pos, max_k, max_dist, query=bd_pos)
    ei, _ = knn_to_edge_index(nbr, padding_value=int(tx_pos.shape[0]))
    return ei
