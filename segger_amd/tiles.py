"""Device-resident tile batcher: what feeds the hot path.

Restates, with tensors that can live in HBM for the whole run,

* ``SquareTiling``           reference ``src/segger/data/tiling.py:238-300`` (+ ``label`` / ``mask``
                             semantics of ``Tiling``, ``:152-196``, for point geometries)
* ``TilePartition``          ``data/partition/dataset.py:340-579`` as used by ``TileFitDataset``
                             (``data/tile_dataset.py:13-153``): nodes permuted so every tile is a
                             contiguous slice, inter-tile edges dropped, O(1) tile slicing, margin ``mask``
* ``best_fit_decreasing`` / ``harmonic_k`` / ``first_fit_decreasing_bucketed`` / ``TileBatchSampler``
                             ``data/partition/sampler.py:11-405`` (batches of tiles up to ``edges_per_batch``)
* ``PredictTiles``           ``data/tile_dataset.py:156-264`` (bbox + margin subgraph, ``predict_mask``)

The reference slices on the host inside DataLoader workers and ships every batch over PCIe; here the
partitioned graph stays on the device and a batch is assembled by index arithmetic on device tensors
(``TilePartition.batch``), so 8 ranks are not limited by Python ``__getitem__`` + collate.
Everything is plain torch (works on CPU tensors too, which is how the unit tests run it).
"""
from __future__ import annotations

import math
import random
from typing import Dict, Iterator, List, Optional, Sequence

import torch
from torch import Tensor

from .hetero import EdgeType, HeteroBatch

NODE_SKIP = ("batch", "num_nodes")


# ------------------------------------------------------------------------------------------ tiling
class SquareTiling:
    """Uniform square grid over the extent of ``positions`` (edge tiles are clipped to the extent)."""

    def __init__(self, positions: Tensor, side_length: float):
        if side_length <= 0:
            raise ValueError(f"side_length must be positive, but got {side_length}.")
        if positions.dim() != 2 or positions.shape[-1] != 2:
            raise ValueError(f"positions must be a tensor of shape (N, 2), but got {positions.shape}.")
        if len(positions) == 0:
            raise ValueError("positions cannot be empty.")
        self.min_x, self.max_x = positions[:, 0].min().item(), positions[:, 0].max().item()
        self.min_y, self.max_y = positions[:, 1].min().item(), positions[:, 1].max().item()
        self.side_length = float(side_length)
        self.nx = max(1, math.ceil((self.max_x - self.min_x) / self.side_length))
        self.ny = max(1, math.ceil((self.max_y - self.min_y) / self.side_length))

    def __len__(self) -> int:
        return self.nx * self.ny

    @property
    def tiles(self) -> Tensor:
        """[T, 4] boxes (x0, y0, x1, y1); tile id = ix * ny + iy (meshgrid 'ij' order of the reference)."""
        ix = torch.arange(self.nx, dtype=torch.float64).repeat_interleave(self.ny)
        iy = torch.arange(self.ny, dtype=torch.float64).repeat(self.nx)
        x0 = self.min_x + ix * self.side_length
        y0 = self.min_y + iy * self.side_length
        x1 = torch.minimum(x0 + self.side_length, torch.tensor(self.max_x, dtype=torch.float64))
        y1 = torch.minimum(y0 + self.side_length, torch.tensor(self.max_y, dtype=torch.float64))
        return torch.stack([x0, y0, x1, y1], 1)

    def _cell(self, pos: Tensor):
        fx = (pos[:, 0].double() - self.min_x) / self.side_length
        fy = (pos[:, 1].double() - self.min_y) / self.side_length
        ix = fx.floor().long().clamp_(0, self.nx - 1)
        iy = fy.floor().long().clamp_(0, self.ny - 1)
        return ix, iy

    def label(self, pos: Tensor) -> Tensor:
        """Tile index of every point ('intersects': boundaries included; a point on a shared edge goes to
        the upper tile, points on the extent's max edge to the last tile); -1 outside the extent."""
        ix, iy = self._cell(pos)
        lab = ix * self.ny + iy
        inside = ((pos[:, 0] >= self.min_x) & (pos[:, 0] <= self.max_x) &
                  (pos[:, 1] >= self.min_y) & (pos[:, 1] <= self.max_y))
        return torch.where(inside, lab, torch.full_like(lab, -1))

    def mask(self, pos: Tensor, margin: float) -> Tensor:
        """True where the point lies strictly inside its tile shrunk by ``margin`` ('contains').  A margin
        that would make a tile vanish is halved until every tile survives (tiling.py:103-127)."""
        if margin < 0:
            raise ValueError(f"The margin must be non-negative, but got {margin}.")
        t = self.tiles
        w = torch.minimum(t[:, 2] - t[:, 0], t[:, 3] - t[:, 1]).min().item()
        eff = float(margin)
        while eff > 0 and 2 * eff >= w:
            eff = eff / 2 if eff > 1e-6 else 0.0
        ix, iy = self._cell(pos)
        tid = ix * self.ny + iy
        box = t.to(pos.device)[tid]
        x, y = pos[:, 0].double(), pos[:, 1].double()
        return ((x > box[:, 0] + eff) & (x < box[:, 2] - eff) & (y > box[:, 1] + eff) & (y < box[:, 3] - eff))


# --------------------------------------------------------------------------------------- partition
class TilePartition:
    """A full-slide graph re-ordered so that tile ``t`` owns nodes ``node_indptr[type][t:t+2]`` and edges
    ``edge_indptr[etype][t:t+2]`` (endpoints stored relative to the tile's first node of each type); edges whose
    endpoints lie in different tiles are dropped (the reference
    does the same, partition/dataset.py:480-494).  ``labels`` maps node type -> tile id per node."""

    def __init__(self, data: HeteroBatch, labels: Dict[str, Tensor], num_tiles: int):
        self.num_tiles = int(num_tiles)
        self.node_perm: Dict[str, Tensor] = {}
        self.node_indptr: Dict[str, Tensor] = {}
        self.node_sizes: Dict[str, Tensor] = {}
        self.edge_indptr: Dict[EdgeType, Tensor] = {}
        self.edge_sizes: Dict[EdgeType, Tensor] = {}
        self.data = HeteroBatch(num_graphs=1)
        inv: Dict[str, Tensor] = {}
        sorted_labels: Dict[str, Tensor] = {}
        for nt in data.node_types:
            lab = labels[nt].long()
            if lab.numel() and (int(lab.min()) < 0 or int(lab.max()) >= self.num_tiles):
                raise ValueError(f"node type '{nt}': tile labels must lie in [0, {self.num_tiles})")
            perm = torch.argsort(lab, stable=True)
            sizes = torch.bincount(lab, minlength=self.num_tiles)
            self.node_perm[nt] = perm
            self.node_sizes[nt] = sizes
            self.node_indptr[nt] = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
            for a, v in data[nt].items():
                if a in NODE_SKIP:
                    continue
                self.data[nt][a] = v.index_select(0, perm) if isinstance(v, Tensor) else v
            i = torch.empty_like(perm)
            i[perm] = torch.arange(perm.numel(), device=perm.device)
            inv[nt] = i
            sorted_labels[nt] = lab[perm]
        for et in data.edge_types:
            s, _, d = et
            ei = data[et].edge_index.long()
            src, dst = inv[s][ei[0]], inv[d][ei[1]]
            ls, ld = sorted_labels[s][src], sorted_labels[d][dst]
            keep = ls == ld
            order = torch.argsort(ls[keep], stable=True)
            src, dst, lab = src[keep][order], dst[keep][order], ls[keep][order]
            sizes = torch.bincount(lab, minlength=self.num_tiles)
            # TILE-LOCAL endpoints (every kept edge is intra-tile): a single-tile batch is then a plain view of this
            # store -- no index arithmetic, no launch -- and a batch of tiles adds each tile's offset inside the batch
            self.data[et]["edge_index"] = torch.stack([src - self.node_indptr[s][lab], dst - self.node_indptr[d][lab]])
            self.edge_sizes[et] = sizes
            self.edge_indptr[et] = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
        # host copies of the pointers: batch assembly computes offsets without device syncs
        self._nptr = {k: v.tolist() for k, v in self.node_indptr.items()}
        self._eptr = {k: v.tolist() for k, v in self.edge_indptr.items()}

    def __len__(self) -> int:
        return self.num_tiles

    # ---- helpers that keep batch assembly free of host <-> device synchronisation ------------------------
    def _zeros(self, nt: str, n: int) -> Tensor:
        """``batch`` vector of a single-tile batch: a view of one persistent zero vector."""
        z = self.__dict__.setdefault("_zero_vec", {}).get(nt)
        if z is None or z.numel() < n:
            z = torch.zeros(max(n, int(max(self._nptr[nt][i + 1] - self._nptr[nt][i] for i in range(self.num_tiles)))),
                            dtype=torch.long, device=self.node_perm[nt].device)
            self._zero_vec[nt] = z
        return z[:n]

    @staticmethod
    def _h2d(values: List[int], device) -> Tensor:
        """A small int64 list on the device through pinned memory (a pageable copy stalls the host on the stream)."""
        if torch.device(device).type != "cuda":
            return torch.tensor(values, dtype=torch.long, device=device)
        return torch.tensor(values, dtype=torch.long).pin_memory().to(device, non_blocking=True)

    # ---- sorted edge views, once per slide ----------------------------------------------------------
    # one `segger_csr_from_coo` call sorts fewer than 2^31 edges (int32 slot / edge ids); a slide-level edge store can be
    # larger (reference _patches.py:1-9 documents >INT_MAX edge masks; neighbors.py:159: "600M transcripts" = 9G kNN edges)
    csr_sort_max_edges = (1 << 31) - (1 << 20)

    def _sort_chunks(self, et: EdgeType) -> List[tuple]:
        """Consecutive tile ranges [t0, t1) whose edges of ``et`` fit one sort.  Tiles are independent graphs with
        contiguous nodes and edges, so the sorted views of a range are the sorted views of its tiles back to back."""
        eptr, cap = self._eptr[et], int(self.csr_sort_max_edges)
        chunks, t0 = [], 0
        for t in range(self.num_tiles):
            if eptr[t + 1] - eptr[t] > cap:
                raise ValueError(f"tile {t} alone holds {eptr[t + 1] - eptr[t]} edges of {et}: more than one sort "
                                 f"takes ({cap}); use smaller tiles")
            if eptr[t + 1] - eptr[t0] > cap:
                chunks.append((t0, t)); t0 = t
        chunks.append((t0, self.num_tiles))
        return chunks

    def build_csr(self, edge_types: Optional[Sequence[EdgeType]] = None) -> None:
        """Sort every edge store ONCE for the whole slide (``segger_csr_from_coo``: by destination and by source)
        and keep the result in tile-local coordinates.  Tiles are independent graphs whose nodes and edges are
        contiguous, so the CSR views of any batch of tiles are concatenations of per-tile slices plus one offset
        per tile: :meth:`batch` then hands the encoder ready-made views and no batch is ever sorted again
        (5 radix sorts per batch otherwise).  An edge store of 2^31 edges or more is sorted in ranges of whole tiles
        (``csr_sort_max_edges``).  Needs the partition on the GPU."""
        from .graph import csr_from_coo, sources_unique
        self.csr_max_tiles = 1
        self._csr: Dict[EdgeType, Dict[str, Dict[str, Tensor]]] = {}
        self._src_unique: Dict[EdgeType, bool] = {}
        for et in (edge_types or list(self.data._edges.keys())):
            s, _, d = et
            ei = self.data[et].edge_index                    # tile-local endpoints
            eptr = self._eptr[et]
            tile_ids = torch.arange(self.num_tiles, device=ei.device)
            # once per slide (one sync here, none per batch): does every source have at most one out-edge?  Then the
            # backward of this edge type needs no by-source view (graph.EdgeGraph, segger_gatv2_bwd_args.src_unique)
            if ei.is_cuda:
                first_src = torch.repeat_interleave(self.node_indptr[s][:-1], self.edge_sizes[et], output_size=int(ei.shape[1]))
                self._src_unique[et] = bool(int(sources_unique(ei[0] + first_src, self.data[s].num_nodes)))
                del first_src
            else:
                self._src_unique[et] = False
            views = {}
            for side, (row_t, col_t, row, col) in {"by_dst": (d, s, ei[1], ei[0]), "by_src": (s, d, ei[0], ei[1])}.items():
                rptr, cptr = self._nptr[row_t], self._nptr[col_t]
                parts = {"ptr": [], "col": [], "eid": []}
                for t0, t1 in self._sort_chunks(et):
                    e0, e1, r0, r1, c0, c1 = eptr[t0], eptr[t1], rptr[t0], rptr[t1], cptr[t0], cptr[t1]
                    tile_of_edge = torch.repeat_interleave(tile_ids[t0:t1], self.edge_sizes[et][t0:t1], output_size=e1 - e0)
                    # chunk-local coordinates for the sort: tile-local id + the tile's first node inside the chunk
                    csr = csr_from_coo(row[e0:e1] + (self.node_indptr[row_t][:-1][tile_of_edge] - r0),
                                       col[e0:e1] + (self.node_indptr[col_t][:-1][tile_of_edge] - c0),
                                       r1 - r0, max(c1 - c0, 1), validate=False)
                    # tile-local coordinates (edges of a tile stay contiguous under the sort: all of them are intra-tile)
                    eid_tile = tile_of_edge[csr.eid.long()]
                    parts["col"].append((csr.col.long() + c0 - self.node_indptr[col_t][:-1][eid_tile]).to(torch.int32))
                    parts["eid"].append((csr.eid.long() + e0 - self.edge_indptr[et][:-1][eid_tile]).to(torch.int32))
                    tile_of_row = torch.repeat_interleave(tile_ids[t0:t1], self.node_sizes[row_t][t0:t1], output_size=r1 - r0)
                    # row starts inside their tile, with every tile's END appended to its rows: tile t owns entries
                    # [nptr[t] + t, nptr[t + 1] + t + 1) -- a complete indptr, so a single-tile batch takes a view
                    starts = csr.indptr[:-1] + e0 - self.edge_indptr[et][:-1][tile_of_row]
                    k = t1 - t0
                    ptr_c = torch.empty(r1 - r0 + k, dtype=starts.dtype, device=starts.device)
                    ptr_c[torch.arange(r1 - r0, device=starts.device) + (tile_of_row - t0)] = starts
                    ptr_c[(self.node_indptr[row_t][t0 + 1:t1 + 1] - r0) + torch.arange(k, device=starts.device)] = \
                        self.edge_sizes[et][t0:t1].to(starts.dtype)
                    parts["ptr"].append(ptr_c)
                views[side] = {k: (v[0] if len(v) == 1 else torch.cat(v)) for k, v in parts.items()}
            self._csr[et] = views

    def _batch_graph(self, et: EdgeType, tile_ids: Sequence[int], base: Dict[str, List[int]], n_nodes: Dict[str, int],
                     need_by_dst: bool = True, need_by_src: bool = True, edge_index: Optional[Tensor] = None):
        """Sorted views of one edge type for a batch of tiles, sliced from the slide-level sort.  ``edge_index`` (the
        batch's COO list) rides along so that a consumer that does need the skipped by-source view -- the generic
        GATv2 kernels have no one-pass backward -- can still sort it on demand (``EdgeGraph.require_by_src``)."""
        from .graph import EdgeCSR, EdgeGraph
        s, _, d = et
        eptr = self._eptr[et]
        dev = self._csr[et]["by_dst"]["col"].device
        e_sizes = [eptr[t + 1] - eptr[t] for t in tile_ids]
        e_base, run = [], 0
        for z in e_sizes:
            e_base.append(run); run += z
        n_edges = run
        single = len(tile_ids) == 1
        out = {"by_dst": None, "by_src": None}
        unique = self._src_unique.get(et, False)
        if need_by_src == "lazy":                            # skipped when the slide-level check allows it
            need_by_src = not unique
        for side, row_t, col_t, need in (("by_dst", d, s, need_by_dst), ("by_src", s, d, need_by_src)):
            if not need:
                continue
            v = self._csr[et][side]
            nptr = self._nptr[row_t]
            if single:                                       # slices are already in batch coordinates
                t = tile_ids[0]
                indptr = v["ptr"][nptr[t] + t:nptr[t + 1] + t + 1]          # the tile's own indptr: a view
                col, eid = v["col"][eptr[t]:eptr[t + 1]], v["eid"][eptr[t]:eptr[t + 1]]
            else:                                            # one H2D copy of all per-tile offsets
                r_sizes = [nptr[t + 1] - nptr[t] for t in tile_ids]
                k = len(tile_ids)
                meta = self._h2d(e_sizes + e_base + r_sizes + list(base[col_t]), dev)
                es, eb, rs, cb = meta[:k], meta[k:2 * k], meta[2 * k:3 * k], meta[3 * k:]
                e_off = torch.repeat_interleave(eb, es, output_size=n_edges)
                ptr = (torch.cat([v["ptr"][nptr[t] + t:nptr[t + 1] + t] for t in tile_ids])
                       + torch.repeat_interleave(eb, rs, output_size=n_nodes[row_t]))
                col = (torch.cat([v["col"][eptr[t]:eptr[t + 1]] for t in tile_ids])
                       + torch.repeat_interleave(cb, es, output_size=n_edges).to(torch.int32))
                eid = torch.cat([v["eid"][eptr[t]:eptr[t + 1]] for t in tile_ids]) + e_off.to(torch.int32)
                indptr = torch.empty(ptr.numel() + 1, dtype=torch.long, device=dev)
                indptr[:-1] = ptr
                indptr[-1:] = n_edges
            out[side] = EdgeCSR(indptr, col, eid, n_nodes[row_t], n_nodes[col_t]).balanced_order()
        return EdgeGraph(out["by_dst"], out["by_src"], n_nodes[s], n_nodes[d], n_edges, edge_index, unique)

    def _persistent_store(self, key: tuple) -> dict:
        """Per-tile-set scratch that survives the batch object (sampler indices, rows-by-gene grouping), kept in an
        LRU of ``persist_max`` entries: a shuffling sampler re-packs the tiles every epoch (``TileBatchSampler``,
        reference data_module.py:344), so almost every tile set is new each epoch and an unbounded store would grow by
        one batch-sized entry per step.  Default bound: two epochs' worth of single-tile batches."""
        from collections import OrderedDict
        store = self.__dict__.setdefault("_persist", OrderedDict())
        cap = getattr(self, "persist_max", None) or max(64, 2 * self.num_tiles)
        entry = store.get(key)
        if entry is None:
            entry = store[key] = {}
            while len(store) > cap:
                store.popitem(last=False)
        else:
            store.move_to_end(key)
        return entry

    def shard(self, tile_ids: Sequence[int]) -> "TilePartition":
        """A partition holding ONLY the given tiles (renumbered 0 .. k-1 in the given order), with its own copies of
        their node / edge slices and of the slide-level CSR slices: what a data-parallel rank keeps resident once it
        knows which packed batches it will train on (tiles are independent graphs: ``partition/dataset.py:480-494``
        drops every inter-tile edge).  ``shard(ids).batch([i, j])`` equals ``self.batch([ids[i], ids[j]])``."""
        ids = [int(t) for t in tile_ids]
        if len(set(ids)) != len(ids) or any(not 0 <= t < self.num_tiles for t in ids):
            raise IndexError("shard: tile ids must be distinct and in range")
        new = TilePartition.__new__(TilePartition)
        new.num_tiles = len(ids)
        new.node_perm, new.node_indptr, new.node_sizes = {}, {}, {}
        new.edge_indptr, new.edge_sizes = {}, {}
        new.data = HeteroBatch(num_graphs=1)
        take = lambda v, ptr: (torch.cat([v[ptr[t]:ptr[t + 1]] for t in ids], 0) if ids else v[:0]).clone()
        for nt, store in self.data._nodes.items():
            ptr = self._nptr[nt]
            sizes = self.node_sizes[nt][ids] if ids else self.node_sizes[nt][:0]
            new.node_sizes[nt] = sizes.clone()
            new.node_indptr[nt] = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
            new.node_perm[nt] = take(self.node_perm[nt], ptr)
            for a, v in store.items():
                new.data[nt][a] = take(v, ptr) if isinstance(v, Tensor) else v
        for et, store in self.data._edges.items():
            ptr = self._eptr[et]
            sizes = self.edge_sizes[et][ids] if ids else self.edge_sizes[et][:0]
            new.edge_sizes[et] = sizes.clone()
            new.edge_indptr[et] = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
            ei = store["edge_index"]
            parts = []
            for t in ids:                                    # (tile-local endpoints: nothing to rebase)
                parts.append(ei[:, ptr[t]:ptr[t + 1]])
            new.data[et]["edge_index"] = torch.cat(parts, 1) if parts else ei[:, :0].clone()
        new._nptr = {k: v.tolist() for k, v in new.node_indptr.items()}
        new._eptr = {k: v.tolist() for k, v in new.edge_indptr.items()}
        if getattr(self, "_csr", None):                     # tile-local coordinates: slices carry over unchanged
            new.csr_max_tiles = self.csr_max_tiles
            new._src_unique = dict(self._src_unique)
            new._csr = {}
            for et, views in self._csr.items():
                s_, _, d_ = et
                new._csr[et] = {}
                for side, row_t in (("by_dst", d_), ("by_src", s_)):
                    v = views[side]
                    rp = self._nptr[row_t]
                    new._csr[et][side] = {"ptr": take(v["ptr"], [rp[t] + t for t in range(self.num_tiles + 1)]),
                                          "col": take(v["col"], self._eptr[et]), "eid": take(v["eid"], self._eptr[et])}
        for k in ("persist_max", "csr_sort_max_edges"):
            if k in self.__dict__:
                setattr(new, k, self.__dict__[k])
        return new

    def resident_bytes(self) -> int:
        """Bytes of every tensor this partition keeps (node / edge stores, permutations, CSR slices)."""
        seen, total = set(), 0

        def add(t):
            nonlocal total
            if isinstance(t, Tensor) and t.data_ptr() not in seen:
                seen.add(t.data_ptr()); total += t.numel() * t.element_size()
        for store in list(self.data._nodes.values()) + list(self.data._edges.values()):
            for v in store.values():
                add(v)
        for d in (self.node_perm, self.node_indptr, self.edge_indptr):
            for v in d.values():
                add(v)
        for views in (getattr(self, "_csr", None) or {}).values():
            for side in views.values():
                for v in side.values():
                    add(v)
        return total

    def checksum(self) -> int:
        """A cheap fingerprint of every integer store batches hand out views of (edge lists, sorted views): equal before
        and after an epoch <=> no consumer wrote through a view.  One reduction per tensor + one sync: debug / tests."""
        total = 0
        tensors = [st["edge_index"] for st in self.data._edges.values()]
        for views in (getattr(self, "_csr", None) or {}).values():
            for side in views.values():
                tensors += list(side.values())
        for i, t in enumerate(tensors):
            if t.numel():
                f = t.reshape(-1)
                total += (i + 1) * (int(f.sum(dtype=torch.int64)) + 3 * int(f[::2].sum(dtype=torch.int64))
                                    + 7 * int(f[1::3].sum(dtype=torch.int64)))
        return total & ((1 << 62) - 1)

    def add_node_attr(self, node_type: str, name: str, value: Tensor, permuted: bool = False) -> None:
        self.data[node_type][name] = value if permuted else value.index_select(0, self.node_perm[node_type])

    def weights(self, mode: str = "edge") -> List[int]:
        """Per-tile packing weight: total edges (all edge types) or total nodes (sampler.py:56-64)."""
        src = self.edge_sizes if mode == "edge" else self.node_sizes
        return torch.stack([v.cpu() for v in src.values()]).sum(0).tolist()

    def tile(self, index: int) -> HeteroBatch:
        return self.batch([index])

    def batch(self, tile_ids: Sequence[int]) -> HeteroBatch:
        """The collated batch of the given tiles (== PyG ``Batch.from_data_list([tile(i) ...])`` for the
        fields of the contract): node slices concatenated, edge ids rebased, ``batch`` = graph id.

        READ-ONLY: a single-tile batch is made of VIEWS of the slide-level stores (node attributes, ``edge_index``, and --
        after :meth:`build_csr` -- the sorted views' ``indptr`` / ``col`` / ``eid``): no launch, no copy per batch.  An
        in-place edit by a consumer (``edge_index.add_``, a self-loop transform, ``sort_``) would corrupt the tile for
        every later epoch and every shard; clone what you need to change.  :meth:`checksum` lets a test or a debug run
        assert that nobody did."""
        tile_ids = [int(t) + (self.num_tiles if t < 0 else 0) for t in tile_ids]
        for t in tile_ids:
            if not 0 <= t < self.num_tiles:
                raise IndexError(f"Index {t} is out of range for dataset with {self.num_tiles} partitions.")
        out = HeteroBatch(num_graphs=len(tile_ids))
        base: Dict[str, List[int]] = {}
        single = len(tile_ids) == 1
        for nt, store in self.data._nodes.items():
            ptr = self._nptr[nt]
            dev = self.node_perm[nt].device
            offs, sizes, run = [], [], 0
            for t in tile_ids:
                offs.append(run)
                sizes.append(ptr[t + 1] - ptr[t])
                run += sizes[-1]
            base[nt] = offs
            for a, v in store.items():
                if isinstance(v, Tensor):
                    if single:                               # a tile is a contiguous slice: a view, no copy
                        out[nt][a] = v[ptr[tile_ids[0]]:ptr[tile_ids[0] + 1]]
                    else:
                        out[nt][a] = torch.cat([v[ptr[t]:ptr[t + 1]] for t in tile_ids], 0) if tile_ids else v[:0]
                else:
                    out[nt][a] = v
            if single:
                out[nt]["batch"] = self._zeros(nt, run)
            else:                                            # output_size: no device -> host sync for the length
                out[nt]["batch"] = torch.repeat_interleave(
                    torch.arange(len(tile_ids), device=dev), self._h2d(sizes, dev), output_size=run)
        for et, store in self.data._edges.items():
            s, _, d = et
            ptr = self._eptr[et]
            ei = store["edge_index"]                         # tile-local endpoints
            if single:
                t = tile_ids[0]
                out[et]["edge_index"] = ei[:, ptr[t]:ptr[t + 1]]          # a view: nothing to rebase
            elif tile_ids:
                e_sizes = [ptr[t + 1] - ptr[t] for t in tile_ids]
                n_e = sum(e_sizes)
                meta = self._h2d(e_sizes + base[s] + base[d], ei.device)
                k = len(tile_ids)
                es = meta[:k]
                shift = torch.stack([meta[k:2 * k], meta[2 * k:]])       # [2, k]: first node of tile j inside the batch
                per_edge = torch.repeat_interleave(shift, es, dim=1, output_size=n_e)
                out[et]["edge_index"] = torch.cat([ei[:, ptr[t]:ptr[t + 1]] for t in tile_ids], 1) + per_edge
            else:
                out[et]["edge_index"] = ei[:, :0]
        from .graph import batch_cache
        # what never changes for this set of tiles (labels, masks -> the loss samplers' indices) survives the batch object
        batch_cache(out)["persistent"] = self._persistent_store(tuple(tile_ids))
        if getattr(self, "_csr", None) and tile_ids:
            # graph.edge_graph() asks this factory before it sorts a batch's edge store itself
            n_nodes = {nt: out[nt].num_nodes for nt in self.data._nodes}
            ids = list(tile_ids)
            mine = {et: (out[et]["edge_index"].data_ptr(), int(out[et]["edge_index"].shape[1])) for et in self._csr}

            def factory(key, edge_index, n_src, n_dst, need_by_dst=True, need_by_src=True):    # (no reference to `out`)
                # single-tile batches are pure slices; for many tiles one radix sort of the batch measured faster
                # than concatenating + re-basing 3 arrays x 18 tile slices per view (50M-tx FOV, 16M-edge batches)
                if len(ids) <= self.csr_max_tiles and mine.get(key) == (edge_index.data_ptr(), int(edge_index.shape[1])):
                    return self._batch_graph(key, ids, base, n_nodes, need_by_dst, need_by_src, edge_index)
                return None
            batch_cache(out)["graph_factory"] = factory
            batch_cache(out)["src_unique"] = self._src_unique      # slide-level: holds for every subset of its edges
        return out


def partition_by_tiling(data: HeteroBatch, tiling: SquareTiling, margin: float, pos_key: str = "pos") -> TilePartition:
    """``TileFitDataset``: label nodes by tile, partition, add the margin ``mask`` (tile_dataset.py:37-60,128-144)."""
    for nt in data.node_types:
        if "mask" in data[nt]:
            raise KeyError(f"Node type '{nt}' in the 'data' object must not contain an attribute 'mask'.")
    labels = {nt: tiling.label(data[nt][pos_key]) for nt in data.node_types}
    part = TilePartition(data, labels, len(tiling))
    for nt in data.node_types:
        part.add_node_attr(nt, "mask", tiling.mask(part.data[nt][pos_key], margin), permuted=True)
    return part


# ------------------------------------------------------------------------------------- bin packing
def _indexed(items: Sequence[float], cap: float, skip_too_big: bool):
    if skip_too_big:
        return [(v, i) for i, v in enumerate(items) if 0 < v <= cap]
    if not all(0 < v <= cap for v in items):
        raise ValueError("All items must be > 0 and <= bin_capacity.")
    return [(v, i) for i, v in enumerate(items)]


def best_fit_decreasing(items: Sequence[float], bin_capacity: float, skip_too_big: bool = False) -> List[List[int]]:
    """Largest first; each item into the open bin it fills most tightly (sampler.py:11-84)."""
    todo = sorted(_indexed(items, bin_capacity, skip_too_big), key=lambda x: x[0], reverse=True)
    bins: List[List[int]] = []
    room: List[float] = []
    for v, i in todo:
        best, slack = -1, float("inf")
        for b, r in enumerate(room):
            if r >= v and r - v < slack:
                best, slack = b, r - v
        if best < 0:
            bins.append([i]); room.append(bin_capacity - v)
        else:
            bins[best].append(i); room[best] -= v
    return bins


def harmonic_k(items: Sequence[float], bin_capacity: float, k: int = 6, skip_too_big: bool = False) -> List[List[int]]:
    """Online Harmonic-k: an item of scaled size in (1/(j+1), 1/j] shares a bin with j-1 peers of its
    class; items <= 1/k are packed first-fit (sampler.py:87-178)."""
    if k < 2:
        raise ValueError("Parameter k must be an integer >= 2.")
    todo = _indexed(items, bin_capacity, skip_too_big)
    done: List[List[int]] = []
    open_class: Dict[int, List[int]] = {}
    small: List[List[int]] = []
    small_room: List[float] = []
    for v, i in todo:
        scaled = v / bin_capacity
        if scaled > 1 / k:
            j = math.floor(1 / scaled)
            open_class.setdefault(j, []).append(i)
            if len(open_class[j]) == j:
                done.append(open_class[j]); open_class[j] = []
        else:
            for b, r in enumerate(small_room):
                if v <= r:
                    small[b].append(i); small_room[b] -= v
                    break
            else:
                small.append([i]); small_room.append(bin_capacity - v)
    done.extend(b for b in open_class.values() if b)
    done.extend(small)
    return done


def first_fit_decreasing_bucketed(items: Sequence[float], bin_capacity: float, skip_too_big: bool = False,
                                  n_buckets: Optional[int] = 1, rng: Optional[random.Random] = None) -> List[List[int]]:
    """First-fit over the descending order (``n_buckets`` None / >= n / >= 2: plain FFD -- see the note on the
    bucket shuffle below; 1: fully random first-fit, what ``PartitionSampler`` uses), sampler.py:186-289."""
    rng = rng or random
    todo = sorted(_indexed(items, bin_capacity, skip_too_big), key=lambda x: x[0], reverse=True)
    n = len(todo)
    if n == 0:
        return []
    if n_buckets is not None and 1 <= n_buckets < n:
        if n_buckets == 1:
            rng.shuffle(todo)
        else:
            gaps = sorted(((todo[i - 1][0] - todo[i][0], i) for i in range(1, n)), reverse=True)
            cuts = sorted({pos for _, pos in gaps[:n_buckets - 1]} | {n})
            start = 0
            for c in cuts:
                # the reference shuffles a temporary slice here (sampler.py:268: ``rng.shuffle(indexed_items[start:i])``):
                # the packing order stays the plain descending one and only the RNG advances.  Reproduced as is, so
                # that the same seed yields the same bins (and the same later draws) as the reference
                rng.shuffle(todo[start:c])
                start = c
    bins: List[List[int]] = []
    room: List[float] = []
    for v, i in todo:
        for b, r in enumerate(room):
            if r >= v:
                bins[b].append(i); room[b] -= v
                break
        else:
            bins.append([i]); room.append(bin_capacity - v)
    return bins


class TileBatchSampler:
    """Batches of tile ids whose summed weight stays within ``max_num`` (PartitionSampler, sampler.py:292-405):
    shuffle -> random order + bucketed first-fit, re-packed every epoch; otherwise best-fit-decreasing once."""

    def __init__(self, partition: TilePartition, max_num: int, mode: str = "edge", shuffle: bool = False,
                 skip_too_big: bool = False, seed: Optional[int] = None):
        if mode not in ("node", "edge"):
            raise ValueError("mode must be 'node' or 'edge'")
        self.weights = partition.weights(mode)
        self.max_num, self.shuffle, self.skip_too_big = max_num, shuffle, skip_too_big
        self._rng = random.Random(seed)
        self._batches: List[List[int]] = []
        self._generate()

    def _generate(self) -> None:
        idx = list(range(len(self.weights)))
        if self.shuffle:
            self._rng.shuffle(idx)
        w = [self.weights[i] for i in idx]
        if self.shuffle:
            bins = first_fit_decreasing_bucketed(w, self.max_num, self.skip_too_big, rng=self._rng)
        else:
            bins = best_fit_decreasing(w, self.max_num, self.skip_too_big)
        self._batches = [[idx[j] for j in b] for b in bins]

    def __iter__(self) -> Iterator[List[int]]:
        yield from self._batches
        if self.shuffle:
            self._generate()

    def __len__(self) -> int:
        return len(self._batches)


# ---------------------------------------------------------------------------------- predict tiles
class PredictTiles:
    """Overlapping prediction tiles: the subgraph of all nodes inside ``tile.buffer(margin)`` with a
    ``predict_mask`` marking nodes inside the tile proper (tile_dataset.py:218-246).  Boxes follow the
    reference's half-open outer test (``>= lo`` and ``< hi``) and closed inner test (``>=`` and ``<=``)."""

    def __init__(self, data: HeteroBatch, tiles: Tensor, margin: float = 0.0):
        missing = [nt for nt in data.node_types if "pos" not in data[nt]]
        if missing:
            raise ValueError(f"Missing 'pos' attribute for node type: {', '.join(missing)}")
        self.data, self.tiles, self.margin = data, tiles, float(margin)

    def __len__(self) -> int:
        return int(self.tiles.shape[0])

    def __getitem__(self, idx: int) -> HeteroBatch:
        if idx < 0 or idx >= len(self):
            raise IndexError(f"Requested {idx}, but tiling only contains {len(self)} tiles.")
        x0, y0, x1, y1 = (float(v) for v in self.tiles[idx])
        m = self.margin
        out = HeteroBatch(num_graphs=1)
        new_id: Dict[str, Tensor] = {}
        for nt, store in self.data._nodes.items():
            pos = store["pos"]
            keep = (pos[:, 0] >= x0 - m) & (pos[:, 0] < x1 + m) & (pos[:, 1] >= y0 - m) & (pos[:, 1] < y1 + m)
            sub = keep.nonzero(as_tuple=False).squeeze(1)
            for a, v in store.items():
                if a in NODE_SKIP:
                    continue
                out[nt][a] = v.index_select(0, sub) if isinstance(v, Tensor) else v
            p = out[nt]["pos"]
            out[nt]["predict_mask"] = (p[:, 0] >= x0) & (p[:, 0] <= x1) & (p[:, 1] >= y0) & (p[:, 1] <= y1)
            out[nt]["batch"] = torch.zeros(sub.numel(), dtype=torch.long, device=pos.device)
            nid = torch.full((pos.shape[0],), -1, dtype=torch.long, device=pos.device)
            nid[sub] = torch.arange(sub.numel(), device=pos.device)
            new_id[nt] = nid
        for et, store in self.data._edges.items():
            s, _, d = et
            ei = store["edge_index"].long()
            a, b = new_id[s][ei[0]], new_id[d][ei[1]]
            ok = (a >= 0) & (b >= 0)
            out[et]["edge_index"] = torch.stack([a[ok], b[ok]])
        return out


class PredictTileIndex(PredictTiles):
    """``PredictTiles`` over the tiles of a :class:`SquareTiling` with work per tile proportional to what the tile
    can contain instead of the whole slide: nodes are binned once (one stable sort), edges by the bin of their source,
    and a prediction tile only looks at the bins its margin can reach.  With ``2 * margin <= side`` every tile is
    binned into 3 x 3 sub-cells -- border strips of the margin's width around an interior -- and a prediction tile takes
    its own nine plus the strips of its eight neighbours that face it (~1.2x its own content for segger's 10 um margin
    on ~220 um tiles); wider margins fall back to whole 3 x 3 tile neighbourhoods (9x).  Returns exactly what
    ``PredictTiles.__getitem__`` returns (same node and edge order).  Needs ``margin <= side_length``."""

    def __init__(self, data: HeteroBatch, tiling: SquareTiling, margin: float = 0.0):
        super().__init__(data, tiling.tiles, margin)
        if margin > tiling.side_length:
            raise ValueError(f"margin ({margin}) must not exceed the tile side ({tiling.side_length})")
        self.nx, self.ny = tiling.nx, tiling.ny
        T = len(tiling)
        self.strips = 2.0 * margin <= tiling.side_length
        self._sub = 9 if self.strips else 1                  # bins per tile
        self._nperm: Dict[str, Tensor] = {}
        self._nptr: Dict[str, List[int]] = {}
        self._new_id: Dict[str, Tensor] = {}
        bin_of: Dict[str, Tensor] = {}
        tiles_t = tiling.tiles
        for nt, store in data._nodes.items():
            pos = store["pos"]
            ix, iy = tiling._cell(pos)                        # clamped: nodes outside the extent go to a border tile
            lab = ix * tiling.ny + iy
            if self.strips:
                # strip index inside the node's own tile: 0 / 2 = within the margin of the low / high border (a hair
                # wider than the margin: the exact box test in __getitem__ decides, this only has to be a superset)
                box = tiles_t.to(pos.device)[lab]
                w = margin * (1.0 + 1e-6) + 1e-6 * tiling.side_length
                sx = (pos[:, 0] >= box[:, 2] - w).long() * 2
                sx = torch.where(pos[:, 0] < box[:, 0] + w, torch.zeros_like(sx), torch.where(sx == 2, sx, torch.ones_like(sx)))
                sy = (pos[:, 1] >= box[:, 3] - w).long() * 2
                sy = torch.where(pos[:, 1] < box[:, 1] + w, torch.zeros_like(sy), torch.where(sy == 2, sy, torch.ones_like(sy)))
                # a tile narrower than two margins cannot happen here (2 * margin <= side), but low wins on a tie
                lab = lab * 9 + sx * 3 + sy
            bin_of[nt] = lab
            self._nperm[nt] = torch.argsort(lab, stable=True)
            sizes = torch.bincount(lab, minlength=T * self._sub)
            self._nptr[nt] = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)]).tolist()
            self._new_id[nt] = torch.full((pos.shape[0],), -1, dtype=torch.long, device=pos.device)
        self._edges: Dict[EdgeType, Tensor] = {}
        self._eptr: Dict[EdgeType, List[int]] = {}
        for et, store in data._edges.items():
            ei = store["edge_index"].long()
            lab = bin_of[et[0]][ei[0]]
            order = torch.argsort(lab, stable=True)
            self._edges[et] = torch.cat([ei[:, order], order[None]], 0)      # rows: src, dst, original edge id
            sizes = torch.bincount(lab, minlength=T * self._sub)
            self._eptr[et] = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)]).tolist()

    def _bins(self, idx: int) -> List[int]:
        """Bins a node of prediction tile ``idx`` can lie in (a node of a neighbouring tile must be within the margin of
        the shared border, i.e. in the strip facing this tile)."""
        ix, iy = idx // self.ny, idx % self.ny
        out: List[int] = []
        for jx in range(max(ix - 1, 0), min(ix + 2, self.nx)):
            for jy in range(max(iy - 1, 0), min(iy + 2, self.ny)):
                t = jx * self.ny + jy
                if not self.strips:
                    out.append(t)
                    continue
                dx, dy = jx - ix, jy - iy
                xs = (2,) if dx < 0 else (0,) if dx > 0 else (0, 1, 2)
                ys = (2,) if dy < 0 else (0,) if dy > 0 else (0, 1, 2)
                out.extend(t * 9 + sx * 3 + sy for sx in xs for sy in ys)
        return out

    def __getitem__(self, idx: int) -> HeteroBatch:
        if idx < 0 or idx >= len(self):
            raise IndexError(f"Requested {idx}, but tiling only contains {len(self)} tiles.")
        x0, y0, x1, y1 = (float(v) for v in self.tiles[idx])
        m = self.margin
        neigh = self._bins(idx)
        out = HeteroBatch(num_graphs=1)
        subs: Dict[str, Tensor] = {}
        for nt, store in self.data._nodes.items():
            ptr, perm = self._nptr[nt], self._nperm[nt]
            cand = torch.cat([perm[ptr[t]:ptr[t + 1]] for t in neigh])
            pos = store["pos"].index_select(0, cand)
            keep = (pos[:, 0] >= x0 - m) & (pos[:, 0] < x1 + m) & (pos[:, 1] >= y0 - m) & (pos[:, 1] < y1 + m)
            sub = cand[keep].sort().values                  # ascending original id, like a boolean-mask subgraph
            subs[nt] = sub
            for a, v in store.items():
                if a in NODE_SKIP:
                    continue
                out[nt][a] = v.index_select(0, sub) if isinstance(v, Tensor) else v
            p = out[nt]["pos"]
            out[nt]["predict_mask"] = (p[:, 0] >= x0) & (p[:, 0] <= x1) & (p[:, 1] >= y0) & (p[:, 1] <= y1)
            out[nt]["batch"] = torch.zeros(sub.numel(), dtype=torch.long, device=pos.device)
            self._new_id[nt][sub] = torch.arange(sub.numel(), device=pos.device)
        for et in self.data._edges:
            s, _, d = et
            ptr, tab = self._eptr[et], self._edges[et]
            e = torch.cat([tab[:, ptr[t]:ptr[t + 1]] for t in neigh], 1)
            a, b = self._new_id[s][e[0]], self._new_id[d][e[1]]
            kept = ((a >= 0) & (b >= 0)).nonzero(as_tuple=False).squeeze(1)      # one compaction (one sync) per type
            kept = kept[torch.argsort(e[2][kept])]                                # back to the original edge order
            out[et]["edge_index"] = torch.stack([a[kept], b[kept]])
        for nt, sub in subs.items():
            self._new_id[nt][sub] = -1
        return out
