"""The HeteroData / Batch contract the hot path reads.

segger hands ``LitISTEncoder`` a torch_geometric ``Batch`` of ``HeteroData``
tiles (built at reference ``src/segger/data/utils/heterodata.py:114-162``,
masked at ``data/tile_dataset.py:128-144,244-246``, collated by PyG's
``DataLoader``, ``data/data_module.py:346-384``).  The path touches only:

* ``batch.x_dict / pos_dict / batch_dict / edge_index_dict``
* ``batch['tx']['mask' | 'cluster' | 'index' | 'x' | 'predict_mask']``,
  ``batch['bd']['mask' | 'cluster' | 'index']``, ``batch['tx'].num_nodes``
* ``batch['tx','belongs','bd'].edge_index``, ``batch['tx','neighbors','bd'].edge_index``
* ``batch.num_graphs``

so a real PyG ``Batch`` works unchanged (everything here is duck-typed), and
:class:`HeteroBatch` provides the same surface where PyG is not installed
(it is not in this image).  :func:`collate` restates PyG's collate for these
fields: node attributes concatenated, ``edge_index`` shifted by the running
node offsets, and a per-node graph id ``batch`` vector.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple, Union

import torch
from torch import Tensor

EdgeType = Tuple[str, str, str]
TX_TX: EdgeType = ("tx", "neighbors", "tx")
TX_BD: EdgeType = ("tx", "belongs", "bd")
TX_NB_BD: EdgeType = ("tx", "neighbors", "bd")


class _Store(dict):
    """Attribute store: ``s['x']`` and ``s.x`` both work (like PyG storages)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class NodeStore(_Store):
    @property
    def num_nodes(self) -> int:
        if "num_nodes" in self:
            return int(self["num_nodes"])
        for k in ("x", "pos", "index"):
            if k in self:
                return int(self[k].shape[0])
        raise AttributeError("num_nodes")


class EdgeStore(_Store):
    @property
    def num_edges(self) -> int:
        return int(self["edge_index"].shape[1])


class HeteroBatch:
    def __init__(self, num_graphs: int = 1):
        self._nodes: Dict[str, NodeStore] = {}
        self._edges: Dict[EdgeType, EdgeStore] = {}
        self.num_graphs = num_graphs
        # per-batch derived structures (CSR/CSC of each edge type) cached by
        # segger_amd.graph so they are built once, not once per layer
        self._cache: dict = {}

    # -- PyG-style access ---------------------------------------------------
    def __getitem__(self, key: Union[str, EdgeType]):
        if isinstance(key, tuple):
            return self._edges.setdefault(tuple(key), EdgeStore())
        return self._nodes.setdefault(key, NodeStore())

    @property
    def node_types(self) -> List[str]:
        return list(self._nodes)

    @property
    def edge_types(self) -> List[EdgeType]:
        return list(self._edges)

    def _collect(self, attr: str) -> Dict[str, Tensor]:
        return {k: s[attr] for k, s in self._nodes.items() if attr in s}

    @property
    def x_dict(self):
        return self._collect("x")

    @property
    def pos_dict(self):
        return self._collect("pos")

    @property
    def batch_dict(self):
        return self._collect("batch")

    @property
    def edge_index_dict(self) -> Dict[EdgeType, Tensor]:
        return {k: s["edge_index"] for k, s in self._edges.items() if "edge_index" in s}

    # -- movement -----------------------------------------------------------
    def to(self, device, non_blocking: bool = False) -> "HeteroBatch":
        out = HeteroBatch(self.num_graphs)
        mv = lambda v: v.to(device, non_blocking=non_blocking) if isinstance(v, Tensor) else v
        for k, s in self._nodes.items():
            out._nodes[k] = NodeStore({a: mv(v) for a, v in s.items()})
        for k, s in self._edges.items():
            out._edges[k] = EdgeStore({a: mv(v) for a, v in s.items()})
        return out

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def __repr__(self):
        n = ", ".join(f"{k}={s.num_nodes}" for k, s in self._nodes.items())
        e = ", ".join(f"{'-'.join(k)}={s.num_edges}" for k, s in self._edges.items())
        return f"HeteroBatch(graphs={self.num_graphs}; {n}; {e})"


def collate(tiles: Iterable[HeteroBatch]) -> HeteroBatch:
    """Concatenate tiles into one batch the way PyG's ``Batch.from_data_list``
    does for the fields of this contract."""
    tiles = list(tiles)
    out = HeteroBatch(num_graphs=len(tiles))
    offsets: Dict[str, List[int]] = {}
    for nt in tiles[0].node_types:
        sizes = [t[nt].num_nodes for t in tiles]
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        offsets[nt] = offs
        store = out[nt]
        for a in tiles[0][nt]:
            if a in ("batch", "num_nodes"):
                continue
            store[a] = torch.cat([t[nt][a] for t in tiles], 0)
        store["batch"] = torch.cat(
            [torch.full((s,), g, dtype=torch.long) for g, s in enumerate(sizes)])
    for et in tiles[0].edge_types:
        s, _, d = et
        eis = []
        for g, t in enumerate(tiles):
            ei = t[et].edge_index.long()
            shift = torch.tensor([[offsets[s][g]], [offsets[d][g]]], dtype=torch.long)
            eis.append(ei + shift)
        out[et]["edge_index"] = torch.cat(eis, 1)
    return out
