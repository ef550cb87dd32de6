"""Inference-only edge scoring with hipGraph replay (BASELINE.json config 5).

``LitISTEncoder.predict_step`` (reference ``src/segger/models/lightning_model.py:263-298``) launches ~150 small
kernels per tile batch; at prediction-tile sizes (<= 50 k nodes, ``data_module.py:155``) the step is
launch-bound.  :class:`GraphedPredictor` captures encoder forward + the fused cosine / arg-max / assignment
kernel into one HIP graph per *shape bucket* and replays it for every batch padded to that bucket:

* :func:`pad_batch` pads a batch to bucket sizes with isolated dummy nodes and dummy->dummy edges, so real
  nodes see exactly their own neighbourhoods (outputs of the real rows are unchanged);
* the CSR views are rebuilt eagerly per batch (a radix sort, outside the graph) into the captured buffers;
* masks (``predict_mask``) and the device->host copy stay outside the graph, as in the reference.

hipGraph capture works because every C-ABI entry point only enqueues on the caller's stream and never
allocates or synchronises (include/segger_amd.h conventions).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .graph import EdgeGraph, build_edge_graph
from .hetero import HeteroBatch, TX_BD, TX_NB_BD, TX_TX

_EDGE_TYPES = (TX_TX, TX_BD, TX_NB_BD)


def bucket_sizes(batch, granularity: float = 1.25, floor: int = 1024) -> Dict[str, int]:
    """Round every node / edge count up to the next power of ``granularity`` (>= floor + 1 for the dummy)."""
    def up(n: int) -> int:
        b = floor
        while b < n + 1:
            b = int(b * granularity) + 1
        return b
    sizes = {"tx": up(batch["tx"].num_nodes), "bd": up(batch["bd"].num_nodes)}
    for et in _EDGE_TYPES:
        sizes["__".join(et)] = up(int(batch[et].edge_index.shape[1]))
    return sizes


def pad_batch(batch, sizes: Dict[str, int]) -> HeteroBatch:
    """Pad to ``sizes``: dummy nodes copy node 0's attributes (positions included, so per-graph min/max
    are unchanged), get ``predict_mask`` False, and padding edges connect the LAST dummy tx to the last dummy
    tx / bd only (spread round-robin over the dummies)."""
    out = HeteroBatch(num_graphs=getattr(batch, "num_graphs", 1))
    n = {}
    for nt in ("tx", "bd"):
        cur = batch[nt].num_nodes
        tgt = sizes[nt]
        if tgt <= cur:
            raise ValueError(f"bucket for '{nt}' ({tgt}) must exceed the batch size ({cur})")
        n[nt] = tgt
        for a, v in batch[nt].items():
            if a == "num_nodes" or not isinstance(v, Tensor):
                continue
            fill = v[:1].expand(tgt - cur, *v.shape[1:])
            if a in ("predict_mask", "mask"):
                fill = torch.zeros_like(fill)
            out[nt][a] = torch.cat([v, fill], 0)
    for et in _EDGE_TYPES:
        s, _, d = et
        ei = batch[et].edge_index.long()
        tgt = sizes["__".join(et)]
        pad = tgt - ei.shape[1]
        if pad <= 0:
            raise ValueError(f"bucket for {et} ({tgt}) must exceed the edge count ({ei.shape[1]})")
        # spread the padding edges round-robin over the dummy nodes: one dummy hub with ~25 % of all edges
        # would serialise a whole kernel on a single row
        k = torch.arange(pad, device=ei.device, dtype=ei.dtype)
        n_ds, n_dd = n[s] - batch[s].num_nodes, n[d] - batch[d].num_nodes
        fill = torch.stack([batch[s].num_nodes + k % n_ds, batch[d].num_nodes + k % n_dd])
        out[et]["edge_index"] = torch.cat([ei, fill], 1)
    return out


class GraphedPredictor:
    """Replays ``model.predict_step`` for batches of ONE bucket shape through a captured HIP graph."""

    def __init__(self, model, sizes: Dict[str, int], bd_dim: int, min_similarity: Optional[float] = None,
                 max_graphs: int = 64):
        self.model, self.sizes, self.min_similarity = model, dict(sizes), min_similarity
        dev = next(model.parameters()).device
        nt, nb = sizes["tx"], sizes["bd"]
        self.inp = {
            "tx_x": torch.zeros(nt, dtype=torch.int32, device=dev),
            "tx_pos": torch.zeros(nt, 2, device=dev), "tx_batch": torch.zeros(nt, dtype=torch.long, device=dev),
            "bd_x": torch.zeros(nb, bd_dim, device=dev), "bd_pos": torch.zeros(nb, 2, device=dev),
            "bd_batch": torch.zeros(nb, dtype=torch.long, device=dev),
            "bd_index": torch.zeros(nb, dtype=torch.long, device=dev),
        }
        self.graphs: Dict[Tuple[str, str, str], EdgeGraph] = {}
        self._csr_buffers: Dict = {}
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out: Dict[str, Tensor] = {}
        # the per-graph min / max tables are sized for max_graphs, so one captured graph serves batches of any
        # number of tiles up to that (graphs without nodes are never read)
        self.num_graphs = int(max_graphs)

    # -- staging ------------------------------------------------------------------------------
    def _stage(self, pb) -> None:
        self.inp["tx_x"].copy_(pb["tx"]["x"]); self.inp["tx_pos"].copy_(pb["tx"]["pos"])
        self.inp["tx_batch"].copy_(pb["tx"]["batch"])
        self.inp["bd_x"].copy_(pb["bd"]["x"]); self.inp["bd_pos"].copy_(pb["bd"]["pos"])
        self.inp["bd_batch"].copy_(pb["bd"]["batch"]); self.inp["bd_index"].copy_(pb["bd"]["index"])
        n = {"tx": self.sizes["tx"], "bd": self.sizes["bd"]}
        for et in _EDGE_TYPES:
            s, _, d = et
            g = build_edge_graph(pb[et].edge_index, n[s], n[d], need_by_dst=et != TX_NB_BD,
                                 need_by_src=et == TX_NB_BD, validate=False)
            if et not in self.graphs:
                self.graphs[et] = g                      # first batch: these tensors become the captured buffers
            else:
                for side in ("by_dst", "by_src"):
                    cur, new = getattr(self.graphs[et], side), getattr(g, side)
                    if cur is not None:
                        cur.indptr.copy_(new.indptr); cur.col.copy_(new.col); cur.eid.copy_(new.eid)

    def _run(self) -> None:
        m = self.model.model
        z = m(
            {"tx": self.inp["tx_x"], "bd": self.inp["bd_x"]},
            {et: None for et in (TX_TX, TX_BD)},
            {"tx": self.inp["tx_pos"], "bd": self.inp["bd_pos"]},
            {"tx": self.inp["tx_batch"], "bd": self.inp["bd_batch"]},
            num_graphs=self.num_graphs, graphs={et: self.graphs[et] for et in (TX_TX, TX_BD)},
        )
        max_sim, _, seg, _ = ops.edge_cos_argmax(self.graphs[TX_NB_BD].by_src, z["tx"], z["bd"],
                                                 dst_index=self.inp["bd_index"], min_similarity=self.min_similarity)
        self.out = {"max_sim": max_sim, "seg_idx": seg}

    # -- public -------------------------------------------------------------------------------
    @torch.no_grad()
    def predict(self, batch):
        """-> (tx_index, seg_idx, max_sim, gene_id) on the CPU, exactly like ``predict_step``."""
        if self.model.training:
            raise RuntimeError("GraphedPredictor needs model.eval()")
        if getattr(batch, "num_graphs", 1) > self.num_graphs:
            raise ValueError(f"batch holds {batch.num_graphs} graphs, predictor was built for <= {self.num_graphs}")
        pb = pad_batch(batch, self.sizes)
        self._stage(pb)
        if self.graph is None:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._run()                              # warm-up on a side stream (lazy inits, allocator)
            torch.cuda.current_stream().wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._run()
        self.graph.replay()
        n = batch["tx"].num_nodes
        mask = batch["tx"]["predict_mask"]
        out = (batch["tx"]["index"][mask], self.out["seg_idx"][:n][mask], self.out["max_sim"][:n][mask],
               batch["tx"]["x"][mask])
        return tuple(t.cpu() for t in out)
