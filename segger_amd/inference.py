"""Inference-only edge scoring with hipGraph replay (BASELINE.json config 5).

``LitISTEncoder.predict_step`` (reference ``src/segger/models/lightning_model.py:263-298``) launches ~150 small
kernels per tile batch; at prediction-tile sizes (<= 50 k nodes, ``data_module.py:155``) the step is
launch-bound.  :class:`GraphedPredictor` captures encoder forward + the fused cosine / arg-max / assignment
kernel into one HIP graph per *shape bucket* and replays it for every batch padded to that bucket:

* a batch is padded to bucket sizes with isolated dummy nodes and dummy->dummy edges, so real nodes see exactly
  their own neighbourhoods (outputs of the real rows are unchanged);
* one ``segger_stage`` launch per batch writes node attributes and the three CSR views (copied from the batch's own
  views, padding by index arithmetic) into the captured buffers -- no per-batch sort or concatenation;
* masks (``predict_mask``) and the device->host copy stay outside the graph, as in the reference; ``predict_device``
  defers both to the caller so that a loop over batches never waits for the GPU.

hipGraph capture works because every C-ABI entry point only enqueues on the caller's stream and never
allocates or synchronises (include/segger_amd.h conventions).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .graph import EdgeCSR, EdgeGraph, batch_cache, build_edge_graph, edge_graph, padded_view_segments
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


class GraphedPredictor:
    """Replays ``model.predict_step`` for batches of ONE bucket shape through a captured HIP graph.

    A batch is written into the static buffers by ONE launch (``ops.stage`` -> ``segger_stage``): node attributes are
    copied and padded with copies of node 0; the three CSR views (tx-neighbors-tx and tx-belongs-bd by destination,
    tx-neighbors-bd by source) are copied from the batch's own views -- slices of the slide-level sort when the batch
    comes from ``tiles.TilePartition`` -- and their padding edges are index arithmetic on the dummy rows
    (``graph.padded_view_segments``): no per-batch sort, no concatenation."""

    def __init__(self, model, sizes: Dict[str, int], bd_dim: int, min_similarity: Optional[float] = None,
                 max_graphs: int = 64):
        self.model, self.sizes, self.min_similarity = model, dict(sizes), min_similarity
        dev = next(model.parameters()).device
        self.dev = dev
        nt, nb = sizes["tx"], sizes["bd"]
        z = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device=dev)
        self.inp = {
            "tx_x": z(nt, dtype=torch.int32), "tx_pos": z(nt, 2), "tx_batch": z(nt, dtype=torch.long),
            "bd_x": z(nb, bd_dim), "bd_pos": z(nb, 2), "bd_batch": z(nb, dtype=torch.long),
            "bd_index": z(nb, dtype=torch.long),
        }

        def csr(n_rows, n_cols, n_edges):
            return EdgeCSR(z(n_rows + 1, dtype=torch.long), z(n_edges, dtype=torch.int32),
                           z(n_edges, dtype=torch.int32), n_rows, n_cols)
        e = {et: sizes["__".join(et)] for et in _EDGE_TYPES}
        self.graphs: Dict[Tuple[str, str, str], EdgeGraph] = {
            TX_TX: EdgeGraph(csr(nt, nt, e[TX_TX]), None, nt, nt, e[TX_TX]),
            TX_BD: EdgeGraph(csr(nb, nt, e[TX_BD]), None, nt, nb, e[TX_BD]),
            TX_NB_BD: EdgeGraph(None, csr(nt, nb, e[TX_NB_BD]), nt, nb, e[TX_NB_BD]),
        }
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out: Dict[str, Tensor] = {}
        # the per-graph min / max tables are sized for max_graphs, so one captured graph serves batches of any
        # number of tiles up to that (graphs without nodes are never read)
        self.num_graphs = int(max_graphs)

    def fits(self, batch) -> bool:
        s = self.sizes
        return (batch["tx"].num_nodes < s["tx"] and batch["bd"].num_nodes < s["bd"]
                and all(int(batch[et].edge_index.shape[1]) <= s["__".join(et)] for et in _EDGE_TYPES)
                and int(getattr(batch, "num_graphs", 1)) <= self.num_graphs)

    def waste(self, batch) -> float:
        s = self.sizes
        return max(s["__".join(TX_TX)] / max(int(batch[TX_TX].edge_index.shape[1]), 1),
                   s["tx"] / max(batch["tx"].num_nodes, 1))

    # -- staging ------------------------------------------------------------------------------
    @torch.no_grad()
    def _stage(self, batch) -> None:
        if not self.fits(batch):
            raise ValueError("batch does not fit this predictor's bucket")
        tx, bd = batch["tx"], batch["bd"]
        n_tx, n_bd, nt, nb = tx.num_nodes, bd.num_nodes, self.sizes["tx"], self.sizes["bd"]
        cache = batch_cache(batch)
        # the same cache keys the eager predict_step uses: a batch scored both ways is sorted (or sliced) once
        g_tt = edge_graph(cache, TX_TX, batch[TX_TX].edge_index, n_tx, n_tx, need_by_src=False, validate=False)
        g_tb = edge_graph(cache, TX_BD, batch[TX_BD].edge_index, n_tx, n_bd, need_by_src=False, validate=False)
        g_nb = edge_graph(cache, TX_NB_BD, batch[TX_NB_BD].edge_index, n_tx, n_bd, need_by_dst=False, validate=False)
        I = self.inp
        segs = [(I["tx_x"], tx["x"], "const", 0, 0, 0), (I["tx_pos"], tx["pos"], "tile", 2, 0, 0),
                (I["tx_batch"], tx["batch"], "tile", 1, 0, 0),
                (I["bd_x"], bd["x"], "tile", max(int(bd["x"][0].numel()), 1), 0, 0),
                (I["bd_pos"], bd["pos"], "tile", 2, 0, 0), (I["bd_batch"], bd["batch"], "tile", 1, 0, 0),
                (I["bd_index"], bd["index"], "tile", 1, 0, 0)]
        segs += padded_view_segments(self.graphs[TX_TX].by_dst, g_tt.by_dst, n_tx, None)
        segs += padded_view_segments(self.graphs[TX_BD].by_dst, g_tb.by_dst, n_bd, ("mod", n_tx, nt - n_tx))
        segs += padded_view_segments(self.graphs[TX_NB_BD].by_src, g_nb.by_src, n_tx, ("mod", n_bd, nb - n_bd))
        ops.stage(segs, self.dev)

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
    def predict_device(self, batch):
        """-> (tx_index, seg_idx, max_sim, gene_id, predict_mask) on the DEVICE, all [n_tx] and NOT yet filtered by the
        mask: nothing here waits for the GPU, so a loop over batches queues ahead of it; apply the mask (one
        compaction, one sync) after the loop."""
        if self.model.training:
            raise RuntimeError("GraphedPredictor needs model.eval()")
        self._stage(batch)
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
        return (batch["tx"]["index"], self.out["seg_idx"][:n].clone(), self.out["max_sim"][:n].clone(),
                batch["tx"]["x"], batch["tx"]["predict_mask"])

    @torch.no_grad()
    def predict(self, batch):
        """-> (tx_index, seg_idx, max_sim, gene_id) on the CPU, exactly like ``predict_step``."""
        index, seg, sim, gene, mask = self.predict_device(batch)
        return tuple(t[mask].cpu() for t in (index, seg, sim, gene))


class GraphedPredictorPool:
    """``pool.predict(batch)`` / ``pool.predict_device(batch)``: the tightest captured bucket the batch fits; a new
    one is captured when none fits or the tightest would pad the transcript side by more than two granules."""

    def __init__(self, model, bd_dim: int, min_similarity: Optional[float] = None, granularity: float = 1.08,
                 max_buckets: int = 32, max_graphs: int = 64):
        self.model, self.bd_dim, self.min_similarity = model, bd_dim, min_similarity
        self.granularity, self.max_buckets, self.max_graphs = granularity, max_buckets, max_graphs
        self.buckets: list = []

    def _pick(self, batch) -> GraphedPredictor:
        fit = [b for b in self.buckets if b.fits(batch)]
        best = min(fit, key=lambda b: b.waste(batch)) if fit else None
        full = len(self.buckets) >= self.max_buckets
        if best is None or (best.waste(batch) > self.granularity ** 2 and not full):
            if full:
                raise RuntimeError(f"no captured bucket holds this batch and {self.max_buckets} buckets exist")
            sizes = bucket_sizes(batch, self.granularity, floor=256)
            for k in sizes:                               # the boundary side is cheap: one granule of headroom more
                if k != "tx" and k != "__".join(TX_TX):
                    sizes[k] = int(sizes[k] * self.granularity) + 1
            best = GraphedPredictor(self.model, sizes, self.bd_dim, self.min_similarity,
                                    max(self.max_graphs, int(getattr(batch, "num_graphs", 1))))
            self.buckets.append(best)
        return best

    def predict(self, batch):
        return self._pick(batch).predict(batch)

    def predict_device(self, batch):
        return self._pick(batch).predict_device(batch)
