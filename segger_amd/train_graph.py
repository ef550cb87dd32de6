"""hipGraph-captured encoder forward + backward for training on tile batches.

At segger's default batch budget (``edges_per_batch = 1 000 000``, reference ``data/data_module.py:158``) one
training step is ~550 kernel launches for ~1.7 ms of device work: the step is launch-bound (5.9 ms measured).
:class:`GraphedEncoder` captures the encoder's forward and backward (≈ 80 % of the launches) as two HIP graphs
per *shape bucket* with ``torch.cuda.make_graphed_callables`` and replays them for every batch padded to the
bucket; the three losses and the optimizer stay eager on the real (unpadded) batch.

What makes the capture valid: every C-ABI entry point only enqueues on the current stream (no allocation,
no synchronisation); the attention-dropout stream advances through a DEVICE counter (``ISTEncoder._step_dev``,
``seed_dev`` in ``segger_gatv2_fwd/bwd``), so every replay draws a fresh mask and the backward replay sees the
counter value of its own forward; the CSR views are rebuilt eagerly per batch into the captured buffers.

Constraint inherited from PyTorch: when the first call of a bucket captures the backward, no autograd graph
from an earlier eager step may still be alive (e.g. a retained loss tensor): its AccumulateGrad nodes force a
cross-stream sync inside the capture, which aborts it.  Keep only detached values between steps.

Superseded for training loops by ``segger_amd.train_step_graph`` (round 2): the WHOLE step -- losses and Adam included --
as one graph, one-launch staging instead of a per-batch re-sort, and a captured forward that runs on parameter aliases
and is therefore immune to the constraint above.  This module remains for callers that want eager losses on the real
batch around a replayed encoder.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import ops
from .graph import EdgeGraph, build_edge_graph
from .hetero import TX_BD, TX_TX
from .inference import bucket_sizes, pad_batch  # noqa: F401  (re-exported for callers)


class _StaticEncoder(torch.nn.Module):
    def __init__(self, encoder, graphs: Dict, max_graphs: int):
        super().__init__()
        self.encoder = encoder
        self.graphs = graphs
        self.max_graphs = max_graphs

    def forward(self, tx_x, tx_pos, tx_batch, bd_x, bd_pos, bd_batch):
        z = self.encoder(
            {"tx": tx_x, "bd": bd_x}, {TX_TX: None, TX_BD: None},
            {"tx": tx_pos, "bd": bd_pos}, {"tx": tx_batch, "bd": bd_batch},
            num_graphs=self.max_graphs, graphs=self.graphs)
        return z["tx"], z["bd"]


class GraphedEncoder:
    """``z = graphed(batch)`` == ``model.forward(batch)`` (same values for the real rows, autograd-connected to
    the parameters), executed as graph replays.  One instance serves one bucket (``sizes``)."""

    def __init__(self, lit_model, sizes: Dict[str, int], bd_dim: int, max_graphs: int = 64):
        self.lit = lit_model
        self.sizes = dict(sizes)
        self.max_graphs = int(max_graphs)
        dev = next(lit_model.parameters()).device
        nt, nb = sizes["tx"], sizes["bd"]
        self.inp = (
            torch.zeros(nt, dtype=torch.int32, device=dev), torch.zeros(nt, 2, device=dev),
            torch.zeros(nt, dtype=torch.long, device=dev),
            torch.zeros(nb, bd_dim, device=dev), torch.zeros(nb, 2, device=dev),
            torch.zeros(nb, dtype=torch.long, device=dev),
        )
        self.graphs: Dict = {}
        self.callable = None
        self._training_at_capture: Optional[bool] = None

    def _stage(self, pb) -> None:
        src = (pb["tx"]["x"], pb["tx"]["pos"], pb["tx"]["batch"], pb["bd"]["x"], pb["bd"]["pos"], pb["bd"]["batch"])
        for dst, s in zip(self.inp, src):
            dst.copy_(s)
        n = {"tx": self.sizes["tx"], "bd": self.sizes["bd"]}
        for et in (TX_TX, TX_BD):
            s, _, d = et
            g = build_edge_graph(pb[et].edge_index, n[s], n[d], validate=False)
            if et not in self.graphs:
                self.graphs[et] = g
            else:
                for side in ("by_dst", "by_src"):
                    cur, new = getattr(self.graphs[et], side), getattr(g, side)
                    cur.indptr.copy_(new.indptr); cur.col.copy_(new.col); cur.eid.copy_(new.eid)
        # rows grouped by gene id for the embedding-table gradient
        emb = self.lit.model.lin_first["tx"]
        bg = ops.rows_by_id(self.inp[0], emb.weight.shape[0])
        if "tx_by_gene" not in self.graphs:
            self.graphs["tx_by_gene"] = bg
        else:
            cur = self.graphs["tx_by_gene"]
            cur.indptr.copy_(bg.indptr); cur.col.copy_(bg.col)

    def __call__(self, batch) -> Dict[str, Tensor]:
        if getattr(batch, "num_graphs", 1) > self.max_graphs:
            raise ValueError(f"batch holds {batch.num_graphs} graphs, bucket was built for <= {self.max_graphs}")
        pb = pad_batch(batch, self.sizes)
        self._stage(pb)
        enc = self.lit.model
        if self.callable is None or self._training_at_capture != enc.training:
            enc._materialize_bd(self.inp[3].shape[1], self.inp[3].device)
            mod = _StaticEncoder(enc, self.graphs, self.max_graphs)
            # allow_unused_input: the integer / position inputs take no gradient
            self.callable = torch.cuda.make_graphed_callables(mod, self.inp, allow_unused_input=True)
            self._training_at_capture = enc.training
        z_tx, z_bd = self.callable(*self.inp)
        return {"tx": z_tx[: batch["tx"].num_nodes], "bd": z_bd[: batch["bd"].num_nodes]}
