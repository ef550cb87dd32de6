"""One whole training step -- encoder forward, the three losses, backward, Adam -- as a single hipGraph replay.

At segger's default batch budget (``edges_per_batch = 1 000 000``, reference ``data/data_module.py:158``) a step is
~2.8 ms of device work behind ~280 launches; eager, the host needs ~4 ms to queue them.  Everything in the step is free
of host synchronisation and of data-dependent shapes (masks are weights, not compactions; samplers are kernels; the
dropout / sampling streams read DEVICE counters), so the step is captured ONCE per shape bucket and replayed for every
batch padded into the bucket's static buffers:

* nodes are padded with copies of node 0 (positions included: per-graph min / max unchanged), ``mask`` False;
* padding edges touch dummy nodes only and are laid out so that the CSR views need no sort: tx-neighbors-tx pads with
  self-loops on the dummies (identical by-destination and by-source views); tx-belongs-bd pads run from the dummy
  transcripts (round robin) to the dummy boundaries.  No loss term reads a dummy row, so every gradient that reaches a
  dummy is exactly zero: the one-pass tx-belongs-bd backward, which WRITES one source row per edge, may see a dummy
  source twice -- both writes store zero -- and the weight gradients, which sum over all rows, are untouched;
* the real part of every view is copied from the batch's own CSR (sliced from the slide-level sort by
  ``tiles.TilePartition``), the padding part is index arithmetic; real edges keep their COO positions, so the
  attention-dropout masks of real edges are the ones the eager step would draw;
* loss means are rescaled from padded to real counts on the device; padded triplets carry ``-1`` and are skipped by the
  kernels; negatives are drawn in ``[0, n_bd_real)``.

What it computes is ``LitISTEncoder.training_step`` + ``optimizer.step()`` (reference lightning_model.py:151-231 with
Lightning's automatic optimisation), for the default ``sg_loss_type='triplet'``.  The optimizer must be capturable
(``LitISTEncoder.configure_optimizers(capturable=True)``); ``ops.invalidate_weight_cache()`` is called after every
replay because the parameters change without Python noticing.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import ops
from .graph import EdgeCSR, EdgeGraph, batch_cache, edge_graph
from .hetero import TX_BD, TX_TX

_NODE_ATTRS = ("x", "pos", "batch", "mask", "cluster")


def step_bucket(batch, granularity: float = 1.06, floor: int = 256) -> Dict[str, int]:
    """Bucket sizes for one batch: counts rounded up to the next power of ``granularity`` (> the count: every node type
    keeps at least one dummy).  The boundary-side counts are a small part of the work and get one granule of headroom
    more, so that batches which agree on the transcript side share a bucket."""
    def up(n: int, extra: int = 0) -> int:
        b = floor
        while b <= n:
            b = int(b * granularity) + 1
        for _ in range(extra):
            b = int(b * granularity) + 1
        return b
    e_tt, e_tb = int(batch[TX_TX].edge_index.shape[1]), int(batch[TX_BD].edge_index.shape[1])
    return {"tx": up(batch["tx"].num_nodes), "bd": up(batch["bd"].num_nodes, 1), "e_tt": up(e_tt), "e_tb": up(e_tb, 1),
            "graphs": up(int(getattr(batch, "num_graphs", 1)), 1)}


class GraphedTrainStep:
    """The captured step of ONE shape bucket (see :class:`GraphedTrainer` for the per-batch dispatch)."""

    def __init__(self, lit_model, optimizer, sizes: Dict[str, int], template):
        if lit_model._sg_loss_type != "triplet":
            raise NotImplementedError("the graphed step covers the (default) triplet segmentation loss")
        self.lit, self.opt, self.sizes = lit_model, optimizer, dict(sizes)
        if lit_model.loss_tx is None or lit_model.loss_bd is None:
            raise RuntimeError("call setup() (or set_similarities) before training")
        dev = next(lit_model.parameters()).device
        self.dev = dev
        for sel in (lit_model.loss_tx.selector, lit_model.loss_bd.selector):      # no host -> device copy in a capture
            sel.similarity, sel.dissimilarity = sel.similarity.to(dev), sel.dissimilarity.to(dev)
        nt, nb, ett, etb = sizes["tx"], sizes["bd"], sizes["e_tt"], sizes["e_tb"]
        z = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device=dev)
        self.nodes = {k: {a: z(n, *template[k][a].shape[1:], dtype=template[k][a].dtype) for a in _NODE_ATTRS}
                      for k, n in (("tx", nt), ("bd", nb))}

        def csr(n_rows, n_cols, n_edges):
            return EdgeCSR(z(n_rows + 1, dtype=torch.long), z(n_edges, dtype=torch.int32),
                           z(n_edges, dtype=torch.int32), n_rows, n_cols)
        self.g_tt = EdgeGraph(csr(nt, nt, ett), csr(nt, nt, ett), nt, nt, ett)
        self.g_tb = EdgeGraph(csr(nb, nt, etb), None, nt, nb, etb, None, True)        # one-pass backward (see above)
        self.ei_tb = z(2, etb, dtype=torch.long)
        self.counts = z(3, dtype=torch.long)                  # real n_tx, n_bd, e_tb of the staged batch
        self.weights = z(3)                                   # scheduled loss weights (tx, bd, sg)
        self._host = torch.zeros(6, dtype=torch.float64).pin_memory()
        self.out: Optional[Tensor] = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._training: Optional[bool] = None
        self._iota = torch.arange(max(nt, nb, ett, etb), device=dev)
        self.draws = None                                     # tests: fixed (tx pos/neg, bd pos/neg/dp/dn, dst_neg)

    def fits(self, batch) -> bool:
        s = self.sizes
        e_tb = int(batch[TX_BD].edge_index.shape[1])
        return (batch["bd"].num_nodes < s["bd"] and int(batch[TX_TX].edge_index.shape[1]) <= s["e_tt"]
                and e_tb <= s["e_tb"] and batch["tx"].num_nodes < s["tx"]
                and int(getattr(batch, "num_graphs", 1)) <= s["graphs"])

    def waste(self, batch) -> float:
        """Padded / real size of the transcript side (what a step costs)."""
        s = self.sizes
        return max(s["e_tt"] / max(int(batch[TX_TX].edge_index.shape[1]), 1), s["tx"] / max(batch["tx"].num_nodes, 1))

    # ------------------------------------------------------------------------------------------- staging
    @torch.no_grad()
    def _stage_view(self, dst: EdgeCSR, src: EdgeCSR, n_real: int, pad_cols: Optional[Tensor]) -> None:
        """Real rows / slots from ``src``; padding edges spread evenly over the dummy rows n_real .. n_rows-1 with
        col = the row itself (``pad_cols`` None) or the given dummy columns."""
        e, e_pad = src.n_edges, dst.n_edges
        dst.indptr[: n_real + 1].copy_(src.indptr)
        dst.col[:e].copy_(src.col)
        dst.eid[:e].copy_(src.eid)
        pad, n_dummy = e_pad - e, dst.n_rows - n_real
        q = max(-(-pad // n_dummy), 1)                        # padding edges per dummy row
        dst.indptr[n_real + 1:] = e + torch.clamp((self._iota[:n_dummy] + 1) * q, max=pad)
        k = self._iota[:pad]
        dst.col[e:] = ((n_real + k // q) if pad_cols is None else pad_cols).to(torch.int32)
        dst.eid[e:] = (e + k).to(torch.int32)

    @torch.no_grad()
    def stage(self, batch) -> None:
        if not self.fits(batch):
            raise ValueError("batch does not fit this bucket")
        n_tx, n_bd = batch["tx"].num_nodes, batch["bd"].num_nodes
        nb = self.sizes["bd"]
        for k, n in (("tx", n_tx), ("bd", n_bd)):
            for a, buf in self.nodes[k].items():
                v = batch[k][a]
                buf[:n].copy_(v)
                if a == "mask":
                    buf[n:] = False
                elif a == "cluster":
                    buf[n:] = 0
                else:
                    buf[n:] = v[0]
        cache = batch_cache(batch)
        g_tt = edge_graph(cache, TX_TX, batch[TX_TX].edge_index, n_tx, n_tx, need_by_src=True, validate="deferred")
        g_tb = edge_graph(cache, TX_BD, batch[TX_BD].edge_index, n_tx, n_bd, need_by_src="lazy", validate="deferred")
        if not g_tb.src_unique():
            raise NotImplementedError("a transcript with two tx-belongs-bd edges (heterodata.py:147 assigns one)")
        e_tb = g_tb.n_edges
        pad = self.sizes["e_tb"] - e_tb
        k = self._iota[:pad]
        pad_src = n_tx + k % (self.sizes["tx"] - n_tx)        # dummy transcripts, round robin
        self._stage_view(self.g_tt.by_dst, g_tt.by_dst, n_tx, None)
        self._stage_view(self.g_tt.by_src, g_tt.require_by_src(), n_tx, None)
        self._stage_view(self.g_tb.by_dst, g_tb.by_dst, n_bd, pad_src)
        # the COO list of tx-belongs-bd for the segmentation loss, padding edges in the order of their CSR slots
        q = max(-(-pad // (nb - n_bd)), 1)
        self.ei_tb[:, :e_tb].copy_(batch[TX_BD].edge_index)
        self.ei_tb[0, e_tb:] = pad_src
        self.ei_tb[1, e_tb:] = n_bd + k // q
        w = self.lit._scheduled_weights(self.lit._w_start, self.lit._w_end)
        self._host[:3] = torch.tensor([n_tx, n_bd, e_tb], dtype=torch.float64)
        self._host[3:] = w.double()
        self.counts.copy_(self._host[:3], non_blocking=True)
        self.weights.copy_(self._host[3:], non_blocking=True)

    # ------------------------------------------------------------------------------------------- the step
    def _run(self) -> None:
        lit, enc, s = self.lit, self.lit.model, self.sizes
        nt, etb = s["tx"], s["e_tb"]
        tx, bd = self.nodes["tx"], self.nodes["bd"]
        self.opt.zero_grad(set_to_none=True)
        z = enc({"tx": tx["x"], "bd": bd["x"]}, {TX_TX: None, TX_BD: None}, {"tx": tx["pos"], "bd": bd["pos"]},
                {"tx": tx["batch"], "bd": bd["batch"]}, num_graphs=s["graphs"],
                graphs={TX_TX: self.g_tt, TX_BD: self.g_tb})
        step = enc._step_dev                                  # advanced by the forward: a fresh stream per replay
        n_bd, e_real = self.counts[1], self.counts[2]
        fixed = self.draws
        # loss_tx / loss_bd: the masked forms of triplet_loss.py, the selector's index rebuilt from the staged labels
        if fixed is None:
            sel = lit.loss_tx.selector
            pos, neg, _, _ = sel.sample_triplets(tx["cluster"], index=sel.build_index(tx["cluster"], tx["mask"]),
                                                 device_seed=(0x7478, step))
        else:
            pos, neg = fixed["tx"]
        l_tx = ops.triplet_edge_loss(z["tx"], None, self._iota[:nt], pos, neg, lit.loss_tx.margin, eps=lit.loss_tx.eps)
        l_tx = l_tx * (float(nt) / tx["mask"].sum().clamp(min=1).float())
        bmask = bd["mask"] & (bd["cluster"] >= 0)
        if fixed is None:
            sel = lit.loss_bd.selector
            pos, neg, dp, dn = sel.sample_triplets(bd["cluster"], index=sel.build_index(bd["cluster"], bmask),
                                                   device_seed=(0x6264, step))
        else:
            pos, neg, dp, dn = fixed["bd"]
        l_bd = ops.metric_loss(z["bd"], pos, neg, dp, dn, bmask.float() / bmask.sum().clamp(min=1).float())
        # segmentation loss over the real tx-belongs-bd edges (lightning_model.py:167-189): negatives in
        # [0, n_bd_real), padded triplets = -1 (skipped), mean over the real edges
        src, dst = self.ei_tb[0], self.ei_tb[1]
        valid = self._iota[:etb] < e_real
        minus = torch.full_like(dst, -1)
        if fixed is None:
            span = (n_bd - 1).clamp(min=1)
            shift = 1 + (torch.rand(etb, device=self.dev) * span.float()).long().clamp(max=span - 1)
            dst_neg = (dst + shift) % n_bd.clamp(min=1)
        else:
            dst_neg = fixed["dst_neg"]
        l_sg = ops.triplet_edge_loss(z["tx"], z["bd"], src, torch.where(valid, dst, minus),
                                     torch.where(valid, dst_neg, minus), lit._sg_margin, eps=1e-6,
                                     pos_groups=self.g_tb.by_dst)
        l_sg = l_sg * (float(etb) / e_real.clamp(min=1).float())
        l_sg = torch.where(n_bd > 1, l_sg, torch.zeros_like(l_sg))              # :173-175
        loss = self.weights[0] * l_tx + self.weights[1] * l_bd + self.weights[2] * l_sg
        loss.backward()
        self.opt.step()
        self.out = torch.stack([l_tx.detach().float(), l_bd.detach().float(), l_sg.detach().float(),
                                loss.detach().float()])

    @torch.no_grad()
    def _snapshot(self):
        params = [p for g in self.opt.param_groups for p in g["params"]]
        state = {id(p): {k: v.clone() for k, v in self.opt.state.get(p, {}).items() if isinstance(v, Tensor)}
                 for p in params}
        return params, [p.detach().clone() for p in params], state, self.lit.model._step_dev.clone()

    @torch.no_grad()
    def _restore(self, keep) -> None:
        """Undo the warm-up step IN PLACE (the capture must see the tensors the optimizer keeps using): parameters
        and the dropout counter from the snapshot; optimizer state from the snapshot, or zeroed where the warm-up
        created it (Adam's own initial state)."""
        params, values, state, step_dev = keep
        torch._foreach_copy_(params, values)
        self.lit.model._step_dev.copy_(step_dev)
        for p in params:
            for k, v in self.opt.state.get(p, {}).items():
                if isinstance(v, Tensor):
                    old = state[id(p)].get(k)
                    v.zero_() if old is None else v.copy_(old)

    def step(self, batch, capture: bool = True) -> Tensor:
        """-> [loss_tx, loss_bd, loss_sg, loss] (a static device tensor, overwritten by the next step)."""
        lit = self.lit
        self.stage(batch)
        if not capture:
            self._run()
        elif self.graph is None or self._training != lit.model.training:
            lit.model._materialize_bd(self.nodes["bd"]["x"].shape[1], self.dev)
            keep = self._snapshot()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                     # warm-up: lazy inits, allocator, optimizer state
                self._run()
            torch.cuda.current_stream().wait_stream(side)
            self._restore(keep)                               # ... which must not count as a training step
            ops.invalidate_weight_cache()                     # the captured step starts with the weight refresh
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._run()
            self._training = lit.model.training
            self.graph.replay()                               # (capturing runs nothing)
        else:
            self.graph.replay()
        ops.invalidate_weight_cache()                         # parameters changed behind Python's back
        return self.out


class GraphedTrainer:
    """``trainer.step(batch)``: stage the batch into the tightest bucket it fits and replay.  A new bucket is captured
    when none fits or the tightest one would pad the transcript side by more than two granules; with ``max_buckets``
    captured, the tightest fit is used whatever it wastes, and a batch no bucket holds is an error."""

    def __init__(self, lit_model, optimizer, granularity: float = 1.06, max_buckets: int = 24):
        self.lit, self.opt, self.granularity, self.max_buckets = lit_model, optimizer, granularity, max_buckets
        self.buckets: List[GraphedTrainStep] = []
        self.n_captures = 0

    def step(self, batch) -> Tensor:
        fit = [b for b in self.buckets if b.fits(batch)]
        best = min(fit, key=lambda b: b.waste(batch)) if fit else None
        full = len(self.buckets) >= self.max_buckets
        if best is None or (best.waste(batch) > self.granularity ** 2 and not full):
            if full:
                raise RuntimeError(f"no captured bucket holds this batch and {self.max_buckets} buckets exist: "
                                   f"raise `granularity` or `max_buckets`")
            best = GraphedTrainStep(self.lit, self.opt, step_bucket(batch, self.granularity), batch)
            self.buckets.append(best)
            self.n_captures += 1
        return best.step(batch)
