"""One whole training step -- encoder forward, the three losses, backward, Adam -- as a single hipGraph replay.

At segger's default batch budget (``edges_per_batch = 1 000 000``, reference ``data/data_module.py:158``) a step is
under 2 ms of device work; eager, the host needs ~4 ms to queue its ~280 launches (50M-transcript FOV, one MI355X:
4.2 ms per step eager, 1.8 ms replayed).  Everything in the step is free
of host synchronisation and of data-dependent shapes (masks are weights, not compactions; samplers are kernels; the
dropout / sampling streams read DEVICE counters), so the step is captured ONCE per shape bucket and replayed for every
batch padded into the bucket's static buffers:

* nodes are padded with copies of node 0 (positions included: per-graph min / max unchanged; dummy transcripts carry
  the last gene id, so the rows-by-gene grouping of the embedding gradient only grows at its end), masked out of the
  losses (the samplers put them in the cluster nobody draws from);
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
(``LitISTEncoder.configure_optimizers(capturable=True)``); ``ops.invalidate_weights(params)`` is called after every
replay because the parameters change without Python noticing.  The capture runs inside ``ops.pack_scope(aliases)``: the
weight refresh baked into the graph touches only the compute-dtype copies of THIS step's parameter aliases, which the
step keeps alive (``_pack_refs``), never another model's.

Data parallelism (``GraphedTrainer(..., grad_bucket=dp.FlatGradBucket(...))``): the step is captured as TWO graphs --
forward + backward + the packing of the gradients into the bucket's persistent flat buffer, and Adam reading that
buffer's slices -- with the gradient exchange between them run eagerly: one all-reduce + one divide per step on every
rank, whatever a rank is capturing; a rank that ran out of batches zeroes the buffer instead of replaying the first
graph.  Covered by a two-rank gloo test on one GPU; not yet run over RCCL on a multi-GPU box.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import ops
from .graph import EdgeCSR, EdgeGraph, batch_cache, edge_graph, padded_view_segments
from .hetero import TX_BD, TX_TX


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


USE_ADAM_KERNEL = True       # (False: torch's fused Adam inside the captured step -- A/B switch)
MERGED_DRAWS = True          # (False: dropout planes, both samplers and the negatives as launches of their own, the dropout
                             #  counter advanced at the head of the step, Adam's step counters by a launch of their own)
DEFER_LOSS_FINISH = True     # (False: the one-launch loss head's finishing launch right behind its forward)
STEP_INC = 256               # what a training forward advances the encoder's dropout counter by (ist_encoder.py)


class GraphedTrainStep:
    """The captured step of ONE shape bucket (see :class:`GraphedTrainer` for the per-batch dispatch).  A batch is
    written into the bucket's static buffers by ONE launch (``ops.stage`` -> ``segger_stage``): node features, the three
    CSR views, the rows-by-gene grouping of the embedding gradient, the segmentation triplets and the loss samplers'
    indices are all "copy the batch's own (cached) arrays, fill the padding by a formula" segments."""

    def __init__(self, lit_model, optimizer, sizes: Dict[str, int], template, grad_sync=None, grad_bucket=None):
        if grad_bucket is None and hasattr(getattr(grad_sync, "__self__", None), "pack"):
            grad_bucket, grad_sync = grad_sync.__self__, None     # bucket.all_reduce_mean given as the callable
        self.grad_sync, self.grad_bucket = grad_sync, grad_bucket
        self.split = grad_sync is not None or grad_bucket is not None
        if grad_bucket is not None:
            grad_bucket._ensure()                             # the flat buffer must exist before anything is captured
        if lit_model.loss_tx is None or lit_model.loss_bd is None:
            raise RuntimeError("call setup() (or set_similarities) before training")
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("the optimizer must keep its step counters on the device: "
                             "LitISTEncoder.configure_optimizers(capturable=True)")
        self.lit, self.opt, self.sizes = lit_model, optimizer, dict(sizes)
        dev = next(lit_model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs the model on a GPU (hipGraph capture)")
        self.dev = dev
        for sel in (lit_model.loss_tx.selector, lit_model.loss_bd.selector):      # no host -> device copy in a capture
            sel.similarity, sel.dissimilarity = sel.similarity.to(dev), sel.dissimilarity.to(dev)
        nt, nb, ett, etb = sizes["tx"], sizes["bd"], sizes["e_tt"], sizes["e_tb"]
        z = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device=dev)
        i32, i64 = torch.int32, torch.int64
        self.n_genes = int(lit_model.model.lin_first["tx"].weight.shape[0])
        # positions / graph ids of both node types back to back: the encoder runs its positional embedder once over
        # the concatenation (boundary graph ids offset by the bucket's graph count), staged here without a launch
        self.pos_all, self.batch_all = z(nt + nb, 2), z(nt + nb, dtype=i64)
        # per-graph min / max of the positions (both node types: 2 * graphs rows): re-armed with +inf / -inf by the staging
        # launch of every step, so that the captured forward needs only the accumulating kernel (ops.segment_minmax)
        self.minmax = (z(2 * sizes["graphs"], 2), z(2 * sizes["graphs"], 2))
        self.nodes = {"tx": {"x": z(nt, dtype=i32), "pos": self.pos_all[:nt], "batch": self.batch_all[:nt]},
                      # boundary features staged in the compute dtype (the cast of ist_encoder.py's `x_dict['bd'].to(dt)` is
                      # made once per tile set and kept with it, not once per step)
                      "bd": {"x": z(nb, *template["bd"]["x"].shape[1:], dtype=self._bd_dtype(lit_model, template)),
                             "pos": self.pos_all[nt:], "batch": z(nb, dtype=i64)}}

        def csr(n_rows, n_cols, n_edges):
            return EdgeCSR(z(n_rows + 1, dtype=i64), z(n_edges, dtype=i32), z(n_edges, dtype=i32), n_rows, n_cols)
        self.g_tt = EdgeGraph(csr(nt, nt, ett), csr(nt, nt, ett), nt, nt, ett)
        self.g_tb = EdgeGraph(csr(nb, nt, etb), None, nt, nb, etb, None, True)        # one-pass backward (see above)
        by_gene_col = z(nt, dtype=i32)
        self.by_gene = EdgeCSR(z(self.n_genes + 1, dtype=i64), by_gene_col, by_gene_col, self.n_genes, nt)
        self.sg_src, self.sg_pos = z(etb, dtype=i64), z(etb, dtype=i64)
        # one-launch loss head: the segmentation triplet anchored at each transcript row (-1: none) and the rows'
        # contribution chains (heads + counts), zero-filled by the staging launch of every step
        self.sg_of_tx, self.tx_state = z(nt, dtype=i32), z(2 * nt + 4, dtype=i32)

        def sampler_index(selector, n):
            k = int(selector.similarity.shape[0])
            return dict(lab=z(n, dtype=i64), members=z(n, dtype=i64), counts=z(k + 1, dtype=i64),
                        offsets=z(k + 1, dtype=i64), cdf_pos_t=z(k, k), cdf_neg_t=z(k, k),
                        dists=(1.0 - selector.similarity).contiguous(), n_clusters=k)
        self.ix_tx = sampler_index(lit_model.loss_tx.selector, nt)
        self.ix_bd = sampler_index(lit_model.loss_bd.selector, nb)
        self.bd_weight = z(nb)
        self.n_bd = z(1, dtype=i64)                           # real boundary count of the staged batch
        # padded -> masked means (padded rows / masked tx count, 1, e_tb_pad / e_tb_real or 0 for <= 1 boundary), loss weights
        self.scal = z(6)
        self.head_a = self.scal[0:3]
        # Adam's (lr, beta1, beta2, eps) as the kernel reads them at replay time (segger_adam_step_dev), staged with every
        # batch from the optimizer's param_group: a learning-rate schedule is followed without a new capture
        self.hyper = z(4, dtype=torch.float64)
        self._hyper_on_device = USE_ADAM_KERNEL and ops.adam_hyper(optimizer) is not None
        self._e_loss = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev)          # d / d out of out[3], the total loss
        self.out: Optional[Tensor] = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.graph_opt: Optional[torch.cuda.CUDAGraph] = None      # split mode (grad_sync): Adam as a graph of its own
        self._grads: list = []
        self._training: Optional[bool] = None
        self._iota = torch.arange(nt, device=dev)
        self._captured_hp = None
        self._late = (0, False)                               # (dropout-counter advance owed to the step's end, Adam's counters advanced)
        self._warming = False
        # The captured forward runs on ALIASES of the parameters (same storage, distinct autograd leaves).  A leaf's
        # gradient sink (its AccumulateGrad node) belongs to the stream it was created on and lives as long as any
        # autograd graph mentions it: after an eager step whose loss the caller still holds, the parameters' own
        # sinks sit on the default stream and would pull that stream into the capture (hipStreamEndCapture then
        # crashes).  The aliases' sinks are born inside the capture; the gradients are handed to the optimizer's
        # parameters by assignment.
        lit_model.model._materialize_bd(int(template["bd"]["x"].shape[1]), dev)
        self._alias = {n: ops.alias_of(p) for n, p in lit_model.model.named_parameters()}
        by_id = {id(p): self._alias[n] for n, p in lit_model.model.named_parameters()}
        self._params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad and id(p) in by_id]
        self._leaves = [by_id[id(p)] for p in self._params]
        self.draws = None                                     # tests: fixed (tx pos/neg, bd pos/neg/dp/dn, dst_neg)
        # queue the backward's partial sums and run them as one launch (ops.deferred_reductions).  Only sound when no
        # autograd node READS a parameter gradient before the backward ends (a parameter used by two nodes has its two
        # gradients added on the spot): verified numerically during the warm-up of every capture, see _warm_up
        self.defer_sums = True

    @staticmethod
    def _bd_dtype(lit_model, template):
        dt, src = lit_model.model.compute_dtype, template["bd"]["x"].dtype
        return dt if (src.is_floating_point and dt in (torch.bfloat16, torch.float16)) else src

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
    def _sampler_segments(self, dst: dict, ix: dict, n_total: int) -> list:
        n = int(ix["lab"].numel())
        return [(dst["lab"], ix["lab"], "const", dst["n_clusters"], 0, 0),          # padding: the masked-out cluster
                (dst["members"], ix["members"], "div", n, 1, 0),
                (dst["counts"], ix["counts"], "const", 0, 0, 0), (dst["offsets"], ix["offsets"], "const", 0, 0, 0),
                (dst["cdf_pos_t"], ix["cdf_pos_t"], "const", 0, 0, 0), (dst["cdf_neg_t"], ix["cdf_neg_t"], "const", 0, 0, 0)]

    @torch.no_grad()
    def stage(self, batch) -> None:
        from .triplet_loss import _cached_index, _masked_count
        if not self.fits(batch):
            raise ValueError("batch does not fit this bucket")
        lit, s = self.lit, self.sizes
        tx, bd = batch["tx"], batch["bd"]
        n_tx, n_bd, nt = tx.num_nodes, bd.num_nodes, s["tx"]
        cache = batch_cache(batch)
        store = cache.get("persistent")
        g_tt = edge_graph(cache, TX_TX, batch[TX_TX].edge_index, n_tx, n_tx, need_by_src=True, validate="deferred")
        g_tb = edge_graph(cache, TX_BD, batch[TX_BD].edge_index, n_tx, n_bd, need_by_src="lazy", validate="deferred")
        if not g_tb.src_unique():
            raise NotImplementedError("a transcript with two tx-belongs-bd edges (heterodata.py:147 assigns one)")
        e_tb = g_tb.n_edges
        # what depends on the tile set only (kept across epochs by tiles.TilePartition; per batch object otherwise)
        keep = store if store is not None else cache
        ix_tx = keep.get("tx_triplet_index") if store is not None else None
        if ix_tx is None:
            ix_tx = _cached_index(lit.loss_tx.selector, "tx_triplet_index", tx["cluster"], tx["mask"], cache)
        a0 = ("rows_over_masked", nt)                         # loss_tx: mean over the padded rows -> mean over the masked real ones
        if a0 not in ix_tx:
            ix_tx[a0] = (float(nt) / _masked_count(ix_tx["mask"])).reshape(1)
        ix_bd = keep.get("bd_metric_index") if store is not None else None
        if ix_bd is None:
            ix_bd = _cached_index(lit.loss_bd.selector, "bd_metric_index", bd["cluster"],
                                  bd["mask"] & (bd["cluster"] >= 0), cache)
        if "weight" not in ix_bd:
            ix_bd["weight"] = ix_bd["mask"].float() / _masked_count(ix_bd["mask"])
        by_gene = keep.get("tx_by_gene")
        if by_gene is None:
            by_gene = keep["tx_by_gene"] = ops.rows_by_id(tx["x"], self.n_genes)
        ei = batch[TX_BD].edge_index
        of_tx = keep.get("sg_of_tx")
        if of_tx is None:
            of_tx = keep["sg_of_tx"] = ops.anchor_index(ei[0], n_tx)
        w = lit._scheduled_weights(lit._w_start, lit._w_end)
        bd_x, x_dt = bd["x"], self.nodes["bd"]["x"].dtype
        if bd_x.dtype != x_dt:
            key = ("bd_x", x_dt)
            hit = keep.get(key)
            if hit is None or hit.shape != bd_x.shape:
                hit = keep[key] = bd_x.to(x_dt).contiguous()
            bd_x = hit
        fb = ops.float_bits
        dummies = ("mod", n_tx, nt - n_tx)                    # dummy transcripts, round robin
        N = self.nodes
        segs = [
            # nodes: dummies are copies of node 0 (gene id: the last gene, whose rows-by-gene group they extend)
            (N["tx"]["x"], tx["x"], "const", self.n_genes - 1, 0, 0),
            (N["tx"]["pos"], tx["pos"], "tile", 2, 0, 0), (N["tx"]["batch"], tx["batch"], "tile", 1, 0, 0),
            (N["bd"]["x"], bd_x, "tile", max(int(bd_x[0].numel()), 1), 0, 0),
            (N["bd"]["pos"], bd["pos"], "tile", 2, 0, 0), (N["bd"]["batch"], bd["batch"], "tile", 1, 0, 0),
            (self.batch_all[nt:], bd["batch"], "tile", 1, 0, 0, s["graphs"]),
            (self.by_gene.indptr, by_gene.indptr[: self.n_genes], "const", nt, 0, 0),
            (self.by_gene.col, by_gene.col, "div", n_tx, 1, 0),
            # segmentation triplets: padded ones carry -1 and are skipped by the kernels
            (self.sg_src, ei[0], *dummies, 0), (self.sg_pos, ei[1], "const", -1, 0, 0),
            (self.sg_of_tx, of_tx, "const", -1, 0, 0), (self.tx_state, None, "const", 0, 0, 0),
            (self.bd_weight, ix_bd["weight"], "const", 0, 0, 0),
            (self.n_bd, None, "const", n_bd, 0, 0),
            (self.minmax[0], None, "const", fb(float("inf")), 0, 0), (self.minmax[1], None, "const", fb(float("-inf")), 0, 0),
            (self.scal[0:1], ix_tx[a0], "const", 0, 0, 0), (self.scal[1:2], None, "const", fb(1.0), 0, 0),
            (self.scal[2:3], None, "const", fb(s["e_tb"] / max(e_tb, 1) if n_bd > 1 else 0.0), 0, 0),   # :173-175
            (self.scal[3:4], None, "const", fb(w[0]), 0, 0), (self.scal[4:5], None, "const", fb(w[1]), 0, 0),
            (self.scal[5:6], None, "const", fb(w[2]), 0, 0),
        ]
        if self._hyper_on_device:
            hp = ops.adam_hyper(self.opt)
            if hp is None:
                raise RuntimeError("the optimizer's learning rate became a tensor / its groups changed after the capture")
            segs += [(self.hyper[i:i + 1], None, "const", ops.double_bits(v), 0, 0) for i, v in enumerate(hp)]
        segs += padded_view_segments(self.g_tt.by_dst, g_tt.by_dst, n_tx, None)
        segs += padded_view_segments(self.g_tt.by_src, g_tt.require_by_src(), n_tx, None)
        segs += padded_view_segments(self.g_tb.by_dst, g_tb.by_dst, n_bd, dummies)
        segs += self._sampler_segments(self.ix_tx, ix_tx, nt)
        segs += self._sampler_segments(self.ix_bd, ix_bd, s["bd"])
        ops.stage(segs, self.dev)

    # ------------------------------------------------------------------------------------------- the step
    def _run(self) -> None:
        self._run_grads()
        self._opt_step()

    def _opt_step(self) -> None:
        """Adam on the hand-written kernel (two launches for all tensors -- one when the step's first launch advanced the
        step counters; torch's fused multi-tensor Adam takes three and ~40 us for the encoder's 60 small tensors); any other
        optimizer steps itself.  The dropout counter a merged-draws step read un-advanced moves on here."""
        inc, adv = self._late
        self._late = (0, False)
        enc = self.lit.model
        if USE_ADAM_KERNEL and ops.adam_step(self.opt, steps_advanced=adv, counter=enc._step_dev if inc else None,
                                             counter_inc=inc, hyper_dev=self.hyper if self._hyper_on_device else None):
            return
        if adv:
            raise RuntimeError("the step advanced Adam's counters for segger_adam_step, which then refused the optimizer")
        if self._hyper_on_device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("segger_adam_step refused an optimizer whose hyper-parameters this step stages on the "
                                   "device (non-fp32 / non-contiguous parameters or gradients, an AMP grad scaler attached)")
            # eager warm-up, optimizer state present, and the kernel still declines (what adam_hyper() cannot see from the
            # param_group: gradient dtype / layout, a GradScaler's grad_scale / found_inf): take the route that works from
            # here on -- torch's own step inside the capture, (lr, betas, eps) baked in and part of the re-capture key
            if any(self.opt.state.get(p) for g in self.opt.param_groups for p in g["params"]):
                self._hyper_on_device = False
        self.opt.step()
        if inc:
            ops.step_advance(enc._step_dev, inc)

    def _run_grads(self) -> None:
        lit, enc, s = self.lit, self.lit.model, self.sizes
        nt = s["tx"]
        tx, bd = self.nodes["tx"], self.nodes["bd"]
        graphs = {TX_TX: self.g_tt, TX_BD: self.g_tb, "tx_by_gene": self.by_gene,
                  "pos_all": (self.pos_all, self.batch_all), "minmax": self.minmax}
        fixed = self.draws
        drawn = None
        self._late = (0, False)
        if MERGED_DRAWS and fixed is None:
            # ALL of the step's random draws in its first launch (ops.step_draws), from the dropout counter as it stands + the
            # advance it receives at the END of the step (inside Adam's launch): the same masks and triplets as advancing
            # first.  Adam's step counters move in the same launch (nothing reads them before the update).
            inc = STEP_INC if enc.training else 0
            views = enc.plane_views(graphs, seed_offset=inc) if enc.training else None
            p_drop = enc.conv_layers[0].conv[TX_TX].dropout
            adv = None
            # (not while warming up -- that step ends in optimizer.step() -- and not in split mode, whose empty steps replay
            # Adam's graph without this one)
            if USE_ADAM_KERNEL and not self._warming and not self.split:
                adv = ops.adam_step_counters(self.opt, self._params)
            planes, samples, dst_neg = ops.step_draws(
                [(c, sd) for _, _, c, sd in views] if views else [], enc.n_heads, p_drop,
                [(self.ix_tx, 0x7478 + inc), (self.ix_bd, 0x6264 + inc)], (self.sg_pos, 0, self.n_bd, 0x7367 + inc),
                enc._step_dev, advance=adv)
            graphs["draws"] = (enc.planes_of(views, planes) if views else None, inc)
            drawn = (samples, dst_neg)
            self._late = (inc, adv is not None)
        z = torch.func.functional_call(
            enc, self._alias,
            ({"tx": tx["x"], "bd": bd["x"]}, {TX_TX: None, TX_BD: None}, {"tx": tx["pos"], "bd": bd["pos"]},
             {"tx": tx["batch"], "bd": bd["batch"]}),
            dict(num_graphs=s["graphs"], graphs=graphs))
        step = enc._step_dev                                  # advanced by the forward: a fresh stream per replay
        # the three losses on the staged sampler indices and their weighted sum as one autograd node (ops.loss_head):
        # means are taken over the padded rows by the kernels and rescaled to the masked real ones by `head_a`
        if drawn is not None:
            (pos, neg, _, _), (bpos, bneg, dp, dn) = drawn[0]
            dst_neg = drawn[1]
        elif fixed is None:
            pos, neg, _, _ = ops.triplet_sample(self.ix_tx, seed=0x7478, seed_dev=step)
            bpos, bneg, dp, dn = ops.triplet_sample(self.ix_bd, seed=0x6264, seed_dev=step)
            # segmentation loss over the real tx-belongs-bd edges (lightning_model.py:167-189): negatives in [0, n_bd_real)
            dst_neg = ops.sample_negatives(self.sg_pos, 0, self.n_bd, seed=0x7367, seed_dev=step)
        else:
            (pos, neg), (bpos, bneg, dp, dn), dst_neg = fixed["tx"], fixed["bd"], fixed["dst_neg"]
        spec = ops.LossHeadSpec((self._iota, pos, neg, lit.loss_tx.margin, lit.loss_tx.eps),
                                (bpos, bneg, dp, dn, self.bd_weight, 1e-8),
                                (self.sg_src, self.sg_pos, dst_neg, lit._sg_margin, 1e-6, self.g_tb.by_dst, True),
                                sg_kind=lit._sg_loss_type, tx_anchors_are_rows=True, sg_of_tx=self.sg_of_tx,
                                tx_state=self.tx_state, grad_out_hint=self._e_loss)
        spec.defer_finish = DEFER_LOSS_FINISH      # the losses are read after the backward (self.out after the replay)
        out = ops.loss_head(z["tx"], z["bd"], self.head_a, self.scal[3:6], spec)
        if self.defer_sums:                                   # ~30 partial sums of the backward as one launch
            with ops.deferred_reductions(self.dev):
                grads = torch.autograd.grad(out, self._leaves, self._e_loss, allow_unused=True)
        else:
            grads = torch.autograd.grad(out, self._leaves, self._e_loss, allow_unused=True)
        for p, g in zip(self._params, grads):
            p.grad = g
        self.out = out.detach()

    def _warm_up(self, keep) -> None:
        """One eager step before the capture.  With deferred partial sums it is run twice from the same random
        streams -- sums launched where they are produced, then queued -- and the two sets of parameter gradients must
        agree (up to the order of float atomics in the loss kernels); if they do not, some autograd node consumed a
        gradient before the queued sums ran, and this step falls back to immediate sums."""
        if self.defer_sums:
            self.defer_sums = False
            self._run_grads()
            ref = [None if p.grad is None else p.grad.detach().clone() for p in self._params]
            with torch.no_grad():
                self.lit.model._step_dev.copy_(keep[3])       # same dropout masks and sampler draws again
                self.tx_state.zero_()                         # (what the staging launch does before every step)
            self.defer_sums = True
            self._run_grads()
            names = {id(p): n for n, p in self.lit.model.named_parameters()}
            why = None
            for p, r in zip(self._params, ref):
                g = p.grad
                if (g is None) != (r is None):
                    why = f"{names.get(id(p))}: gradient {'missing' if g is None else 'unexpected'}"
                elif g is not None:
                    scale, diff = float(r.abs().max()), float((g - r).abs().max())
                    # the loss kernels accumulate with (packed 16-bit) float atomics, so two runs of the same backward
                    # differ by a few per cent of a small gradient (seen: 1.4 % on lin_last's bias); a gradient that was
                    # read before its sum was launched is garbage, not a rounding difference: 10 % of the largest
                    # entry separates the two
                    if not bool(torch.isfinite(g).all()) or diff > 0.1 * scale + 1e-6:
                        why = f"{names.get(id(p))}: differs by {diff:.3e} (scale {scale:.3e})"
                if why:
                    break
            if why:
                self.defer_sums = False
                self.tx_state.zero_()
                import warnings
                warnings.warn("GraphedTrainStep: a parameter gradient is consumed inside the backward pass "
                              f"({why}); partial sums are launched where they are produced (more kernel nodes)")
                self._run_grads()
            self.opt.step()
        else:
            self._run()

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

    def _hyper(self) -> tuple:
        """The optimizer hyper-parameters a capture bakes in: none of (lr, betas, eps) when the Adam kernel reads them from
        the staged device array (``_hyper_on_device``: the common case -- a scheduler changing ``lr`` every step replays the
        same graph); otherwise (torch's own optimizer step inside the capture: Python floats become kernel arguments) all
        of them -- a change between steps then triggers a new capture instead of being ignored."""
        if self._hyper_on_device:
            return tuple((g.get("weight_decay", 0), bool(g.get("amsgrad")), bool(g.get("maximize"))) for g in self.opt.param_groups)
        return tuple((g["lr"] if not isinstance(g["lr"], Tensor) else id(g["lr"]), tuple(g["betas"]), g["eps"],
                      g.get("weight_decay", 0)) for g in self.opt.param_groups)

    def step(self, batch, capture: bool = True) -> Tensor:
        """-> [loss_tx, loss_bd, loss_sg, loss] (a static device tensor, overwritten by the next step).  Optimizer step
        hooks (``register_step_pre_hook`` / ``_post_hook``) do not run inside a replay; a changed learning rate / betas / eps
        is staged with the batch (no new capture) on the Adam-kernel route, and re-captures the bucket on any other."""
        lit = self.lit
        self.stage(batch)
        if self.graph is not None and self._captured_hp != self._hyper():
            self.graph = self.graph_opt = None                # (the old graphs' memory returns to the pool with them)
        if not capture:
            self._run()
        elif self.graph is None or self._training != lit.model.training:
            lit.model._materialize_bd(self.nodes["bd"]["x"].shape[1], self.dev)
            keep = self._snapshot()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                     # warm-up: lazy inits, allocator, optimizer state
                self._warming = True
                try:
                    self._warm_up(keep)
                finally:
                    self._warming = False
            torch.cuda.current_stream().wait_stream(side)
            self._restore(keep)                               # ... which must not count as a training step
            self.tx_state.zero_()                             # (the warm-up's forward left its chains: staged state again)
            aliases = list(self._alias.values())
            ops.invalidate_weights(aliases)                   # the captured step starts with the weight refresh ...
            self.graph = torch.cuda.CUDAGraph()
            with ops.pack_scope(aliases):                     # ... of this step's own packs only
                if not self.split:
                    with torch.cuda.graph(self.graph):
                        self._run()
                else:                                         # forward + backward (+ pack) | <exchange> | Adam
                    with torch.cuda.graph(self.graph):
                        self._run_grads()
                        if self.grad_bucket is not None:      # gradients -> the bucket's flat buffer; .grad = its slices
                            self.grad_bucket.pack()
                    self.graph_opt = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph_opt, pool=self.graph.pool()):
                        self._opt_step()
            # the graph holds raw pointers into these buffers: they must outlive it whatever happens to the cache
            self._pack_refs = [(pk, pk.w, pk.b, pk._wt) for pk in ops.packs_of(aliases)]
            self._grads = [p.grad for p in self._params]
            self._training = lit.model.training
            self._captured_hp = self._hyper()
            self._replay()                                    # (capturing runs nothing)
        else:
            self._replay()
        self._params_changed()
        return self.out

    def _params_changed(self) -> None:
        """The replay stepped the parameters behind Python's back: eager users' compute-dtype copies are stale."""
        ops.invalidate_weights(p for g in self.opt.param_groups for p in g["params"])

    def _replay(self, empty: bool = False) -> None:
        """Fused mode: one replay.  Split mode: replay forward + backward (``empty``: zero the gradients instead -- this
        rank ran out of batches), hand this bucket's gradient tensors to the parameters, let ``grad_sync`` exchange
        them (exactly ONE collective per step on every rank: the warm-up of a capture never synchronises), replay
        Adam."""
        if not self.split:
            self.graph.replay()
            return
        if self.grad_bucket is not None:
            if empty:
                self.grad_bucket.zero()
            else:
                self.graph.replay()                           # ends with the pack into the flat buffer
                for p, v in zip(self.grad_bucket.params, self.grad_bucket.views):
                    p.grad = v
            self.grad_bucket.all_reduce_mean(packed=True)
            self.graph_opt.replay()                           # reads the buffer's slices
            return
        if empty:
            torch._foreach_zero_([g for g in self._grads if g is not None])
        else:
            self.graph.replay()
        for p, g in zip(self._params, self._grads):
            p.grad = g
        self.grad_sync()          # must average IN PLACE: the captured optimizer reads the tensors .grad points at now
        self.graph_opt.replay()

    def empty_step(self) -> None:
        self._replay(empty=True)
        self._params_changed()


class GraphedTrainer:
    """``trainer.step(batch)``: stage the batch into the tightest bucket it fits and replay.  A new bucket is captured
    when none fits or the tightest one would pad the transcript side by more than two granules; with ``max_buckets``
    captured, the tightest fit is used whatever it wastes, and a batch no bucket holds is an error."""

    def __init__(self, lit_model, optimizer, granularity: float = 1.06, max_buckets: int = 24, grad_sync=None,
                 grad_bucket=None):
        """Data parallelism: ``grad_bucket`` = a ``dp.FlatGradBucket`` over the model's parameters (the captured
        backward packs into its flat buffer, Adam reads its slices; per step one all-reduce and one divide run eagerly
        between the two graphs), or ``grad_sync`` = any callable that averages the tensors ``.grad`` points at IN PLACE.
        One collective per ``step`` call on every rank; ``step(None)`` is the empty step of a rank that ran out of
        batches: zeros into the exchange, then Adam."""
        self.lit, self.opt, self.granularity, self.max_buckets = lit_model, optimizer, granularity, max_buckets
        if grad_bucket is None and hasattr(getattr(grad_sync, "__self__", None), "pack"):
            grad_bucket, grad_sync = grad_sync.__self__, None
        self.grad_sync, self.grad_bucket = grad_sync, grad_bucket
        self.buckets: List[GraphedTrainStep] = []
        self.n_captures = 0
        self._last: Optional[GraphedTrainStep] = None

    def _empty_step(self) -> None:
        if self.grad_sync is None and self.grad_bucket is None:
            return
        if self._last is not None and self._last.graph_opt is not None:
            self._last.empty_step()
            return
        if self.grad_bucket is not None:                      # no bucket captured yet on this rank: eager
            self.grad_bucket.zero()
            self.grad_bucket.all_reduce_mean(packed=True)
        else:
            for g in self.opt.param_groups:
                for p in g["params"]:
                    if p.requires_grad:
                        p.grad = torch.zeros_like(p)
            self.grad_sync()
        self.opt.step()

    def step(self, batch) -> Optional[Tensor]:
        if batch is None:
            return self._empty_step()
        fit = [b for b in self.buckets if b.fits(batch)]
        best = min(fit, key=lambda b: b.waste(batch)) if fit else None
        full = len(self.buckets) >= self.max_buckets
        if best is None or (best.waste(batch) > self.granularity ** 2 and not full):
            if full:
                raise RuntimeError(f"no captured bucket holds this batch and {self.max_buckets} buckets exist: "
                                   f"raise `granularity` or `max_buckets`")
            best = GraphedTrainStep(self.lit, self.opt, step_bucket(batch, self.granularity), batch, self.grad_sync,
                                    self.grad_bucket)
            self.buckets.append(best)
            self.n_captures += 1
        self._last = best
        return best.step(batch)
