"""``LitISTEncoder`` for MI355X: constructor, ``forward`` / ``get_losses`` /
``training_step`` / ``validation_step`` / ``predict_step`` /
``configure_optimizers`` as in reference
``src/segger/models/lightning_model.py:19-303`` -- so ``segger segment`` can
build it from the same CLI-parsed arguments -- over the HIP kernels:

* ``forward``       -> :class:`segger_amd.ist_encoder.ISTEncoder`
* ``predict_step``  -> fused cosine / arg-max / assignment kernel (``:275-293``)
* segmentation loss -> fused triplet kernel (``:178-187``) / fused BCE kernel
  (``:190-207``; torch ops only on the CPU or for an odd channel count).

``lightning`` is used when importable; otherwise a minimal stand-in base class
provides ``log`` / ``trainer`` / ``current_epoch`` / ``device`` so the module
can be driven by a plain loop (bench.py, tests).
"""
from __future__ import annotations

import math
from typing import Any, Optional

import torch
from torch import Tensor
from torch.nn import BCEWithLogitsLoss

from . import ops
from .graph import batch_cache, edge_graph, flush_validation
from .hetero import TX_BD, TX_NB_BD
from .ist_encoder import ISTEncoder
from .triplet_loss import MetricLoss, TripletLoss

try:  # pragma: no cover - lightning is not in the build image
    from lightning import LightningModule as _Base
    _HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    _HAVE_LIGHTNING = False

    class _Base(torch.nn.Module):
        """The slice of LightningModule this path touches."""

        def __init__(self):
            super().__init__()
            self.trainer = None
            self.current_epoch = 0
            self.logged: dict = {}
            self.hparams: dict = {}

        def save_hyperparameters(self, **kw):
            import inspect
            frame = inspect.currentframe().f_back
            args = {k: v for k, v in frame.f_locals.items() if k not in ("self", "__class__") and not k.startswith("_")}
            self.hparams = args

        def log(self, name, value, **kw):
            # detached, like Lightning's: a logged loss must not keep the step's autograd graph alive
            # (a live graph from an earlier step also breaks a later hipGraph capture of the backward)
            self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def setup(self, stage):
            return None

        automatic_optimization = True

        def optimizers(self):
            """Lightning's ``LightningModule.optimizers()``: what the trainer configured (one optimizer or a list)."""
            opts = list(getattr(self.trainer, "optimizers", None) or [])
            return opts[0] if len(opts) == 1 else opts


class LitISTEncoder(_Base):
    def __init__(
        self,
        n_genes: int,
        in_channels: int,
        hidden_channels: int = 64,
        out_channels: int = 64,
        n_mid_layers: int = 2,
        n_heads: int = 2,
        learning_rate: float = 1e-3,
        sg_loss_type: str = 'triplet',
        tx_margin: float = 0.3,
        sg_margin: float = 0.4,
        tx_weight_start: float = 1.,
        tx_weight_end: float = 1.,
        bd_weight_start: float = 1.,
        bd_weight_end: float = 1.,
        sg_weight_start: float = 0.,
        sg_weight_end: float = 0.5,
        update_gene_embedding: bool = True,
        use_positional_embeddings: bool = True,
        normalize_embeddings: bool = True,
    ):
        super().__init__()
        self.save_hyperparameters()
        if sg_loss_type not in ("triplet", "bce"):
            # the reference raises this from setup() (:120-124); failing at construction is stricter
            raise ValueError(f"Unrecognized segmentation loss: '{sg_loss_type}'. "
                             f"Acceptable values are 'triplet' and 'bce'.")
        self.model = ISTEncoder(
            n_genes=n_genes, in_channels=in_channels, hidden_channels=hidden_channels,
            out_channels=out_channels, n_mid_layers=n_mid_layers, n_heads=n_heads,
            normalize_embeddings=normalize_embeddings,
            use_positional_embeddings=use_positional_embeddings,
        )
        self.learning_rate = learning_rate
        self._sg_loss_type = sg_loss_type
        self._tx_margin = tx_margin
        self._sg_margin = sg_margin
        self._w_start = torch.tensor([tx_weight_start, bd_weight_start, sg_weight_start])
        self._w_end = torch.tensor([tx_weight_end, bd_weight_end, sg_weight_end])
        self._freeze_gene_embedding = not update_gene_embedding
        self.loss_tx = None
        self.loss_bd = None
        self._max_epochs_override: Optional[int] = None
        self._graphed_kw: Optional[dict] = None        # enable_graphed_training()
        self._graphed_trainer = None
        import os
        on = lambda name: os.environ.get(name, "").strip().lower() in ("1", "true", "yes", "on")
        if on("SEGGER_AMD_FAST"):
            self.fast()            # `SEGGER_AMD_FAST=1 segger segment ...`: the CLI builds the module from parsed arguments only
        elif on("SEGGER_AMD_GRAPHED"):
            # captured steps WITHOUT changing the arithmetic width: the reference's fp32 storage, every training step one
            # hipGraph replay -- the recommended drop-in (bench.py `strong.graphed_f32`)
            self.enable_graphed_training()

    # ------------------------------------------------------------------ setup
    def set_similarities(self, tx_similarity: Tensor, bd_similarity: Tensor) -> None:
        """What ``setup`` takes from the datamodule (lightning_model.py:109-115)."""
        self.loss_tx = TripletLoss(tx_similarity, margin=self._tx_margin)
        self.loss_bd = MetricLoss(bd_similarity)

    def setup(self, stage):
        dm = getattr(self.trainer, "datamodule", None)
        if dm is None or not (hasattr(dm, "tx_similarity") and hasattr(dm, "bd_similarity")):
            raise TypeError(
                f"Expected data module to be `ISTDataModule` but got {type(dm).__name__}.")
        if hasattr(dm, "gene_embedding"):                                    # :95-106
            w = dm.gene_embedding
            if not isinstance(w, Tensor):                                    # polars frame in segger
                from importlib import import_module
                fields = import_module("segger.io.fields").StandardTranscriptFields()
                w = w.drop(fields.feature).to_torch()
            self.model.lin_first['tx'] = torch.nn.Embedding.from_pretrained(
                w.to(torch.float), freeze=self._freeze_gene_embedding)
        self.set_similarities(dm.tx_similarity, dm.bd_similarity)
        return super().setup(stage)

    # ---------------------------------------------------------------- forward
    def forward(self, batch) -> dict:
        return self.model(
            batch.x_dict, batch.edge_index_dict, batch.pos_dict, batch.batch_dict,
            num_graphs=getattr(batch, "num_graphs", None), cache=batch_cache(batch),
        )

    def _scheduled_weights(self, w_start: Tensor, w_end: Tensor, normalize: bool = True) -> Tensor:
        """Cosine ramp (lightning_model.py:136-149).  Stays on the host: three floats."""
        trainer_max = self._max_epochs_override
        if trainer_max is None:
            trainer_max = getattr(self.trainer, "max_epochs", 1) or 1
        max_epochs = max(1, trainer_max - 1)
        t = min(self.current_epoch, max_epochs) / max_epochs
        alpha = 0.5 * (1.0 + math.cos(math.pi * t))
        w = w_end + (w_start - w_end) * alpha
        if normalize:
            w = w / (w.sum() + 1e-8)
        return w

    def _segmentation_loss(self, embeddings, batch, dst_neg: Optional[Tensor] = None) -> Tensor:
        src_pos, dst_pos = batch[TX_BD].edge_index
        z_tx, z_bd = embeddings['tx'], embeddings['bd']
        num_bd = z_bd.size(0)
        n = src_pos.size(0)
        if num_bd <= 1:                                                      # :173-175
            return torch.tensor(0.0, device=z_bd.device, requires_grad=True)
        if dst_neg is None:                                                  # :178-180
            if dst_pos.is_cuda:
                dst_neg = ops.sample_negatives(dst_pos, num_bd)              # the same draw, one launch
            else:
                dst_neg = (dst_pos + torch.randint(1, num_bd, (n,), device=dst_pos.device)) % num_bd
        if self._sg_loss_type == 'triplet':
            # the positives are the edges' own destinations: their grouping is the by-destination view of this edge
            # store, which the encoder's forward has already built and cached on the batch
            g = edge_graph(batch_cache(batch), TX_BD, batch[TX_BD].edge_index, z_tx.size(0), num_bd,
                           need_by_src="lazy" if torch.is_grad_enabled() else False, validate="deferred")
            return ops.triplet_edge_loss(z_tx, z_bd, src_pos, dst_pos, dst_neg, self._sg_margin, eps=1e-6,
                                         pos_groups=g.by_dst, anchors_unique=g.src_unique)
        # BCE on dot-product logits (:190-207); unique/inverse in the reference only dedups gathers
        if z_tx.is_cuda and z_tx.shape[1] % 2 == 0 and z_tx.dtype == z_bd.dtype:
            g = edge_graph(batch_cache(batch), TX_BD, batch[TX_BD].edge_index, z_tx.size(0), num_bd,
                           need_by_src="lazy" if torch.is_grad_enabled() else False, validate="deferred")
            return ops.bce_edge_loss(z_tx, z_bd, src_pos, dst_pos, dst_neg, pos_groups=g.by_dst, anchors_unique=g.src_unique)
        src = torch.cat([src_pos, src_pos]).long()
        dst = torch.cat([dst_pos, dst_neg]).long()
        logits = (z_tx.float()[src] * z_bd.float()[dst]).sum(dim=-1)
        labels = torch.cat([torch.ones(n, device=logits.device), torch.zeros(n, device=logits.device)])
        return BCEWithLogitsLoss()(logits, labels)

    def get_losses(self, batch, dst_neg: Optional[Tensor] = None, embeddings: Optional[dict] = None,
                   uniforms: Optional[tuple] = None):
        """(loss_tx, loss_bd, loss_sg, loss), lightning_model.py:151-213.  ``embeddings`` lets a caller supply
        the encoder output it already has;
        ``dst_neg`` / ``uniforms`` = (per-transcript, per-boundary) selector draws replace the random numbers
        (replaying recorded vectors)."""
        if self.loss_tx is None or self.loss_bd is None:
            raise RuntimeError("call setup() (or set_similarities) before computing losses")
        if embeddings is None:
            embeddings = self.forward(batch)
        tx_mask = batch['tx']['mask']
        # (kept with the batch: the samplers' per-batch indices are keyed by the mask tensor they were built from, and a
        # mask recomputed every step would rebuild the boundary index -- ~50 small launches -- every step)
        bm, bc = batch['bd']['mask'], batch['bd']['cluster']
        cache = batch_cache(batch)
        mkey = ("bd_loss_mask", bm.data_ptr(), bc.data_ptr(), int(bm.numel()))
        bd_mask = cache.get(mkey)
        if bd_mask is None:
            bd_mask = cache[mkey] = bm & (bc >= 0)
        u_tx, u_bd = uniforms if uniforms is not None else (None, None)
        if (self.fused_loss_head and embeddings['tx'].is_cuda and embeddings['tx'].shape[1] % 2 == 0
                and embeddings['tx'].shape[0] > 0 and embeddings['bd'].shape[0] > 0):
            return self._losses_fused(batch, embeddings, tx_mask, bd_mask, u_tx, u_bd, dst_neg)
        loss_tx = self.loss_tx.forward_masked(embeddings['tx'], batch['tx']['cluster'], tx_mask, batch_cache(batch),
                                              uniforms=u_tx)
        loss_bd = self.loss_bd.forward_masked(embeddings['bd'], batch['bd']['cluster'], bd_mask, uniforms=u_bd,
                                              cache=batch_cache(batch))
        loss_sg = self._segmentation_loss(embeddings, batch, dst_neg)
        w_tx, w_bd, w_sg = (float(v) for v in self._scheduled_weights(self._w_start, self._w_end))
        loss = w_tx * loss_tx + w_bd * loss_bd + w_sg * loss_sg
        return loss_tx, loss_bd, loss_sg, loss

    fused_loss_head = True      # the three losses + their weighted sum as one autograd node (ops.loss_head) on the GPU

    def _losses_fused(self, batch, embeddings, tx_mask, bd_mask, u_tx, u_bd, dst_neg):
        """The same four values through ``ops.loss_head``: samplers as in the unfused route (same draws from the same
        generators, in the same order), then ONE node for the three loss kernels and the weighted sum."""
        from .triplet_loss import _cached_index, _masked_count
        cache = batch_cache(batch)
        z_tx, z_bd = embeddings['tx'], embeddings['bd']
        dev = z_tx.device
        n_tx, n_bd = z_tx.shape[0], z_bd.shape[0]
        tx_lab, bd_lab = batch['tx']['cluster'], batch['bd']['cluster']
        ix_tx = _cached_index(self.loss_tx.selector, "tx_triplet_index", tx_lab, tx_mask, cache)
        if "anchors" not in ix_tx:
            ix_tx["anchors"] = torch.arange(n_tx, device=dev)
            ix_tx["rescale"] = float(n_tx) / _masked_count(tx_mask)
        if "head_a" not in ix_tx:               # per-loss rescaling: loss_tx from a mean over all rows to the masked ones
            ix_tx["head_a"] = torch.cat([ix_tx["rescale"].reshape(1).float(), torch.ones(2, device=dev)])
        pos, neg, _, _ = self.loss_tx.selector.sample_triplets(tx_lab, u_tx, index=ix_tx)
        ix_bd = _cached_index(self.loss_bd.selector, "bd_metric_index", bd_lab, bd_mask, cache)
        if "weight" not in ix_bd:
            ix_bd["weight"] = bd_mask.float() / _masked_count(bd_mask)
        bpos, bneg, dp, dn = self.loss_bd.selector.sample_triplets(bd_lab, u_bd, index=ix_bd)
        sg = None
        if n_bd > 1:                                                         # :173-175
            src_pos, dst_pos = batch[TX_BD].edge_index
            if dst_neg is None:
                dst_neg = ops.sample_negatives(dst_pos, n_bd)                # :178-180, one launch
            g = edge_graph(cache, TX_BD, batch[TX_BD].edge_index, n_tx, n_bd,
                           need_by_src="lazy" if torch.is_grad_enabled() else False, validate="deferred")
            sg = (src_pos, dst_pos, dst_neg, self._sg_margin, 1e-6, g.by_dst, g.src_unique)
            # "the segmentation triplet anchored at transcript r": what lets the one-launch loss head gather a transcript's
            # whole gradient row instead of scattering into it; depends on the edge list only -> kept with the tile set
            store = cache.get("persistent")
            keep, okey = (store, "sg_of_tx") if store is not None else (cache, ("sg_of_tx", src_pos.data_ptr(), int(src_pos.numel()), n_tx))
            of_tx = (lambda: keep.get(okey) if keep.get(okey) is not None else keep.setdefault(okey, ops.anchor_index(src_pos, n_tx)))
        w = self._scheduled_weights(self._w_start, self._w_end)
        key = tuple(float(v) for v in w)
        wdev = self.__dict__.setdefault("_head_weights", {})
        b = wdev.get((key, dev))
        if b is None:
            if len(wdev) > 64:
                wdev.clear()
            b = wdev[(key, dev)] = torch.tensor(key, dtype=torch.float32, device=dev)
        spec = ops.LossHeadSpec((ix_tx["anchors"], pos, neg, self.loss_tx.margin, self.loss_tx.eps),
                                (bpos, bneg, dp, dn, ix_bd["weight"], 1e-8), sg, sg_kind=self._sg_loss_type,
                                tx_anchors_are_rows=True, sg_of_tx=of_tx if sg is not None else None)
        out = ops.loss_head(z_tx, z_bd, ix_tx["head_a"], b, spec)
        return out[0], out[1], out[2], out[3]

    def _step(self, batch, prefix: str) -> Tensor:
        loss_tx, loss_bd, loss_sg, loss = self.get_losses(batch)
        bs = getattr(batch, "num_graphs", 1)
        for name, v in (("loss_tx", loss_tx), ("loss_bd", loss_bd), ("loss_sg", loss_sg)):
            self.log(f"{prefix}:{name}", v, prog_bar=True, batch_size=bs)
        return loss

    def training_step(self, batch, batch_idx: int) -> Tensor:
        if self._graphed_kw is not None and batch['tx'].x.is_cuda:
            return self._graphed_training_step(batch)
        return self._step(batch, "train")

    # ------------------------------------------------- whole-step hipGraph under Lightning (opt-in, not in the reference)
    def enable_graphed_training(self, enabled: bool = True, **trainer_kw) -> "LitISTEncoder":
        """Opt in BEFORE ``Trainer.fit``: every ``training_step`` becomes one hipGraph replay of the whole step (batch
        staging, forward, the three losses, backward, fused Adam: :class:`segger_amd.train_step_graph.GraphedTrainer`,
        1.3 ms instead of a host-bound 3-5 ms per default 1M-edge batch).  The module switches to Lightning's MANUAL
        optimisation (``automatic_optimization = False``): ``training_step`` drives the optimizer that
        ``configure_optimizers`` returned (now with device-side step counters) and Lightning's own backward / optimizer
        step / ``zero_grad`` are not run -- so gradient clipping, ``accumulate_grad_batches`` and precision plugins do
        not apply (the reference's ``Trainer(logger, max_epochs, reload_dataloaders_every_n_epochs, callbacks)``,
        cli/segment.py:400-405, uses none of them).  Single process: under data parallelism build the
        ``GraphedTrainer`` yourself with a ``dp.FlatGradBucket``.  ``trainer_kw``: ``granularity`` / ``max_buckets``."""
        self._graphed_kw = dict(trainer_kw) if enabled else None
        self._graphed_trainer = None
        self.automatic_optimization = not enabled
        return self

    def fast(self, enabled: bool = True, **trainer_kw) -> "LitISTEncoder":
        """THE switch from "drop-in" to "fast" (not in the reference): bf16 activation storage (fp32 accumulation, fp32
        master weights and Adam state) + every training step as one hipGraph replay (:meth:`enable_graphed_training`).
        The import swap alone keeps the reference's arithmetic width -- fp32 storage, eager steps: ``default_dropin`` in
        bench.py's line -- and this is the configuration the headline numbers are quoted on (``strong.graphed``).  Call it
        before ``Trainer.fit``, or set ``SEGGER_AMD_FAST=1`` in the environment (the CLI constructs the module from parsed
        arguments only, cli/segment.py:366-385).  ``fast(False)`` goes back.  Edge-AUROC of the two configurations agrees to
        1e-3 (bench.py ``auroc``; tests/test_gpu_fov.py)."""
        self.model.compute_dtype = torch.bfloat16 if enabled else torch.float32
        return self.enable_graphed_training(enabled, **trainer_kw)

    def _graphed_training_step(self, batch) -> Tensor:
        tr = self._graphed_trainer
        if tr is None:
            from .train_step_graph import GraphedTrainer
            opt = self.optimizers()
            if isinstance(opt, (list, tuple)):
                if len(opt) != 1:
                    raise RuntimeError("graphed training drives exactly one optimizer (configure_optimizers' Adam)")
                opt = opt[0]
            opt = getattr(opt, "optimizer", opt)             # Lightning hands out a LightningOptimizer wrapper
            if not all(g.get("capturable", False) for g in opt.param_groups):
                raise RuntimeError("graphed training needs a capturable optimizer: call enable_graphed_training() "
                                   "before the trainer calls configure_optimizers()")
            tr = self._graphed_trainer = GraphedTrainer(self, opt, **self._graphed_kw)
        out = tr.step(batch).clone()         # [loss_tx, loss_bd, loss_sg, loss]; the trainer's own tensor is overwritten by the next step
        self._advance_manual_step_progress()
        bs = getattr(batch, "num_graphs", 1)
        for i, name in enumerate(("loss_tx", "loss_bd", "loss_sg")):
            self.log(f"train:{name}", out[i], prog_bar=True, batch_size=bs)
        return out[3]

    def _advance_manual_step_progress(self) -> None:
        """The replay stepped the RAW optimizer inside the hipGraph.  Under Lightning's manual optimisation
        ``trainer.global_step`` is the count of completed ``LightningOptimizer.step`` calls
        (``fit_loop.epoch_loop.manual_optimization.optim_step_progress``), advanced by the hooks the loop installs on the
        wrapper (``_on_before_step`` / ``_on_after_step``).  Without them ``global_step`` stays 0 for the whole fit:
        ``ModelCheckpoint`` never saves, ``max_steps`` / ``every_n_train_steps`` / step-based ``val_check_interval`` never
        fire.  So the hooks are called here, once per replayed step -- the wrapper's if it has them, else the loop's
        progress tracker directly; a plain loop (the stand-in base, bench.py) has neither and nothing happens."""
        lopt = self.optimizers()
        if isinstance(lopt, (list, tuple)):
            lopt = lopt[0] if len(lopt) == 1 else None
        before, after = getattr(lopt, "_on_before_step", None), getattr(lopt, "_on_after_step", None)
        if callable(before) and callable(after):
            before()
            after()
            return
        loop = getattr(getattr(getattr(self.trainer, "fit_loop", None), "epoch_loop", None), "manual_optimization", None)
        prog = getattr(loop, "optim_step_progress", None)
        if prog is not None and hasattr(prog, "increment_ready") and hasattr(prog, "increment_completed"):
            prog.increment_ready()
            prog.increment_completed()

    def validation_step(self, batch, batch_idx: int) -> Tensor:
        return self._step(batch, "val")

    # ---------------------------------------------------------------- predict
    @torch.no_grad()
    def predict_step(self, batch, batch_idx: int, min_similarity: Optional[float] = None):
        """-> (tx_index, seg_idx, max_sim, gene_id) on the CPU, filtered by
        ``predict_mask`` (lightning_model.py:263-298)."""
        embeddings = self.forward(batch)
        z_tx, z_bd = embeddings['tx'], embeddings['bd']
        ei = batch[TX_NB_BD].edge_index
        n_tx = batch['tx'].num_nodes
        g = edge_graph(batch_cache(batch), TX_NB_BD, ei, n_tx, z_bd.shape[0], need_by_dst=False, validate="deferred")
        max_sim, _, seg_idx, _ = ops.edge_cos_argmax(
            g.by_src, z_tx, z_bd, dst_index=batch['bd']['index'], min_similarity=min_similarity)
        mask = batch['tx']['predict_mask']
        keep = mask.nonzero(as_tuple=True)[0]          # ONE compaction (one sync) for the four outputs, not one each
        out = (batch['tx']['index'][keep], seg_idx[keep], max_sim[keep], batch['tx']['x'][keep])
        out = tuple(t.cpu() for t in out)
        flush_validation()                 # the copies above synchronised: surface a bad edge_index of this batch now
        return out

    def configure_optimizers(self, capturable: bool = False) -> torch.optim.Optimizer:
        """Adam as in the reference (lightning_model.py:300-303).  On the GPU it is ``segger_amd.optim.Adam`` -- a
        ``torch.optim.Adam`` (same state and ``state_dict``) with device-side step counters whose ``step()`` runs the
        hand-written update kernel; ``capturable`` is kept for old call sites (always on there)."""
        params = list(self.parameters())
        fused = bool(params) and all(p.is_cuda for p in params)     # one multi-tensor kernel on the GPU
        if fused:
            # on the GPU: device-side step counters always, and the update on the hand-written kernel (optim.Adam is a
            # torch.optim.Adam: same state, same state_dict; falls through to torch's step for anything it does not cover)
            from .optim import Adam
            return Adam(params, lr=self.learning_rate, fused=True, capturable=True)
        return torch.optim.Adam(params, lr=self.learning_rate)
