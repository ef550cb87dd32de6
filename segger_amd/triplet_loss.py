"""Cluster-aware metric losses on embeddings (``loss_tx`` / ``loss_bd``).

Restates reference ``src/segger/models/triplet_loss.py``: positives / negatives
are sampled per anchor from clusters drawn with probability proportional to the
cluster similarity / dissimilarity (``FastTripletSelector``, ``:8-125``), then a
triplet margin loss (``TripletLoss``, ``:128-160``) or a cosine-vs-distance MSE
(``MetricLoss``, ``:163-204``) is taken.  All tensors stay on the embeddings'
device; sampling uses torch's device RNG exactly where the reference does
(four ``torch.rand`` draws per call, in the same order).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.nn import functional as F


class FastTripletSelector:
    """``sample_triplets`` without a single device -> host synchronisation (no ``nonzero`` / ``bincount`` / boolean
    indexing, whose output sizes live on the device): the host can queue a whole training step ahead of the GPU,
    which is what segger's default 1 M-edge batches need (a step is ~4 ms of device work).

    Absent clusters stay in the tables with probability zero instead of being compacted away (reference
    ``_build_index``, triplet_loss.py:27-80): a cumulative distribution is unchanged by zero-probability entries, so
    the drawn clusters are the same.  A ``mask`` restricts anchors AND candidates to the masked nodes, which is what
    the reference gets by calling the loss on ``embeddings[mask], labels[mask]`` (lightning_model.py:158-165):
    unmasked nodes go to a dummy cluster nobody draws from, and their own triplets come back as ``-1``."""
    _MIN_PROB = 1e-8

    @torch.no_grad()
    def __init__(self, cluster_similarity: Tensor):
        s = cluster_similarity.clone().float()
        s.fill_diagonal_(1)
        self.similarity = s.clamp_min(self._MIN_PROB)
        self.dissimilarity = (-s).clamp_min(self._MIN_PROB)

    @staticmethod
    def _cdf(m: Tensor, present: Tensor) -> Tensor:
        k = m.shape[1]
        cols = torch.arange(k, device=m.device)
        p = m * present.to(m.dtype)[None, :]
        c = torch.cumsum(p / p.sum(dim=1, keepdim=True), dim=1)
        last = (cols * present.long()).max()                 # the reference pins its last (present) column to 1
        return torch.where(cols[None, :] >= last, torch.ones((), dtype=c.dtype, device=c.device), c)

    @torch.no_grad()
    def build_index(self, labels: Tensor, mask: Optional[Tensor] = None) -> dict:
        """Everything ``sample_triplets`` derives from the labels (and the mask) alone.  Labels of a batch do not change
        between steps, so callers may cache it."""
        dev = labels.device
        sim, dis = self.similarity.to(dev), self.dissimilarity.to(dev)
        n_clusters = sim.shape[0]
        labels = labels.long()
        lab = labels if mask is None else torch.where(mask, labels, torch.full_like(labels, n_clusters))
        counts = torch.zeros(n_clusters + 1, dtype=torch.long, device=dev).index_add_(0, lab, torch.ones_like(lab))
        offsets = counts.cumsum(0) - counts
        # default (unstable) argsort, the very call the reference makes (triplet_loss.py:41): the order of
        # equal labels is implementation-defined (torch's CPU and GPU sorts differ) and decides WHICH member of
        # the drawn cluster is returned; pass a ready-made ``index`` to pin it
        members = torch.argsort(lab)
        present = counts[:n_clusters] > 0
        # per-cluster tables; the torch path below expands them to per-node rows on first use (``_rows``)
        return dict(labels=labels, lab=lab, counts=counts, offsets=offsets, members=members, present=present,
                    cdf_pos_t=self._cdf(sim, present).contiguous(), cdf_neg_t=self._cdf(dis, present).contiguous(),
                    dists=(1.0 - sim).contiguous(), mask=mask, n_clusters=n_clusters)

    @staticmethod
    def _rows(ix: dict, which: str) -> Tensor:
        if which not in ix:
            row = ix["lab"].clamp(max=ix["n_clusters"] - 1)  # dummy-cluster anchors: any row, their triplets are dropped
            ix[which] = ix[which + "_t"][row]
        return ix[which]

    @torch.no_grad()
    def sample_triplets(self, labels: Tensor, uniforms: Optional[Tuple[Tensor, Tensor, Tensor, Tensor]] = None,
                        index: Optional[dict] = None, mask: Optional[Tensor] = None, device_seed=None):
        """-> positives, negatives, dists_pos, dists_neg (all [N]).  ``uniforms`` overrides the
        four U[0,1) draws (positive cluster, positive member, negative cluster, negative member).
        With a ``mask`` (given here or baked into ``index``), rows outside it hold ``-1``."""
        ix = index if index is not None else self.build_index(labels, mask)
        mask = ix.get("mask")
        dev = ix["labels"].device
        n = ix["labels"].numel()
        if dev.type == "cuda":                               # one fused kernel (segger_triplet_sample)
            from . import ops
            if device_seed is not None:                      # (constant, device word): fresh draws on every graph replay
                return ops.triplet_sample(ix, uniforms, seed=device_seed[0], seed_dev=device_seed[1])
            return ops.triplet_sample(ix, uniforms)
        if uniforms is None:
            u = [torch.rand(n, device=dev) for _ in range(4)]
            # reference draw order: cluster(pos), member(pos), cluster(neg), member(neg)
        else:
            u = [t.to(dev) for t in uniforms]
        counts, offsets, members = ix["counts"], ix["offsets"], ix["members"]
        k_max = counts.numel() - 2

        def draw(cdf_rows: Tensor, u_cluster: Tensor, u_member: Tensor) -> Tensor:
            cl = torch.searchsorted(cdf_rows, u_cluster.unsqueeze(-1)).squeeze(-1).clamp_(max=k_max)
            within = (u_member * counts[cl].float()).floor().long()
            return members[(offsets[cl] + within).clamp_(max=max(n - 1, 0))]

        positives = draw(self._rows(ix, "cdf_pos"), u[0], u[1])
        negatives = draw(self._rows(ix, "cdf_neg"), u[2], u[3])
        labels, dists = ix["labels"], ix["dists"]
        row = ix["lab"].clamp(max=k_max)
        d_pos, d_neg = dists[row, labels[positives].clamp(0, k_max)], dists[row, labels[negatives].clamp(0, k_max)]
        if mask is not None:
            minus = torch.full_like(positives, -1)
            positives, negatives = torch.where(mask, positives, minus), torch.where(mask, negatives, minus)
        return positives, negatives, d_pos, d_neg


def _masked_count(mask: Tensor) -> Tensor:
    return mask.sum().clamp_(min=1).float()


def _cached_index(selector: FastTripletSelector, name: str, labels: Tensor, mask: Tensor, cache: Optional[dict]) -> dict:
    """The selector's index of a batch, kept (i) in the batch's own cache, keyed by the tensors it was built from, or
    (ii) in the store a tile partition attaches to the batches of one tile set (``cache["persistent"]``: labels and
    masks of a tile never change, so the index survives from epoch to epoch)."""
    if cache is None:
        return selector.build_index(labels, mask)
    store = cache.get("persistent")
    key = name if store is not None else (name, mask.data_ptr(), labels.data_ptr(), int(labels.numel()))
    if store is None:
        store = cache
    index = store.get(key)
    if index is None:
        index = store[key] = selector.build_index(labels, mask)
    return index


class TripletLoss(torch.nn.TripletMarginLoss):
    def __init__(self, cluster_similarity: Tensor, margin: float = 1.0, **kwargs):
        super().__init__(margin=margin, **kwargs)
        self.selector = FastTripletSelector(cluster_similarity)

    def forward(self, embeddings: Tensor, labels: Tensor, uniforms=None):  # type: ignore[override]
        """``uniforms`` (not in the reference): the four U[0,1) draws of the selector, for replaying given vectors."""
        if labels.numel() == 0:
            return 0.0
        pos, neg, _, _ = self.selector.sample_triplets(labels, uniforms)
        if embeddings.is_cuda:
            from . import ops
            idx = torch.arange(labels.numel(), device=embeddings.device)
            return ops.triplet_edge_loss(embeddings, None, idx, pos, neg, self.margin, eps=self.eps)
        e = embeddings.float()
        return super().forward(e, e[pos], e[neg])

    def forward_masked(self, embeddings: Tensor, labels: Tensor, mask: Tensor, cache: Optional[dict] = None,
                       uniforms=None):
        """``forward(embeddings[mask], labels[mask])`` (how LitISTEncoder.get_losses calls it,
        lightning_model.py:158-161) without compacting anything: the selector works under the mask, the fused
        triplet kernel gathers anchor / positive / negative rows itself and skips the ``-1`` triplets of unmasked
        nodes, and the mean is rescaled from all nodes to the masked ones on the device.  ``uniforms`` are per NODE
        here (the reference draws one set per masked node).  ``cache`` (per batch) keeps the selector's index."""
        n = labels.numel()
        if n == 0:
            return 0.0
        index = _cached_index(self.selector, "tx_triplet_index", labels, mask, cache)
        if "anchors" not in index:
            index["anchors"] = torch.arange(n, device=labels.device)
            index["rescale"] = float(n) / _masked_count(mask)
        pos, neg, _, _ = self.selector.sample_triplets(labels, uniforms, index=index)
        if embeddings.is_cuda:
            from . import ops
            loss = ops.triplet_edge_loss(embeddings, None, index["anchors"], pos, neg, self.margin, eps=self.eps)
            return loss * index["rescale"]
        idx = mask.nonzero(as_tuple=False).squeeze(1)
        if idx.numel() == 0:
            return 0.0
        e = embeddings.float()
        return super().forward(e[idx], e[pos[idx]], e[neg[idx]])


class MetricLoss:
    def __init__(self, cluster_similarity: Tensor):
        self.selector = FastTripletSelector(cluster_similarity)

    def forward(self, embeddings: Tensor, labels: Tensor, uniforms=None):
        if labels.numel() == 0:
            return 0.0
        pos, neg, d_pos, d_neg = self.selector.sample_triplets(labels, uniforms)
        e = embeddings.float()
        cos_pos = torch.cosine_similarity(e, e[pos])
        cos_neg = torch.cosine_similarity(e, e[neg])
        return (F.mse_loss(cos_pos, 1 - d_pos.float(), reduction="mean")
                + F.mse_loss(cos_neg, 1 - d_neg.float(), reduction="mean"))

    def forward_masked(self, embeddings: Tensor, labels: Tensor, mask: Tensor, uniforms=None,
                       cache: Optional[dict] = None):
        """``forward(embeddings[mask], labels[mask])`` (lightning_model.py:162-165) with the mask applied as a weight:
        no boolean indexing, hence no device -> host synchronisation."""
        if labels.numel() == 0:
            return 0.0
        index = _cached_index(self.selector, "bd_metric_index", labels, mask, cache)
        if "weight" not in index:
            index["weight"] = mask.float() / _masked_count(mask)
        pos, neg, d_pos, d_neg = self.selector.sample_triplets(labels, uniforms, index=index)
        if embeddings.is_cuda:
            from . import ops
            return ops.metric_loss(embeddings, pos, neg, d_pos, d_neg, index["weight"])
        e = embeddings.float()
        w = index["weight"]
        cos_pos = torch.cosine_similarity(e, e[pos.clamp(min=0)])
        cos_neg = torch.cosine_similarity(e, e[neg.clamp(min=0)])
        return ((w * (cos_pos - (1 - d_pos.float())) ** 2).sum() + (w * (cos_neg - (1 - d_neg.float())) ** 2).sum())

    __call__ = forward
