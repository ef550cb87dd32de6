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
    _MIN_PROB = 1e-8

    @torch.no_grad()
    def __init__(self, cluster_similarity: Tensor):
        s = cluster_similarity.clone().float()
        s.fill_diagonal_(1)
        self.similarity = s.clamp_min(self._MIN_PROB)
        self.dissimilarity = (-s).clamp_min(self._MIN_PROB)

    @staticmethod
    def _cdf(m: Tensor) -> Tensor:
        c = torch.cumsum(m / m.sum(dim=1, keepdim=True), dim=1)
        c[:, -1] = 1.0
        return c

    @torch.no_grad()
    def build_index(self, labels: Tensor) -> dict:
        """Everything ``sample_triplets`` derives from the labels alone (reference ``_build_index``,
        triplet_loss.py:27-80).  Labels of a batch do not change between steps, so callers may cache it."""
        dev = labels.device
        labels = labels.long()
        sim, dis = self.similarity.to(dev), self.dissimilarity.to(dev)
        n_clusters = sim.shape[0]
        counts = torch.bincount(labels, minlength=n_clusters)
        offsets = counts.cumsum(0) - counts
        # default (unstable) argsort, the very call the reference makes (triplet_loss.py:41): the order of
        # equal labels is implementation-defined (torch's CPU and GPU sorts differ) and decides WHICH member of
        # the drawn cluster is returned; pass a ready-made ``index`` to pin it
        members = torch.argsort(labels)
        present = torch.nonzero(counts > 0).flatten()
        slot_of = torch.full((n_clusters,), -1, dtype=torch.long, device=dev)
        slot_of[present] = torch.arange(present.numel(), device=dev)
        row = slot_of[labels]
        return dict(labels=labels, counts=counts, offsets=offsets, members=members, present=present,
                    cdf_pos=self._cdf(sim[present][:, present])[row], cdf_neg=self._cdf(dis[present][:, present])[row],
                    dists=1.0 - sim)

    @torch.no_grad()
    def sample_triplets(self, labels: Tensor, uniforms: Optional[Tuple[Tensor, Tensor, Tensor, Tensor]] = None,
                        index: Optional[dict] = None):
        """-> positives, negatives, dists_pos, dists_neg (all [N]).  ``uniforms`` overrides the
        four U[0,1) draws (positive cluster, positive member, negative cluster, negative member)."""
        ix = index if index is not None else self.build_index(labels)
        dev = ix["labels"].device
        n = ix["labels"].numel()
        if uniforms is None:
            u = [torch.rand(n, device=dev) for _ in range(4)]
            # reference draw order: cluster(pos), member(pos), cluster(neg), member(neg)
        else:
            u = [t.to(dev) for t in uniforms]
        counts, offsets, members, present = ix["counts"], ix["offsets"], ix["members"], ix["present"]

        def draw(cdf_rows: Tensor, u_cluster: Tensor, u_member: Tensor) -> Tensor:
            k = torch.searchsorted(cdf_rows, u_cluster.unsqueeze(-1)).squeeze(-1)
            cl = present[k]
            within = (u_member * counts[cl].float()).floor().long()
            return members[offsets[cl] + within]

        positives = draw(ix["cdf_pos"], u[0], u[1])
        negatives = draw(ix["cdf_neg"], u[2], u[3])
        labels, dists = ix["labels"], ix["dists"]
        return positives, negatives, dists[labels, labels[positives]], dists[labels, labels[negatives]]


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
        lightning_model.py:158-161) without materialising the three gathered [n, C] matrices:
        on the GPU the fused triplet kernel gathers anchor / positive / negative rows itself.
        ``cache`` (per batch) keeps the mask's index list and the selector's label index across steps."""
        key = ("tx_triplet_index", mask.data_ptr(), labels.data_ptr(), int(mask.numel()))
        hit = cache.get(key) if cache is not None else None
        if hit is None:
            idx = mask.nonzero(as_tuple=False).squeeze(1)
            index = self.selector.build_index(labels[idx]) if idx.numel() else None
            hit = (idx, index)
            if cache is not None:
                cache[key] = hit
        idx, index = hit
        if idx.numel() == 0:
            return 0.0
        pos, neg, _, _ = self.selector.sample_triplets(labels[idx] if index is None else index["labels"], uniforms,
                                                       index=index)
        if embeddings.is_cuda:
            from . import ops
            return ops.triplet_edge_loss(embeddings, None, idx, idx[pos], idx[neg], self.margin, eps=self.eps)
        e = embeddings[idx].float()
        return super().forward(e, e[pos], e[neg])


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

    __call__ = forward
