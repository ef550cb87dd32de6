"""CPU oracle for segger's GNN hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``segger_amd/`` imports it, and the product path raises when
its HIP library is missing instead of falling back to this code.

PARITY UNPINNED: the reference (dpeerlab/segger @ /root/reference) ships no
tests, fixtures or golden vectors for this path, and its arithmetic lives in
un-vendored wheels (torch_geometric 2.7.0, torch_scatter 2.1.2, pixi.lock:171-172)
that are not installed here.  The functions below restate, with plain torch
CPU ops, (i) the reference's own Python (each function cites the file:line it
follows, relative to /root/reference) and (ii) the published algorithms of the
PyG / torch_scatter operators it calls.  The one importable reference file,
``src/segger/models/triplet_loss.py``, pins :func:`FastTripletSelectorOracle`
through ``tests/golden/triplet_selector_*.npz`` (see tests/golden/make_golden.py).

All functions are dtype-generic: run them in float64 for the reference answer,
float32 to reproduce what PyG's CPU path would compute (that is what
``bench.py`` times as ``cpu_baseline``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor
import torch.nn.functional as F

TX_TX = ("tx", "neighbors", "tx")
TX_BD = ("tx", "belongs", "bd")
TX_NB_BD = ("tx", "neighbors", "bd")


def pyg_key(edge_type: Tuple[str, str, str]) -> str:
    """torch_geometric.nn.module_dict.ModuleDict key mangling (PyG 2.7.0)."""
    return "<" + "___".join(edge_type) + ">"


# --------------------------------------------------------------------------
# Positional embedding -- src/segger/models/ist_encoder.py:22-79
# --------------------------------------------------------------------------
def sinusoidal_embedding(x: Tensor, dim: int, max_period: float = 1000) -> Tensor:
    """ist_encoder.py:22-31.  cos|sin of x * exp(-ln(max_period) * i / half)."""
    half = dim // 2
    # the reference builds freqs in float32 and multiplies x.float(); keep the
    # float32 *frequencies* (they are constants of the model) but carry the
    # product in x's dtype so a float64 oracle is a true higher-precision answer
    freqs = torch.exp(
        -math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half
    ).to(x.dtype)
    args = x[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def normalize_positions(pos: Tensor, batch: Optional[Tensor]) -> Tensor:
    """ist_encoder.py:62-74.  Per-graph min/max normalisation.

    ``batch is None``: no epsilon (``:63-64``).  Otherwise ``(pos - min) /
    (max - min + 1e-8)`` with per-graph min/max; graphs with no nodes keep 0/0
    (``:67-73``) which never gets indexed.
    """
    if batch is None:
        pos = pos - pos.min(dim=0).values
        return pos / pos.max(dim=0).values
    nb = int(batch.max()) + 1 if batch.numel() else 0
    mins = torch.zeros((nb, 2), dtype=pos.dtype)
    maxs = torch.zeros((nb, 2), dtype=pos.dtype)
    for b in range(nb):
        m = batch == b
        if m.any():
            mins[b] = pos[m].min(dim=0).values
            maxs[b] = pos[m].max(dim=0).values
    return (pos - mins[batch]) / (maxs[batch] - mins[batch] + 1e-8)


def positional_2d_embed(pos, batch, w1, b1, w2, b2, freq_dim: int = 256) -> Tensor:
    """Positional2dEmbedder.forward, ist_encoder.py:57-79 (embed() default
    max_period=10000, ``:51``; MLP = Linear-SiLU-Linear per coordinate, ``:43-47``;
    flatten(-2) puts the x-coordinate block first, ``:78``)."""
    pos = normalize_positions(pos, batch)
    n = pos.shape[0]
    freq = sinusoidal_embedding(pos.flatten(), freq_dim, max_period=10000).reshape(n, 2, freq_dim)
    h = F.silu(freq @ w1.T + b1)
    h = h @ w2.T + b2
    return h.flatten(-2)


# --------------------------------------------------------------------------
# GATv2Conv (torch_geometric 2.7.0; constructed at ist_encoder.py:111-131)
# --------------------------------------------------------------------------
def segment_softmax(src: Tensor, index: Tensor, num_nodes: int) -> Tensor:
    """torch_geometric.utils.softmax: max-shifted exp, sum + 1e-16."""
    shape = (num_nodes,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    smax = torch.full(shape, float("-inf"), dtype=src.dtype).scatter_reduce(
        0, idx, src.detach(), reduce="amax", include_self=True)
    out = (src - smax.index_select(0, index)).exp()
    ssum = torch.zeros(shape, dtype=src.dtype).index_add_(0, index, out) + 1e-16
    return out / ssum.index_select(0, index)


def gatv2_conv(
    x_src: Tensor, x_dst: Tensor, edge_index: Tensor,
    lin_l_w: Tensor, lin_l_b: Tensor, lin_r_w: Tensor, lin_r_b: Tensor,
    att: Tensor, bias: Tensor, heads: int,
    negative_slope: float = 0.2,
    dropout_p: float = 0.0, dropout_keep: Optional[Tensor] = None,
    return_alpha: bool = False, storage_round: Optional[torch.dtype] = None,
):
    """GATv2Conv.forward / edge_update / message / aggregate('add'), concat=True,
    residual=False, add_self_loops=False, share_weights=False.

    ``edge_index[0]`` = source (``x_j = lin_l(x_src)[src]``), ``edge_index[1]`` =
    target (``x_i = lin_r(x_dst)[dst]``).  ``dropout_keep`` is an explicit
    ``[E, H]`` 0/1 mask (attention dropout, ist_encoder.py:116,123); the kept
    coefficients are scaled by ``1/(1-p)`` like ``F.dropout``.
    """
    H = heads
    C = lin_l_w.shape[0] // H
    src, dst = edge_index[0].long(), edge_index[1].long()
    n_dst = x_dst.shape[0]
    x_l = (x_src @ lin_l_w.T + lin_l_b).view(-1, H, C)
    x_r = (x_dst @ lin_r_w.T + lin_r_b).view(-1, H, C)
    if storage_round is not None:       # diagnostic only: the projections as a 16-bit store would keep them
        x_l, x_r = x_l.to(storage_round).to(x_src.dtype), x_r.to(storage_round).to(x_src.dtype)
    x_j = x_l.index_select(0, src)
    x_i = x_r.index_select(0, dst)
    e = (F.leaky_relu(x_i + x_j, negative_slope) * att.view(1, H, C)).sum(-1)
    alpha = segment_softmax(e, dst, n_dst)
    if dropout_keep is not None and dropout_p > 0.0:
        alpha = alpha * dropout_keep.to(alpha.dtype) / (1.0 - dropout_p)
    out = torch.zeros((n_dst, H, C), dtype=x_l.dtype).index_add_(0, dst, x_j * alpha.unsqueeze(-1))
    out = out.view(n_dst, H * C) + bias
    return (out, alpha) if return_alpha else out


# --------------------------------------------------------------------------
# ISTEncoder.forward -- src/segger/models/ist_encoder.py:289-333
# --------------------------------------------------------------------------
def ist_encoder_forward(
    sd: Dict[str, Tensor],
    x_dict: Dict[str, Tensor],
    edge_index_dict: Dict[Tuple[str, str, str], Tensor],
    pos_dict: Dict[str, Tensor],
    batch_dict: Dict[str, Optional[Tensor]],
    *,
    n_heads: int,
    prefix: str = "model.",
    use_positional_embeddings: bool = True,
    normalize_embeddings: bool = True,
    dropout_p: float = 0.0,
    dropout_keep: Optional[Dict[Tuple[int, Tuple[str, str, str]], Tensor]] = None,
    return_attention: bool = False,
    trace: Optional[dict] = None,
    storage_round: Optional[torch.dtype] = None,
):
    """Embed -> (+pos) -> GELU -> n x (HeteroConv(GATv2) -> GELU) -> lin_last -> L2.

    ``trace`` (diagnostics): filled with ``input`` / ``layer{i}`` (the activations after each GELU) and ``pre_norm``
    (lin_last's output before F.normalize).  ``storage_round`` (diagnostics): a model of THIS arithmetic with its
    activations kept in a 16-bit type -- every tensor a layer hands to the next (and the GATv2 projections) and every
    weight MATRIX (a GEMM operand) is rounded to that type and back; att, biases, accumulation and the softmax stay in
    ``sd``'s dtype.  It answers "how far does
    16-bit storage alone move the reference's own scores" (bench.py auroc.*.frac_over_atol, tools/bf16_miss.py).

    ``sd`` is a state dict with the reference's key names.
    HeteroConv(aggr='sum') runs the convs whose edge type is present in
    ``edge_index_dict`` and sums per destination type (ist_encoder.py:109-134);
    only tx-neighbors-tx and tx-belongs-bd exist in segger's data
    (data/utils/heterodata.py:138,147) so each sum has one term.
    """
    dt = sd[prefix + "lin_first.tx.weight"].dtype
    if storage_round is None:
        g = lambda k: sd[prefix + k]
    else:       # GEMM operands in the 16-bit type too (weight matrices; att / biases stay in fp32 as accumulators do)
        g = lambda k: (sd[prefix + k].to(storage_round).to(dt) if k.endswith(".weight") and sd[prefix + k].dim() == 2
                       else sd[prefix + k])
    x = {
        "tx": g("lin_first.tx.weight").index_select(0, x_dict["tx"].long()),   # :260,312
        "bd": x_dict["bd"].to(dt) @ g("lin_first.bd.weight").T + g("lin_first.bd.bias"),  # :261
    }
    if use_positional_embeddings:                                               # :314-318
        pe = lambda k: positional_2d_embed(
            pos_dict[k].to(dt), batch_dict.get(k),
            g("pos_emb.mlp.0.weight"), g("pos_emb.mlp.0.bias"),
            g("pos_emb.mlp.2.weight"), g("pos_emb.mlp.2.bias"))
        x = {k: torch.cat((v, pe(k)), -1) for k, v in x.items()}
    x = {k: F.gelu(v) for k, v in x.items()}                                    # :320
    rnd = (lambda v: v) if storage_round is None else (lambda v: v.to(storage_round).to(dt))
    x = {k: rnd(v) for k, v in x.items()}
    if trace is not None:
        trace["input"] = dict(x)

    n_layers = 0
    while (prefix + f"conv_layers.{n_layers}.conv.convs.{pyg_key(TX_TX)}.att") in sd:
        n_layers += 1
    attn = {}
    for li in range(n_layers):                                                  # :323-325
        outs: Dict[str, list] = {}
        for et in (TX_TX, TX_BD):
            if et not in edge_index_dict:
                continue
            p = f"conv_layers.{li}.conv.convs.{pyg_key(et)}."
            keep = None if dropout_keep is None else dropout_keep.get((li, et))
            o, a = gatv2_conv(
                x[et[0]], x[et[2]], edge_index_dict[et],
                g(p + "lin_l.weight"), g(p + "lin_l.bias"),
                g(p + "lin_r.weight"), g(p + "lin_r.bias"),
                g(p + "att").reshape(-1), g(p + "bias"), n_heads,
                dropout_p=dropout_p, dropout_keep=keep, return_alpha=True, storage_round=storage_round)
            outs.setdefault(et[2], []).append(o)
            attn[(li, et)] = a
        x = {k: rnd(F.gelu(torch.stack(v, 0).sum(0))) for k, v in outs.items()}
        if trace is not None:
            trace[f"layer{li}"] = dict(x)

    z = {k: x[k] @ g(f"lin_last.lins.{k}.weight").T + g(f"lin_last.lins.{k}.bias") for k in x}  # :328
    z = {k: rnd(v) for k, v in z.items()}
    if trace is not None:
        trace["pre_norm"] = dict(z)
    if normalize_embeddings:                                                    # :331-332
        z = {k: F.normalize(v, dim=-1) for k, v in z.items()}
    return (z, attn) if return_attention else z


# --------------------------------------------------------------------------
# Scoring head -- src/segger/models/lightning_model.py:263-298
# --------------------------------------------------------------------------
def scatter_max(src: Tensor, index: Tensor, dim_size: int):
    """torch_scatter.scatter_max (2.1.2) CPU semantics: strict '>' update so the
    first (lowest edge id) maximum wins; untouched rows -> (0, len(src))."""
    E = src.shape[0]
    out = torch.zeros(dim_size, dtype=src.dtype)
    arg = torch.full((dim_size,), E, dtype=torch.long)
    if E:
        smax = torch.full((dim_size,), float("-inf"), dtype=src.dtype).scatter_reduce(
            0, index, src, reduce="amax", include_self=True)
        is_max = src == smax.index_select(0, index)
        eid = torch.where(is_max, torch.arange(E), torch.full((E,), E))
        arg = arg.scatter_reduce(0, index, eid, reduce="amin", include_self=True)
        touched = arg < E
        out = torch.where(touched, smax, out)
    return out, arg


def edge_scores(z_tx: Tensor, z_bd: Tensor, edge_index: Tensor) -> Tensor:
    """lightning_model.py:275-279: torch.cosine_similarity (eps 1e-8) per edge."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    return torch.cosine_similarity(z_tx[src], z_bd[dst])


def predict_assign(z_tx, z_bd, edge_index, bd_index, min_similarity=None):
    """lightning_model.py:275-293 -> (seg_idx[Nt] (-1 = none), max_sim[Nt])."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    sim = edge_scores(z_tx, z_bd, edge_index)
    max_sim, max_idx = scatter_max(sim, src, z_tx.shape[0])
    valid = max_idx < dst.shape[0]
    if min_similarity is not None:
        valid = valid & (max_sim >= min_similarity)
    seg = torch.full_like(max_idx, -1)
    seg[valid] = bd_index.long()[dst[max_idx[valid]]]
    return seg, max_sim


def predict_step(sd, batch, *, n_heads, min_similarity=None, **enc_kw):
    """LitISTEncoder.predict_step, lightning_model.py:263-298."""
    z = ist_encoder_forward(sd, batch.x_dict, batch.edge_index_dict, batch.pos_dict,
                            batch.batch_dict, n_heads=n_heads, **enc_kw)
    seg, max_sim = predict_assign(z["tx"], z["bd"], batch[TX_NB_BD].edge_index,
                                  batch["bd"]["index"], min_similarity)
    m = batch["tx"]["predict_mask"]
    return batch["tx"]["index"][m], seg[m], max_sim[m], batch["tx"]["x"][m]


# --------------------------------------------------------------------------
# Training head -- src/segger/models/lightning_model.py:136-213
# --------------------------------------------------------------------------
def triplet_margin_loss(a, p, n, margin: float, eps: float = 1e-6) -> Tensor:
    """torch.nn.TripletMarginLoss(p=2, eps=1e-6, swap=False, 'mean'):
    d(x,y) = ||x - y + eps||_2 ; mean(max(d_ap - d_an + margin, 0))."""
    d_ap = (a - p + eps).pow(2).sum(-1).sqrt()
    d_an = (a - n + eps).pow(2).sum(-1).sqrt()
    return (d_ap - d_an + margin).clamp_min(0).mean()


def segmentation_loss(z_tx, z_bd, edge_index, dst_neg, loss_type="triplet", margin=0.4):
    """lightning_model.py:167-207 with the sampled negatives ``dst_neg`` given
    (``(dst + randint(1, Nb)) % Nb``, ``:178-180``)."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    if z_bd.shape[0] <= 1:                                                       # :173-175
        return torch.zeros((), dtype=z_tx.dtype)
    if loss_type == "triplet":                                                   # :182-187
        return triplet_margin_loss(z_tx[src], z_bd[dst], z_bd[dst_neg.long()], margin)
    # BCE on dot-product logits, positives then negatives                        # :190-207
    s2 = torch.cat([src, src])
    d2 = torch.cat([dst, dst_neg.long()])
    logits = (z_tx[s2] * z_bd[d2]).sum(-1)
    labels = torch.cat([torch.ones(src.numel()), torch.zeros(src.numel())]).to(logits.dtype)
    return F.binary_cross_entropy_with_logits(logits, labels)


def scheduled_weights(w_start: Tensor, w_end: Tensor, epoch: int, trainer_max_epochs: int,
                      normalize: bool = True) -> Tensor:
    """lightning_model.py:136-149: cosine ramp, then w /= w.sum() + 1e-8."""
    max_epochs = max(1, trainer_max_epochs - 1)
    t = min(epoch, max_epochs) / max_epochs
    alpha = 0.5 * (1.0 + math.cos(math.pi * t))
    w = w_end + (w_start - w_end) * alpha
    if normalize:
        w = w / (w.sum() + 1e-8)
    return w


# --------------------------------------------------------------------------
# Metric losses -- src/segger/models/triplet_loss.py:8-204
# --------------------------------------------------------------------------
class FastTripletSelectorOracle:
    """triplet_loss.py:8-125.  Cluster-similarity-weighted positive / negative
    sampling through per-cluster CDFs + searchsorted.  The four uniform draws
    (``:95,101,106,112``) are explicit inputs so the sampling is a pure function."""

    def __init__(self, cluster_similarity: Tensor):
        s = cluster_similarity.clone()
        s.fill_diagonal_(1)                                    # :22
        self.similarity = s.clamp_min(1e-8)                    # :23
        self.dissimilarity = (-s).clamp_min(1e-8)              # :24

    def sample(self, labels: Tensor, u_pos, u_pos2, u_neg, u_neg2):
        C = self.similarity.shape[0]
        labels = labels.long()
        counts = torch.bincount(labels, minlength=C)
        offsets = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])[:-1]
        sorted_idx = torch.argsort(labels)      # unstable, as the reference (:41): order inside a cluster is implementation-defined
        present = torch.nonzero(counts > 0).flatten()
        def cdf(m):
            m = m[present][:, present]
            c = torch.cumsum(m / m.sum(1, keepdim=True), 1)
            c[:, -1] = 1.0
            return c
        cdf_neg, cdf_pos = cdf(self.dissimilarity), cdf(self.similarity)
        present_idx = -torch.ones(C, dtype=torch.long)
        present_idx[present] = torch.arange(present.numel())
        pres = present_idx[labels]

        def draw(cdf_m, u, u2):
            k = torch.searchsorted(cdf_m[pres], u.unsqueeze(-1)).squeeze(-1)
            clust = present[k]
            within = (u2 * counts[clust].float()).floor().long()
            return sorted_idx[offsets[clust] + within]
        positives = draw(cdf_pos, u_pos, u_pos2)
        negatives = draw(cdf_neg, u_neg, u_neg2)
        dists = 1.0 - self.similarity
        return positives, negatives, dists[labels, labels[positives]], dists[labels, labels[negatives]]


def tx_triplet_loss(z, positives, negatives, margin):
    """TripletLoss.forward, triplet_loss.py:144-160 (given sampled triplets)."""
    if z.shape[0] == 0:
        return torch.zeros((), dtype=z.dtype)
    return triplet_margin_loss(z, z[positives], z[negatives], margin)


def bd_metric_loss(z, positives, negatives, dists_pos, dists_neg):
    """MetricLoss.forward, triplet_loss.py:176-204."""
    if z.shape[0] == 0:
        return torch.zeros((), dtype=z.dtype)
    cp = torch.cosine_similarity(z, z[positives])
    cn = torch.cosine_similarity(z, z[negatives])
    return F.mse_loss(cp, 1 - dists_pos.to(z.dtype)) + F.mse_loss(cn, 1 - dists_neg.to(z.dtype))


# --------------------------------------------------------------------------
# Attention-dropout mask: restates the counter-based generator the HIP kernels
# use (include/segger_amd.h, "dropout"), so that training-mode parity can be
# checked with an explicit mask.  Not part of the reference (its F.dropout
# stream cannot be reproduced by any other implementation).
# --------------------------------------------------------------------------
def _splitmix64(z: int) -> int:
    m = 0xFFFFFFFFFFFFFFFF
    z = (z + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def dropout_keep_mask(seed: int, n_edges: int, heads: int, p: float) -> Tensor:
    """keep[e,h] for ORIGINAL (COO) edge id e, as include/segger_amd.h defines it:
    (lo, hi) = halves of splitmix64(seed); x = (e*H + h) ^ lo; x *= 0x85ebca6b; x ^= x>>13;
    x *= 0xc2b2ae35; x ^= x>>16; x ^= hi; keep = (x >> 8) >= floor(p * 2^24)   (uint32)."""
    import numpy as np
    s = _splitmix64(seed & 0xFFFFFFFFFFFFFFFF)
    lo, hi = np.uint32(s & 0xFFFFFFFF), np.uint32(s >> 32)
    c = np.arange(n_edges * heads, dtype=np.uint64).astype(np.uint32)
    with np.errstate(over="ignore"):
        x = c ^ lo
        x = x * np.uint32(0x85EBCA6B)
        x = x ^ (x >> np.uint32(13))
        x = x * np.uint32(0xC2B2AE35)
        x = x ^ (x >> np.uint32(16))
        x = x ^ hi
    thr = np.uint32(int(p * float(1 << 24)))
    return torch.from_numpy(((x >> np.uint32(8)) >= thr).reshape(n_edges, heads))


# --------------------------------------------------------------------------
# Ranking metric used by BASELINE.json ("AUROC vs ref")
# --------------------------------------------------------------------------
def auroc(scores: Tensor, labels: Tensor) -> float:
    """Mann-Whitney AUROC with average ranks for ties (float64)."""
    s = scores.double()
    order = torch.argsort(s, stable=True)
    ss = s[order]
    n = s.numel()
    ranks = torch.empty(n, dtype=torch.float64)
    # average ranks over tie groups
    _, inv, cnt = torch.unique_consecutive(ss, return_inverse=True, return_counts=True)
    ends = cnt.cumsum(0).double()
    starts = ends - cnt.double() + 1
    ranks[order] = ((starts + ends) / 2)[inv]
    pos = labels.bool()
    n1, n0 = int(pos.sum()), int((~pos).sum())
    if n1 == 0 or n0 == 0:
        return float("nan")
    return float((ranks[pos].sum() - n1 * (n1 + 1) / 2) / (n1 * n0))
