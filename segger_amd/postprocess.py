"""What consumes the ``predict_step`` 4-tuples: the final transcript -> cell table.

Restates ``ISTSegmentationWriter.assign_transcripts_to_cells`` (reference ``src/segger/data/writer.py:131-259``)
as segmented tensor operations that run wherever the tensors live (on the MI355X for a 100 M-transcript slide:
two stable sorts, one 2-D histogram, one vectorised fixed-point iteration -- no per-gene Python loop, no polars):

* best row per transcript over overlapping prediction tiles (``:186-190``: sort by row_index, similarity
  descending, keep the first);
* per-gene similarity threshold over ASSIGNED transcripts = ``min(threshold_yen, threshold_li)`` (``:196-235``):
  Yen's maximum-correlation criterion on a 256-bin histogram and Li's iterative minimum cross entropy (the
  algorithms of scikit-image 0.26 the reference calls; ``threshold_li_custom``, ``data/utils/threshold.py:3-11``,
  gives up after 250 callbacks), all genes at once;
* genes whose Li iteration does not converge take the median of the converged thresholds (``:238-241``).

Differences from the reference, both documented in oracle/postprocess_oracle.py: no 10 M-value subsampling of
large genes (every value is used), and similarity ties in the dedup go to the first row of the concatenation.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
from torch import Tensor

N_BINS = 256


def best_assignment(predictions: Sequence[Sequence[Tensor]], device=None) -> Dict[str, Tensor]:
    """Concatenate ``(tx_index, seg_idx, max_sim, gene_id)`` tuples and keep, per ``tx_index``, the row of
    highest similarity.  Rows come back sorted by ``row_index``."""
    if len(predictions) == 0:
        raise ValueError("no predictions")
    dev = torch.device(device) if device is not None else predictions[0][0].device
    idx = torch.cat([p[0].to(dev) for p in predictions]).long()
    seg = torch.cat([p[1].to(dev) for p in predictions]).long()
    sim = torch.cat([p[2].to(dev) for p in predictions]).float()
    gene = torch.cat([p[3].to(dev) for p in predictions]).long()
    o1 = torch.sort(sim, descending=True, stable=True).indices
    o2 = torch.sort(idx[o1], stable=True).indices
    order = o1[o2]
    si = idx[order]
    first = torch.ones_like(si, dtype=torch.bool)
    first[1:] = si[1:] != si[:-1]
    keep = order[first]
    return {"row_index": idx[keep], "cell_encoding": seg[keep], "similarity": sim[keep], "gene": gene[keep]}


def _edges(i: Tensor, lo: Tensor, hi: Tensor, step: Tensor) -> Tensor:
    """numpy.linspace(lo, hi, N_BINS + 1)[i]: lo + i * step, with the last edge pinned to hi."""
    return torch.where(i == N_BINS, hi, lo + i.double() * step)


def per_gene_thresholds(similarity: Tensor, gene: Tensor, assigned: Tensor, max_iter: int = 250):
    """-> (genes [G] sorted, threshold [G] f64, converged [G] bool, global_threshold float)."""
    v = similarity[assigned].double()
    genes, inv, counts = torch.unique(gene[assigned], return_inverse=True, return_counts=True)
    ng = int(genes.numel())
    dev = v.device
    if ng == 0:
        return genes, v.new_zeros(0), torch.zeros(0, dtype=torch.bool, device=dev), float("nan")
    o1 = torch.sort(v, stable=True).indices
    o2 = torch.sort(inv[o1], stable=True).indices
    order = o1[o2]
    vs, gs = v[order], inv[order]                          # grouped by gene, ascending inside a gene
    ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    beg, end = ptr[:-1], ptr[1:]
    vmin, vmax = vs[beg], vs[end - 1]
    flat = vmin == vmax

    # ---- Yen: 256-bin histogram over [min, max] per gene (numpy.histogram's uniform-bin indexing) ---------
    lo = torch.where(flat, vmin - 0.5, vmin)
    hi = torch.where(flat, vmax + 0.5, vmax)
    step = (hi - lo) / N_BINS
    glo, ghi, gstep = lo[gs], hi[gs], step[gs]
    ind = (((vs - glo) / (ghi - glo)) * N_BINS).long()
    ind = torch.where(ind == N_BINS, ind - 1, ind)
    ind = ind - (vs < _edges(ind, glo, ghi, gstep)).long()
    ind = ind + ((vs >= _edges(ind + 1, glo, ghi, gstep)) & (ind != N_BINS - 1)).long()
    hist = torch.bincount(gs * N_BINS + ind, minlength=ng * N_BINS).view(ng, N_BINS).double()
    pmf = hist / hist.sum(1, keepdim=True)
    p1 = pmf.cumsum(1)
    p1_sq = (pmf * pmf).cumsum(1)
    p2_sq = (pmf * pmf).flip(1).cumsum(1).flip(1)
    crit = torch.log((p1_sq[:, :-1] * p2_sq[:, 1:]).reciprocal() * (p1[:, :-1] * (1.0 - p1[:, :-1])) ** 2)
    crit = torch.where(torch.isnan(crit), torch.full_like(crit, float("inf")), crit)     # numpy.argmax: nan wins
    k = crit.argmax(1)
    yen = (_edges(k, lo, hi, step) + _edges(k + 1, lo, hi, step)) / 2.0

    # ---- Li: t <- (mean_back - mean_fore) / (log mean_back - log mean_fore) on values shifted to min 0 -----
    a = vs - vmin[gs]
    d = a[1:] - a[:-1]
    ok = (gs[1:] == gs[:-1]) & (d > 0)
    gap = torch.full((ng,), float("inf"), dtype=torch.float64, device=dev)
    gap = gap.scatter_reduce(0, gs[1:][ok], d[ok], reduce="amin", include_self=True)
    tol = gap / 2.0
    cs = torch.cat([a.new_zeros(1), a.cumsum(0)])
    n_all = counts.double()
    t_next = (cs[end] - cs[beg]) / n_all
    t_curr = -2.0 * tol
    key = gs.double() * 4.0 + a                           # a in [0, 2]: one ascending key over all genes
    base = torch.arange(ng, device=dev, dtype=torch.float64) * 4.0
    calls = torch.ones(ng, dtype=torch.long, device=dev)
    failed = torch.zeros(ng, dtype=torch.bool, device=dev)
    active = ~flat & ((t_next - t_curr).abs() > tol)
    for _ in range(max_iter):
        if not bool(active.any()):
            break
        t_curr = torch.where(active, t_next, t_curr)
        pos = torch.searchsorted(key, base + t_curr, right=True)
        pos = torch.minimum(torch.maximum(pos, beg), end)
        n_back = (pos - beg).double()
        mean_back = (cs[pos] - cs[beg]) / n_back
        mean_fore = (cs[end] - cs[pos]) / (n_all - n_back)
        upd = active & ~(mean_back == 0)
        new_t = (mean_back - mean_fore) / (torch.log(mean_back) - torch.log(mean_fore))
        t_next = torch.where(upd, new_t, t_next)
        calls = calls + upd.long()
        fail = upd & (calls > max_iter)
        failed |= fail
        active = upd & ~fail & ((t_next - t_curr).abs() > tol)
    li = torch.where(flat, vmin, t_next + vmin)

    thr = torch.where(li < yen, li, yen)                   # python min(tye, tli): tye unless tli is smaller
    converged = ~failed
    glob = float(torch.quantile(thr[converged], 0.5)) if bool(converged.any()) else float("nan")
    thr = torch.where(converged, thr, torch.full_like(thr, glob))
    return genes, thr, converged, glob


def assign_transcripts_to_cells(predictions: Sequence[Sequence[Tensor]], device=None,
                                max_iter: int = 250) -> Dict[str, Tensor]:
    """-> ``row_index`` (unique, ascending), ``cell_encoding`` (-1 = unassigned), ``similarity``, ``gene``,
    ``similarity_threshold`` (nan for genes without an assigned transcript), plus ``global_threshold`` and
    ``failed_genes``.  A transcript counts as segmented when ``similarity >= similarity_threshold``
    (``writer.py:96-99``)."""
    out = best_assignment(predictions, device)
    genes, thr, converged, glob = per_gene_thresholds(out["similarity"], out["gene"], out["cell_encoding"] >= 0, max_iter)
    t = torch.full(out["gene"].shape, float("nan"), dtype=torch.float64, device=out["gene"].device)
    if genes.numel():
        j = torch.searchsorted(genes, out["gene"]).clamp_(max=genes.numel() - 1)
        hit = genes[j] == out["gene"]
        t = torch.where(hit, thr[j], t)
    out["similarity_threshold"] = t
    out["global_threshold"] = glob
    out["failed_genes"] = genes[~converged]
    return out


def to_frame(result: Dict[str, Tensor], obs=None, cell_id: str = "cell_id", cell_encoding: str = "cell_encoding"):
    """The columns of ``segger_segmentation.parquet`` as a pandas frame: ``row_index``, ``segger_cell_id``
    (from ``obs[[cell_id, cell_encoding]]``; the integer encoding itself when ``obs`` is None), ``segger_similarity``,
    ``similarity_threshold``."""
    import numpy as np
    import pandas as pd
    enc = result["cell_encoding"].cpu().numpy()
    if obs is not None:
        lut = pd.Series(obs[cell_id].to_numpy(), index=obs[cell_encoding].to_numpy().astype(np.int64))
        ids = lut.reindex(enc).to_numpy()
    else:
        ids = pd.array(np.where(enc >= 0, enc, 0), dtype="Int64")
        ids[enc < 0] = pd.NA
    return pd.DataFrame({
        "row_index": result["row_index"].cpu().numpy(),
        "segger_cell_id": ids,
        "segger_similarity": result["similarity"].cpu().numpy(),
        "similarity_threshold": result["similarity_threshold"].cpu().numpy(),
    })
