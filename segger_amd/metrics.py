"""Edge-level metrics evaluated on the device (BASELINE.json: "edge-AUROC vs reference")."""
from __future__ import annotations

import torch
from torch import Tensor


def auroc(scores: Tensor, labels: Tensor) -> float:
    """Area under the ROC curve of ``scores`` against boolean ``labels`` (Mann-Whitney U with average
    ranks for ties, i.e. the trapezoidal ROC area).  Runs where the tensors live; ``nan`` when one class
    is empty."""
    if scores.shape != labels.shape or scores.dim() != 1:
        raise ValueError("auroc: scores and labels must be 1-D tensors of the same length")
    lab = labels.to(torch.bool)
    n = scores.numel()
    n_pos = int(lab.sum())
    n_neg = n - n_pos
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    s, order = torch.sort(scores.float())
    _, inverse, counts = torch.unique_consecutive(s, return_inverse=True, return_counts=True)
    end = counts.cumsum(0).double()                     # 1-based rank of the last member of each tie group
    avg_rank = end - (counts.double() - 1.0) * 0.5
    pos_sorted = lab[order]
    rank_sum = avg_rank[inverse[pos_sorted]].sum()
    u = rank_sum - n_pos * (n_pos + 1) / 2.0
    return float(u / (float(n_pos) * float(n_neg)))


def assignment_accuracy(seg_idx: Tensor, true_cell: Tensor, has_true_candidate: Tensor) -> float:
    """Fraction of transcripts, among those whose true nucleus is one of their candidates, that the arg-max
    assignment sends to it."""
    m = has_true_candidate.to(torch.bool)
    if int(m.sum()) == 0:
        return float("nan")
    return float((seg_idx[m] == true_cell[m]).double().mean())
