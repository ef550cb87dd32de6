"""CPU oracle for the step AFTER the hot path -- TEST INFRASTRUCTURE ONLY.

Restates, in plain numpy with a Python loop per gene,

* ``ISTSegmentationWriter.assign_transcripts_to_cells`` reference ``src/segger/data/writer.py:131-259``:
  concatenate the ``predict_step`` 4-tuples, keep for every transcript ``row_index`` the row of highest
  similarity (``:186-190``), per-gene threshold ``min(threshold_yen, threshold_li)`` over the ASSIGNED
  transcripts (``:196-235``), median back-fill for genes whose Li iteration does not converge (``:238-241``);
* ``threshold_li_custom`` reference ``src/segger/data/utils/threshold.py:3-11`` (stop after ``max_iter`` callbacks);
* ``skimage.filters.threshold_yen`` / ``threshold_li`` (scikit-image 0.26.0, ``pixi.lock:156``; NOT installed in this
  image and absent from /root/reference): restated from the published algorithms (Yen et al. 1995: maximum
  correlation criterion on a 256-bin histogram; Li & Tam 1998: iterative minimum cross entropy on the raw values,
  start = mean, tolerance = half the smallest gap between distinct values).

PARITY UNPINNED: neither skimage nor polars can be imported here, and the reference holds no fixture for this
step.  Known, documented differences: the reference samples 10 M values (polars RNG) from larger genes (``:218-220``);
this restatement uses all values.  Ties in the per-transcript dedup are unordered in the reference (unstable
sort); here the row that comes first in the concatenation wins.
"""
import numpy as np


def threshold_yen(arr: np.ndarray, nbins: int = 256) -> float:
    a = np.asarray(arr, dtype=np.float64).ravel()
    lo, hi = a.min(), a.max()
    if lo == hi:                       # numpy widens a degenerate range by +-0.5
        lo, hi = lo - 0.5, hi + 0.5
    counts, edges = np.histogram(a, bins=nbins, range=(lo, hi))
    centers = (edges[:-1] + edges[1:]) / 2.0
    pmf = counts.astype(np.float64) / counts.sum()
    p1 = np.cumsum(pmf)
    p1_sq = np.cumsum(pmf ** 2)
    p2_sq = np.cumsum(pmf[::-1] ** 2)[::-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        crit = np.log(((p1_sq[:-1] * p2_sq[1:]) ** -1) * (p1[:-1] * (1.0 - p1[:-1])) ** 2)
    return float(centers[int(np.argmax(crit))])


def threshold_li(arr: np.ndarray, max_iter: int = 250) -> float:
    """Raises StopIteration like ``threshold_li_custom`` when more than ``max_iter`` callbacks fire
    (one for the initial guess, one per iteration)."""
    a = np.asarray(arr, dtype=np.float64).ravel()
    if np.all(a == a[0]):
        return float(a[0])
    a_min = a.min()
    a = a - a_min
    tol = np.min(np.diff(np.unique(a))) / 2.0
    t_next = a.mean()
    t_curr = -2.0 * tol
    calls = 1
    while abs(t_next - t_curr) > tol:
        t_curr = t_next
        fg = a > t_curr
        mean_fore = a[fg].mean()
        mean_back = a[~fg].mean()
        if mean_back == 0:
            break
        t_next = (mean_back - mean_fore) / (np.log(mean_back) - np.log(mean_fore))
        calls += 1
        if calls > max_iter:
            raise StopIteration
    return float(t_next + a_min)


def assign_transcripts_to_cells(predictions, max_iter: int = 250):
    """-> dict of numpy arrays: row_index (sorted, unique), cell_encoding (-1 = none), similarity, gene,
    similarity_threshold (nan where the gene has no assigned transcript)."""
    idx = np.concatenate([np.asarray(p[0]) for p in predictions]).astype(np.int64)
    seg = np.concatenate([np.asarray(p[1]) for p in predictions]).astype(np.int64)
    sim = np.concatenate([np.asarray(p[2]) for p in predictions]).astype(np.float32)
    gene = np.concatenate([np.asarray(p[3]) for p in predictions]).astype(np.int64)
    order = np.lexsort((np.arange(idx.size), -sim.astype(np.float64), idx))     # row_index asc, similarity desc
    first = np.ones(idx.size, dtype=bool)
    first[1:] = idx[order][1:] != idx[order][:-1]
    keep = order[first]
    idx, seg, sim, gene = idx[keep], seg[keep], sim[keep], gene[keep]
    thresholds, failed = {}, []
    for g in np.unique(gene[seg >= 0]):
        arr = sim[(gene == g) & (seg >= 0)]
        try:
            thresholds[int(g)] = min(threshold_yen(arr), threshold_li(arr, max_iter))
        except StopIteration:
            failed.append(int(g))
    glob = float(np.quantile(list(thresholds.values()), 0.5)) if thresholds else float("nan")
    for g in failed:
        thresholds[g] = glob
    thr = np.array([thresholds.get(int(g), np.nan) for g in gene], dtype=np.float64)
    return {"row_index": idx, "cell_encoding": seg, "similarity": sim, "gene": gene, "similarity_threshold": thr,
            "global_threshold": glob, "failed_genes": np.array(sorted(failed), dtype=np.int64)}
