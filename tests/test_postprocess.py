"""Prediction post-processing: the vectorised product (segger_amd/postprocess.py) against
the per-gene numpy oracle (oracle/postprocess_oracle.py) and hand-derived known answers.
Thresholds are float64 results of float sums taken in different orders: atol 1e-9."""
import numpy as np
import pytest
import torch

from segger_amd import postprocess as pp


@pytest.fixture(scope="module")
def po():
    import postprocess_oracle
    return postprocess_oracle


def fake_predictions(seed, n_tx=4000, n_genes=12, n_batches=5, overlap=0.3):
    g = torch.Generator().manual_seed(seed)
    gene_of = torch.randint(0, n_genes, (n_tx,), generator=g)
    preds = []
    for b in range(n_batches):
        m = torch.rand(n_tx, generator=g) < (1.0 / n_batches + overlap)
        idx = m.nonzero().squeeze(1)
        idx = idx[torch.randperm(idx.numel(), generator=g)]
        # bimodal similarities (assigned-well vs noise), some unassigned, some exact duplicates
        hi = torch.rand(idx.numel(), generator=g) < 0.6
        sim = torch.where(hi, 0.7 + 0.1 * torch.randn(idx.numel(), generator=g), 0.1 + 0.15 * torch.randn(idx.numel(), generator=g))
        sim = (sim.clamp(-1, 1) * 64).round() / 64 if b == 0 else sim.clamp(-1, 1)          # ties in batch 0
        seg = torch.randint(0, 50, (idx.numel(),), generator=g)
        none = torch.rand(idx.numel(), generator=g) < 0.1
        seg = torch.where(none, torch.full_like(seg, -1), seg)
        sim = torch.where(none, torch.zeros_like(sim), sim)
        preds.append((idx, seg, sim.float(), gene_of[idx].int()))
    return preds


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_matches_oracle(po, seed):
    preds = fake_predictions(seed)
    got = pp.assign_transcripts_to_cells(preds)
    ref = po.assign_transcripts_to_cells([[t.numpy() for t in p] for p in preds])
    assert np.array_equal(got["row_index"].numpy(), ref["row_index"])
    assert np.array_equal(got["cell_encoding"].numpy(), ref["cell_encoding"])
    assert np.array_equal(got["similarity"].numpy(), ref["similarity"])
    assert np.array_equal(got["gene"].numpy(), ref["gene"])
    assert np.allclose(got["similarity_threshold"].numpy(), ref["similarity_threshold"], atol=1e-9, equal_nan=True)
    assert len(np.unique(ref["row_index"])) == len(ref["row_index"])
    # the kept row is a maximum-similarity row of its transcript
    allidx = torch.cat([p[0] for p in preds]); allsim = torch.cat([p[2] for p in preds])
    best = torch.full((4000,), -2.0).scatter_reduce(0, allidx, allsim, reduce="amax")
    assert torch.equal(best[got["row_index"]], got["similarity"])


def test_nonconverging_genes_take_the_median(po):
    preds = fake_predictions(7, n_genes=9)
    got = pp.assign_transcripts_to_cells(preds, max_iter=6)
    ref = po.assign_transcripts_to_cells([[t.numpy() for t in p] for p in preds], max_iter=6)
    assert np.array_equal(got["failed_genes"].numpy(), ref["failed_genes"])
    assert 0 < len(ref["failed_genes"]) < 9
    assert abs(got["global_threshold"] - ref["global_threshold"]) < 1e-9
    assert np.allclose(got["similarity_threshold"].numpy(), ref["similarity_threshold"], atol=1e-9, equal_nan=True)
    bad = np.isin(ref["gene"], ref["failed_genes"])
    assert np.allclose(got["similarity_threshold"].numpy()[bad], ref["global_threshold"])


def test_known_answers(po):
    # two equal spikes at 0.25 and 0.75: Li stops at the mean (mean_back == 0 after the shift) -> 0.5;
    # Yen's criterion is flat over all 255 cuts -> first bin centre = min + range / 512
    v = np.array([0.25] * 10 + [0.75] * 10, dtype=np.float32)
    assert abs(po.threshold_li(v) - 0.5) < 1e-12
    assert abs(po.threshold_yen(v) - (0.25 + 0.5 / 512)) < 1e-12
    sim = torch.tensor(v)
    genes, thr, conv, glob = pp.per_gene_thresholds(sim, torch.zeros(20, dtype=torch.long), torch.ones(20, dtype=torch.bool))
    assert genes.tolist() == [0] and bool(conv.all())
    assert abs(float(thr[0]) - (0.25 + 0.5 / 512)) < 1e-12 and abs(glob - float(thr[0])) < 1e-12
    # a gene whose assigned similarities are all equal: Li returns the value, Yen the centre of its +-0.5 window
    flat = np.full(5, 0.4, dtype=np.float32)
    assert po.threshold_li(flat) == pytest.approx(float(np.float32(0.4)))
    g2, t2, _, _ = pp.per_gene_thresholds(torch.tensor(flat), torch.zeros(5, dtype=torch.long), torch.ones(5, dtype=torch.bool))
    assert float(t2[0]) == pytest.approx(min(po.threshold_yen(flat), po.threshold_li(flat)), abs=1e-12)


def test_unassigned_genes_get_nan_and_frame_columns():
    idx = torch.arange(6)
    preds = [(idx, torch.tensor([0, 1, -1, -1, 2, 0]), torch.tensor([0.9, 0.2, 0.0, 0.0, 0.5, 0.7]),
              torch.tensor([0, 0, 1, 1, 0, 0], dtype=torch.int32))]
    out = pp.assign_transcripts_to_cells(preds)
    assert torch.isnan(out["similarity_threshold"][2:4]).all() and not torch.isnan(out["similarity_threshold"][[0, 1, 4, 5]]).any()
    import pandas as pd
    obs = pd.DataFrame({"cell_id": ["a", "b", "c"], "cell_encoding": [0, 1, 2]})
    df = pp.to_frame(out, obs)
    assert list(df.columns) == ["row_index", "segger_cell_id", "segger_similarity", "similarity_threshold"]
    assert df["segger_cell_id"].tolist()[:2] == ["a", "b"] and df["segger_cell_id"].isna().tolist() == [False, False, True, True, False, False]
    assert pp.to_frame(out)["segger_cell_id"].isna().sum() == 2


def test_metrics_auroc_matches_sklearn_and_oracle(oracle):
    from sklearn.metrics import roc_auc_score
    from segger_amd.metrics import auroc
    g = torch.Generator().manual_seed(0)
    lab = torch.rand(5000, generator=g) < 0.3
    s = torch.randn(5000, generator=g) + lab.float()
    s = (s * 8).round() / 8                                  # plenty of ties
    a = auroc(s, lab)
    assert abs(a - roc_auc_score(lab.numpy(), s.numpy())) < 1e-12
    assert abs(a - oracle.auroc(s, lab)) < 1e-12
    assert np.isnan(auroc(s, torch.zeros_like(lab)))
