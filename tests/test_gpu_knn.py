"""Grid kNN (C ABI segger_knn_grid) against scipy's KDTree, the reference's own neighbour search
(src/segger/data/utils/neighbors.py:139-150): same distances, same neighbour sets, padding = n."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def kdtree(points, queries, k, max_dist):
    from scipy.spatial import cKDTree
    d, i = cKDTree(points).query(queries, k=k, distance_upper_bound=max_dist)
    return d.reshape(len(queries), k), i.reshape(len(queries), k)


@pytest.mark.parametrize("n,k,max_dist", [(5000, 15, np.inf), (5000, 5, 3.0), (300, 8, np.inf), (40, 64, np.inf),
                                          (2000, 1, np.inf), (3000, 20, 0.5)])
def test_knn_matches_kdtree(cuda, n, k, max_dist):
    from segger_amd.neighbors import knn_grid
    rng = np.random.default_rng(n + k)
    # clustered points (like transcripts around nuclei) + a few far outliers
    centres = rng.uniform(0, 100, size=(max(n // 50, 1), 2))
    pts = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 2.0, size=(n, 2))
    pts[:3] += 500.0
    pts = pts.astype(np.float32)
    d_ref, i_ref = kdtree(pts.astype(np.float64), pts.astype(np.float64), k, max_dist)
    nbr, dist = knn_grid(torch.from_numpy(pts).to(cuda), k, max_dist, return_dist=True)
    nbr, dist = nbr.cpu().numpy(), dist.cpu().numpy()
    assert np.array_equal(np.isinf(d_ref), nbr == n), "padding differs"
    ok = ~np.isinf(d_ref)
    assert np.allclose(dist[ok], d_ref[ok], rtol=1e-5, atol=1e-5)
    assert (nbr[:, 0] == np.arange(n)).mean() > 0.999            # nearest neighbour of a point is itself
    # neighbour ids agree wherever distances are distinct
    strict = ok.copy()
    strict[:, 1:] &= np.abs(d_ref[:, 1:] - d_ref[:, :-1]) > 1e-4
    strict[:, :-1] &= np.abs(d_ref[:, 1:] - d_ref[:, :-1]) > 1e-4
    assert (nbr[strict] == i_ref[strict]).mean() > 0.9999


def test_knn_separate_queries_and_edge_index(cuda):
    from segger_amd.neighbors import knn_grid, knn_to_edge_index, transcripts_graph, prediction_graph_uniform
    rng = np.random.default_rng(0)
    pts = rng.uniform(0, 50, size=(4000, 2)).astype(np.float32)
    qs = rng.uniform(-5, 55, size=(300, 2)).astype(np.float32)
    d_ref, i_ref = kdtree(pts.astype(np.float64), qs.astype(np.float64), 6, 4.0)
    nbr, dist = knn_grid(torch.from_numpy(pts).to(cuda), 6, 4.0, query=torch.from_numpy(qs).to(cuda), return_dist=True)
    assert np.array_equal(np.isinf(d_ref), nbr.cpu().numpy() == 4000)
    ok = ~np.isinf(d_ref)
    assert np.allclose(dist.cpu().numpy()[ok], d_ref[ok], rtol=1e-5, atol=1e-5)
    ei, indptr = knn_to_edge_index(nbr, padding_value=4000)       # padding = number of POINTS (reference :154)
    deg = ok.sum(1)
    assert ei.shape[1] == ok.sum() and torch.equal(indptr.cpu(), torch.from_numpy(np.concatenate([[0], deg.cumsum()])))
    assert torch.equal(ei[0].cpu(), torch.from_numpy(np.repeat(np.arange(300), deg)))      # row = query (source)
    ett = transcripts_graph(torch.from_numpy(pts).to(cuda), 5)
    assert ett.shape == (2, 4000 * 5) and ett.dtype == torch.int64
    assert ((ett[0] == ett[1]).view(4000, 5).sum(1) == 1).all()
    ep = prediction_graph_uniform(torch.from_numpy(pts).to(cuda), torch.from_numpy(qs).to(cuda), 3)
    assert ep.shape == (2, 900) and int(ep[1].max()) < 4000


def test_knn_million_points_property(cuda):
    """C2 scale: k=15 over 1M points -- rows sorted, self first, distances match a brute-force check on a sample."""
    from segger_amd.neighbors import knn_grid
    g = torch.Generator(device=cuda).manual_seed(0)
    pts = torch.rand(1_000_000, 2, device=cuda, generator=g) * 1000
    nbr, dist = knn_grid(pts, 15, return_dist=True)
    assert (nbr[:, 0] == torch.arange(1_000_000, device=cuda, dtype=torch.int32)).all()
    assert (dist[:, 1:] >= dist[:, :-1]).all() and int(nbr.max()) < 1_000_000
    idx = torch.randint(0, 1_000_000, (64,), device=cuda, generator=g)
    d = (pts[None, :, :] - pts[idx][:, None, :]).pow(2).sum(-1).sqrt()       # [64, 1M] brute force, exact differences
    ref = d.topk(15, largest=False).values
    assert torch.allclose(ref, dist[idx], atol=1e-3)
