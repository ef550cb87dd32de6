"""Synthetic transcript / nucleus graphs shaped like segger's HeteroData tiles.

Follows SURVEY.md 8(d): nuclei uniformly on an L x L field (L = 10 um *
sqrt(Nb)), every transcript drawn around a true cell with a 3 um Gaussian,
gene ~ Categorical(profile[type(cell)]), ``bd.x`` = type mean + noise, and the
three edge stores of reference ``src/segger/data/utils/heterodata.py:138-162``:

* ``tx-neighbors-tx``: exact kNN incl. self, source = query point, target =
  neighbour (``data/utils/neighbors.py:77-82,145-156``; scipy KDTree like the
  reference);
* ``tx-belongs-bd``  : transcripts within ``belongs_radius`` of their own centre;
* ``tx-neighbors-bd``: up to ``pred_k`` nearest centres within ``pred_radius``.

Node ids are Morton-sorted so tiles are spatially coherent, as segger's
quadtree tiles are (``data/tiling.py:198-233``).  Everything is generated on the
CPU from one seed; the result is a :class:`segger_amd.hetero.HeteroBatch`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .hetero import HeteroBatch, TX_TX, TX_BD, TX_NB_BD


@dataclass
class SyntheticSpec:
    n_tx: int = 1000
    n_bd: int = 100
    k_tx: int = 5
    n_genes: int = 256
    n_types: int = 8
    bd_dim: int = 128
    belongs_radius: float = 3.0     # ~39 % of transcripts (1 - exp(-r^2 / 2 sigma^2))
    pred_k: int = 3
    pred_radius: float = 8.0
    sigma: float = 3.0
    n_graphs: int = 1               # >1: square grid of tiles -> batch vector
    seed: int = 0


# the BASELINE.json configurations
C1 = SyntheticSpec(n_tx=1_000, n_bd=100, k_tx=5)
C2 = SyntheticSpec(n_tx=1_000_000, n_bd=10_000, k_tx=15)


def _morton(xy: np.ndarray, lo: float, hi: float) -> np.ndarray:
    q = np.clip(((xy - lo) / (hi - lo) * 65535.0), 0, 65535).astype(np.uint64)

    def spread(v):
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))


def _morton_torch(xy: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    q = ((xy - lo) / (hi - lo) * 65535.0).clamp_(0, 65535).long()

    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1)


def fov_nodes(spec: SyntheticSpec, device) -> dict:
    """The NODES of :func:`make_fov` (nuclei, transcripts in Morton order, genes, the similarity matrices): cheap -- 40 bytes
    per transcript -- next to the edge stores (16 bytes per edge, ~18 edges per transcript), which :func:`fov_edges` builds
    for all transcripts or for a subset of them."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev).manual_seed(spec.seed)
    Nt, Nb, T, G = spec.n_tx, spec.n_bd, spec.n_types, spec.n_genes
    L = 10.0 * float(np.sqrt(Nb))
    lo, hi = -4 * spec.sigma, L + 4 * spec.sigma

    def rand(*s):
        return torch.rand(*s, device=dev, generator=gen)

    def randn(*s):
        return torch.randn(*s, device=dev, generator=gen)

    centres = rand(Nb, 2) * L
    centres = centres[torch.argsort(_morton_torch(centres, lo, hi), stable=True)]
    bd_type = torch.randint(0, T, (Nb,), device=dev, generator=gen)
    profiles = torch.from_numpy(np.random.default_rng(spec.seed).dirichlet(np.full(G, 0.3), size=T)).to(dev)
    type_mean = randn(T, spec.bd_dim)
    bd_x = type_mean[bd_type] + 0.5 * randn(Nb, spec.bd_dim)

    cell = torch.randint(0, Nb, (Nt,), device=dev, generator=gen)
    pos = centres[cell] + spec.sigma * randn(Nt, 2)
    order = torch.argsort(_morton_torch(pos, lo, hi), stable=True)
    cell, pos = cell[order], pos[order].contiguous()
    del order
    # gene ~ Categorical(profile[type(cell)]): one searchsorted over the row-offset CDFs
    cdf = profiles.cumsum(1)
    cdf[:, -1] = 1.0
    flat = (cdf + torch.arange(T, device=dev, dtype=cdf.dtype)[:, None]).reshape(-1)
    ttype = bd_type[cell]
    u = torch.rand(Nt, device=dev, generator=gen, dtype=torch.float64) * (1.0 - 1e-12)
    gene = (torch.searchsorted(flat, u + ttype.to(torch.float64), right=True) - ttype * G).clamp_(0, G - 1)
    del u, flat
    gene_cluster = profiles.argmax(0)
    pn = profiles - profiles.mean(1, keepdim=True)
    pn = pn / pn.norm(dim=1, keepdim=True)
    tn = type_mean.double() / type_mean.double().norm(dim=1, keepdim=True)
    aux = {"tx_similarity": (pn @ pn.T).to(torch.float32), "bd_similarity": (tn @ tn.T).to(torch.float32)}
    return dict(centres=centres, bd_type=bd_type, bd_x=bd_x, cell=cell, pos=pos, gene=gene, gene_cluster=gene_cluster, aux=aux)


def fov_edges(nodes: dict, spec: SyntheticSpec, sel: Optional[torch.Tensor] = None):
    """(tx-neighbors-tx, tx-belongs-bd, tx-neighbors-bd) ``edge_index`` tensors [2, E] in GLOBAL node ids, for the query
    transcripts ``sel`` (ascending int64 ids; None = all): every store lists its edges by ascending source transcript, so
    the edges of a subset are exactly the corresponding rows of the full stores (a rank of a data-parallel run builds only
    the edges of its own tiles' transcripts; the grid kNN searches ALL points either way)."""
    from .neighbors import knn_grid, knn_to_edge_index
    pos, centres, cell = nodes["pos"], nodes["centres"], nodes["cell"]
    Nt, Nb = int(pos.shape[0]), int(centres.shape[0])
    q = pos if sel is None else pos.index_select(0, sel)
    own = (lambda t: t) if sel is None else (lambda t: sel[t])
    k = min(spec.k_tx, Nt)
    nbr, _ = knn_grid(pos, k, query=None if sel is None else q)
    ett, _ = knn_to_edge_index(nbr, padding_value=Nt)
    del nbr
    ett = torch.stack([own(ett[0]), ett[1]])
    cq = cell if sel is None else cell.index_select(0, sel)
    d_own = (q - centres[cq]).norm(dim=1)
    inside = (d_own < spec.belongs_radius).nonzero(as_tuple=False).squeeze(1)
    etb = torch.stack([own(inside), cq[inside]])
    del d_own, inside
    pk = min(spec.pred_k, Nb)
    cnb, _ = knn_grid(centres, pk, spec.pred_radius, query=q)
    ep, _ = knn_to_edge_index(cnb, padding_value=Nb)
    del cnb
    ep = torch.stack([own(ep[0]), ep[1]])
    return ett, etb, ep


def fov_graph(nodes: dict, edges, tx_ids: Optional[torch.Tensor] = None, bd_ids: Optional[torch.Tensor] = None) -> HeteroBatch:
    """The HeteroBatch of :func:`make_fov` from its nodes and edge stores -- or, with ``tx_ids`` / ``bd_ids`` (ascending
    global ids), the subgraph on those nodes: ``edges`` must then connect kept nodes only and come in global ids."""
    pos, centres, cell, gene = nodes["pos"], nodes["centres"], nodes["cell"], nodes["gene"]
    dev = pos.device
    Nt, Nb = int(pos.shape[0]), int(centres.shape[0])
    pick = lambda t, ids: t if ids is None else t.index_select(0, ids)
    b = HeteroBatch(num_graphs=1)
    tx, bd = b["tx"], b["bd"]
    g = pick(gene, tx_ids)
    nt = int(g.shape[0])
    tx["x"] = g.to(torch.int32)
    tx["pos"] = pick(pos, tx_ids).to(torch.float32)
    tx["batch"] = torch.zeros(nt, dtype=torch.long, device=dev)
    tx["cluster"] = nodes["gene_cluster"][g]
    tx["index"] = torch.arange(Nt, dtype=torch.int64, device=dev) if tx_ids is None else tx_ids.clone()
    tx["cell"] = pick(cell, tx_ids)
    bx = pick(nodes["bd_x"], bd_ids)
    nb = int(bx.shape[0])
    bd["x"] = bx.to(torch.float32)
    bd["pos"] = pick(centres, bd_ids).to(torch.float32)
    bd["batch"] = torch.zeros(nb, dtype=torch.long, device=dev)
    bd["cluster"] = pick(nodes["bd_type"], bd_ids).to(torch.int32)
    bd["index"] = (torch.arange(Nb, dtype=torch.int32, device=dev) if bd_ids is None else bd_ids.to(torch.int32))
    ett, etb, ep = edges
    if tx_ids is not None or bd_ids is not None:
        def local(ids, n):
            if ids is None:
                return lambda t: t
            m = torch.full((n,), -1, dtype=torch.int64, device=dev)
            m[ids] = torch.arange(ids.numel(), device=dev)
            return lambda t: m[t]
        lt, lb = local(tx_ids, Nt), local(bd_ids, Nb)
        ett = torch.stack([lt(ett[0]), lt(ett[1])])
        etb = torch.stack([lt(etb[0]), lb(etb[1])])
        ep = torch.stack([lt(ep[0]), lb(ep[1])])
    b[TX_TX]["edge_index"] = ett
    b[TX_BD]["edge_index"] = etb
    b[TX_NB_BD]["edge_index"] = ep
    return b


def make_fov(spec: SyntheticSpec, device, return_aux: bool = False):
    """The same generative model as :func:`make_graph`, built ON THE DEVICE for full-FOV sizes (BASELINE
    configs 3 / 5: 50-100 M transcripts): torch RNG instead of numpy's, and the three edge stores come from the
    HIP grid kNN (``segger_knn_grid``) instead of a host KD-tree.  Adds ``tx.cell`` (index of the true nucleus),
    so the label of a candidate edge is ``bd.index[dst] == tx.cell[src]`` in any tile batch."""
    nodes = fov_nodes(spec, device)
    b = fov_graph(nodes, fov_edges(nodes, spec))
    return (b, nodes["aux"]) if return_aux else b


def make_graph(spec: SyntheticSpec = C1, return_aux: bool = False):
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(spec.seed)
    Nt, Nb, T, G = spec.n_tx, spec.n_bd, spec.n_types, spec.n_genes
    L = 10.0 * np.sqrt(Nb)

    centres = rng.uniform(0, L, size=(Nb, 2))
    centres = centres[np.argsort(_morton(centres, -4 * spec.sigma, L + 4 * spec.sigma), kind="stable")]
    bd_type = rng.integers(0, T, size=Nb)
    profiles = rng.dirichlet(np.full(G, 0.3), size=T)
    type_mean = rng.normal(size=(T, spec.bd_dim))
    bd_x = type_mean[bd_type] + 0.5 * rng.normal(size=(Nb, spec.bd_dim))

    cell = rng.integers(0, Nb, size=Nt)
    pos = centres[cell] + spec.sigma * rng.normal(size=(Nt, 2))
    order = np.argsort(_morton(pos, -4 * spec.sigma, L + 4 * spec.sigma), kind="stable")
    cell, pos = cell[order], pos[order]
    # gene per transcript: inverse-CDF sampling from its cell type's profile
    cdf = np.cumsum(profiles, axis=1)
    cdf[:, -1] = 1.0
    u = rng.random(Nt)
    ttype = bd_type[cell]
    gene = np.empty(Nt, dtype=np.int64)
    for t in range(T):
        m = ttype == t
        gene[m] = np.searchsorted(cdf[t], u[m])
    gene = np.minimum(gene, G - 1)
    gene_cluster = profiles.argmax(0)            # cluster of a gene = its dominant type

    # --- edges ---------------------------------------------------------------
    # host KD-tree threads: all cores for a single process, a fair share when N ranks build their tiles at once
    import os
    world = max(1, int(os.environ.get("WORLD_SIZE", "1")))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    workers = -1 if world == 1 else max(1, cores // world)
    tree = cKDTree(pos, leafsize=64)
    k = min(spec.k_tx, Nt)
    _, nbr = tree.query(pos, k=k, workers=workers)
    nbr = nbr.reshape(Nt, k)
    ett = np.stack([np.repeat(np.arange(Nt), k), nbr.reshape(-1)])       # src = query, dst = neighbour

    d_own = np.linalg.norm(pos - centres[cell], axis=1)
    inside = np.nonzero(d_own < spec.belongs_radius)[0]
    etb = np.stack([inside, cell[inside]])

    ctree = cKDTree(centres, leafsize=32)
    pk = min(spec.pred_k, Nb)
    dist, cidx = ctree.query(pos, k=pk, distance_upper_bound=spec.pred_radius, workers=workers)
    dist, cidx = dist.reshape(Nt, pk), cidx.reshape(Nt, pk)
    ok = np.isfinite(dist)
    ep = np.stack([np.repeat(np.arange(Nt), pk)[ok.reshape(-1)], cidx[ok]])

    # --- graph ids (square grid of tiles) ------------------------------------
    g = int(round(np.sqrt(spec.n_graphs)))
    assert g * g == spec.n_graphs, "n_graphs must be a square number"

    def graph_id(p):
        gx = np.clip((p[:, 0] / L * g).astype(np.int64), 0, g - 1)
        gy = np.clip((p[:, 1] / L * g).astype(np.int64), 0, g - 1)
        return gy * g + gx

    b = HeteroBatch(num_graphs=spec.n_graphs)
    tx, bd = b["tx"], b["bd"]
    tx["x"] = torch.from_numpy(gene).to(torch.int32)
    tx["pos"] = torch.from_numpy(pos).to(torch.float32)
    tx["batch"] = torch.from_numpy(graph_id(pos))
    tx["cluster"] = torch.from_numpy(gene_cluster[gene]).to(torch.int64)
    tx["index"] = torch.arange(Nt, dtype=torch.int64)
    tx["mask"] = torch.ones(Nt, dtype=torch.bool)
    tx["predict_mask"] = torch.ones(Nt, dtype=torch.bool)
    bd["x"] = torch.from_numpy(bd_x).to(torch.float32)
    bd["pos"] = torch.from_numpy(centres).to(torch.float32)
    bd["batch"] = torch.from_numpy(graph_id(centres))
    bd["cluster"] = torch.from_numpy(bd_type).to(torch.int32)
    bd["index"] = torch.arange(Nb, dtype=torch.int32)
    bd["mask"] = torch.ones(Nb, dtype=torch.bool)
    b[TX_TX]["edge_index"] = torch.from_numpy(ett).to(torch.int64)
    b[TX_BD]["edge_index"] = torch.from_numpy(etb).to(torch.int64)
    b[TX_NB_BD]["edge_index"] = torch.from_numpy(ep).to(torch.int64)
    if not return_aux:
        return b
    # cluster similarity matrices (what ISTDataModule exposes as tx_similarity /
    # bd_similarity, reference data/utils/anndata.py:105-128): cosine similarity
    # of the type profiles / type means, in [-1, 1]
    pn = profiles - profiles.mean(1, keepdims=True)
    pn /= np.linalg.norm(pn, axis=1, keepdims=True)
    tn = type_mean / np.linalg.norm(type_mean, axis=1, keepdims=True)
    aux = {
        "label": torch.from_numpy((ep[1] == cell[ep[0]])),       # true tx->cell edge
        "cell": torch.from_numpy(cell),
        "tx_similarity": torch.from_numpy(pn @ pn.T).to(torch.float32),
        "bd_similarity": torch.from_numpy(tn @ tn.T).to(torch.float32),
    }
    return b, aux
