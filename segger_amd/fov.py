"""A synthetic field of view as a resident, partitioned graph and its stream of packed tile batches: the workload of
BASELINE.json configs 3 / 4 (``tools/fov_stream.py``; ``bench.py --gpus N`` strong-scaling record).

Restates what ``ISTDataModule`` does around the path (reference ``src/segger/data/data_module.py:155-158,336-384``:
tiles of ~``tile_nodes`` transcripts, batches packed up to ``edges_per_batch`` edges) with the device tile batcher of
:mod:`segger_amd.tiles`."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .synthetic import SyntheticSpec, fov_edges, fov_graph, fov_nodes, make_fov
from .tiles import SquareTiling, TileBatchSampler, TilePartition, partition_by_tiling


def build_fov_batches(spec: SyntheticSpec, device, *, tile_nodes: int = 50_000, margin: float = 10.0,
                      edges_per_batch: int = 1_000_000, slide_csr: bool = True, keep_data: bool = False):
    """-> (partition, batches, aux, tiling[, data]).  Every rank of a data-parallel run calls this with the same seed
    and gets the same partition and the same batch list (no data-path collective)."""
    data, aux = make_fov(spec, device, return_aux=True)
    L = 10.0 * math.sqrt(spec.n_bd)
    side = math.sqrt(tile_nodes / (spec.n_tx / (L * L)))
    tiling = SquareTiling(data["tx"]["pos"], side)
    part = partition_by_tiling(data, tiling, margin=margin)
    part.add_node_attr("tx", "predict_mask", torch.ones(spec.n_tx, dtype=torch.bool, device=device), permuted=True)
    if slide_csr:
        part.build_csr()
    batches: List[List[int]] = list(TileBatchSampler(part, edges_per_batch, mode="edge", skip_too_big=True))
    if keep_data:
        return part, batches, aux, tiling, data
    del data
    return part, batches, aux, tiling


def batch_weights(part: TilePartition, batches) -> List[float]:
    w = part.weights("edge")
    return [float(sum(w[t] for t in ids)) for ids in batches]


class _TileWeights:
    """What :class:`TileBatchSampler` asks of a partition, from per-tile counts alone."""

    def __init__(self, edge_sizes: Dict, node_sizes: Dict):
        self.edge_sizes, self.node_sizes = edge_sizes, node_sizes

    def weights(self, mode: str = "edge") -> List[int]:
        src = self.edge_sizes if mode == "edge" else self.node_sizes
        return torch.stack([v.cpu() for v in src.values()]).sum(0).tolist()


def _intra_tile_edge_counts(nodes: dict, spec: SyntheticSpec, lab_tx: torch.Tensor, lab_bd: torch.Tensor, n_tiles: int,
                            chunk: int) -> Dict:
    """Edges per tile and edge type that :class:`TilePartition` would keep (both endpoints in the tile), counted over
    chunks of query transcripts without keeping a single edge: the packing weights of every tile, known to every rank."""
    from .hetero import TX_BD, TX_NB_BD, TX_TX
    from .neighbors import knn_grid
    pos, centres, cell = nodes["pos"], nodes["centres"], nodes["cell"]
    Nt, Nb = int(pos.shape[0]), int(centres.shape[0])
    dev = pos.device
    counts = {et: torch.zeros(n_tiles, dtype=torch.int64, device=dev) for et in (TX_TX, TX_BD, TX_NB_BD)}
    lab_tx_pad = torch.cat([lab_tx, lab_tx.new_full((1,), -1)])        # (padding entries of a neighbour table point here)
    lab_bd_pad = torch.cat([lab_bd, lab_bd.new_full((1,), -1)])
    k, pk = min(spec.k_tx, Nt), min(spec.pred_k, Nb)
    for a in range(0, Nt, chunk):
        b = min(a + chunk, Nt)
        q, lq = pos[a:b], lab_tx[a:b]
        nbr, _ = knn_grid(pos, k, query=q)
        same = lab_tx_pad[nbr.long()] == lq[:, None]
        counts[TX_TX] += torch.bincount(lq, weights=same.sum(1).double(), minlength=n_tiles).long()
        del nbr, same
        cq = cell[a:b]
        inside = ((q - centres[cq]).norm(dim=1) < spec.belongs_radius) & (lab_bd[cq] == lq)
        counts[TX_BD] += torch.bincount(lq[inside], minlength=n_tiles)
        cnb, _ = knn_grid(centres, pk, spec.pred_radius, query=q)
        same = lab_bd_pad[cnb.long()] == lq[:, None]
        counts[TX_NB_BD] += torch.bincount(lq, weights=same.sum(1).double(), minlength=n_tiles).long()
        del cnb, same
    return counts


def build_fov_shard(spec: SyntheticSpec, device, rank: int, world_size: int, *, tile_nodes: int = 50_000, margin: float = 10.0,
                    edges_per_batch: int = 1_000_000, slide_csr: bool = True, count_chunk: int = 1 << 23):
    """The rank-local form of :func:`build_fov_batches` + ``dp.rank_schedule`` + ``TilePartition.shard``: the same seed-``spec``
    FOV, tiling, batch list and schedule on every rank (no data-path collective), but a rank builds the EDGES -- 16 bytes
    each, ~18 per transcript: 9/10 of the FOV's bytes -- of its own tiles' transcripts only.  The nodes (40 bytes per
    transcript) and a chunked counting pass over all transcripts (the packing weights of every tile: kNN results counted,
    not kept) are replicated.  -> (partition of this rank's tiles, renumbered 0..k-1 in ascending global order: tile for tile
    what ``build_fov_batches(...)[0].shard(tiles)`` holds; batches: the global list of tile-id lists; local_batches: {batch
    number: local tile ids} of this rank; schedule; aux; tiling; info: tiles, per-batch weights and units)."""
    from .dp import rank_schedule
    from .hetero import TX_BD, TX_TX
    nodes = fov_nodes(spec, device)
    pos, centres = nodes["pos"], nodes["centres"]
    dev = pos.device
    L = 10.0 * math.sqrt(spec.n_bd)
    side = math.sqrt(tile_nodes / (spec.n_tx / (L * L)))
    tiling = SquareTiling(pos.to(torch.float32), side)
    n_tiles = len(tiling)
    lab_tx, lab_bd = tiling.label(pos.to(torch.float32)), tiling.label(centres.to(torch.float32))
    counts = _intra_tile_edge_counts(nodes, spec, lab_tx, lab_bd, n_tiles, count_chunk)
    node_sizes = {"tx": torch.bincount(lab_tx, minlength=n_tiles), "bd": torch.bincount(lab_bd, minlength=n_tiles)}
    plan = _TileWeights(counts, node_sizes)
    batches: List[List[int]] = list(TileBatchSampler(plan, edges_per_batch, mode="edge", skip_too_big=True))
    w_tile = plan.weights("edge")
    weights = [float(sum(w_tile[t] for t in ids)) for ids in batches]
    schedule = rank_schedule(weights, world_size)
    mine = [k for k in schedule[rank] if k is not None]
    tiles_mine = sorted({t for k in mine for t in batches[k]})
    remap = {t: i for i, t in enumerate(tiles_mine)}
    own = torch.zeros(n_tiles, dtype=torch.bool, device=dev)
    if tiles_mine:
        own[torch.tensor(tiles_mine, device=dev)] = True
    tx_ids = own[lab_tx].nonzero(as_tuple=False).squeeze(1)
    bd_ids = own[lab_bd].nonzero(as_tuple=False).squeeze(1)
    ett, etb, ep = fov_edges(nodes, spec, sel=tx_ids)
    # intra-tile edges only (what the partition keeps anyway): their other endpoint is then one of this rank's nodes
    ett = ett[:, lab_tx[ett[0]] == lab_tx[ett[1]]]
    etb = etb[:, lab_tx[etb[0]] == lab_bd[etb[1]]]
    ep = ep[:, lab_tx[ep[0]] == lab_bd[ep[1]]]
    data = fov_graph(nodes, (ett, etb, ep), tx_ids, bd_ids)
    local_of = torch.full((n_tiles,), -1, dtype=torch.int64, device=dev)
    if tiles_mine:
        local_of[torch.tensor(tiles_mine, device=dev)] = torch.arange(len(tiles_mine), device=dev)
    part = TilePartition(data, {"tx": local_of[lab_tx[tx_ids]], "bd": local_of[lab_bd[bd_ids]]}, len(tiles_mine))
    # permutations in GLOBAL node numbers, as a shard of the whole partition carries them
    part.node_perm = {"tx": tx_ids[part.node_perm["tx"]], "bd": bd_ids[part.node_perm["bd"]]}
    for nt in ("tx", "bd"):
        part.add_node_attr(nt, "mask", tiling.mask(part.data[nt]["pos"], margin), permuted=True)
    part.add_node_attr("tx", "predict_mask", torch.ones(int(tx_ids.numel()), dtype=torch.bool, device=dev), permuted=True)
    if slide_csr:
        part.build_csr()
    e_tb, e_tt = counts[TX_BD].tolist(), counts[TX_TX].tolist()
    info = {"tiles": tiles_mine, "n_tiles": n_tiles, "weights": weights,
            "units": {k: (sum(e_tb[t] for t in ids), sum(e_tt[t] for t in ids)) for k, ids in enumerate(batches)}}
    local_batches = {k: [remap[t] for t in batches[k]] for k in mine}
    return part, batches, local_batches, schedule, nodes["aux"], tiling, info
