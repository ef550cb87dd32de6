"""A synthetic field of view as a resident, partitioned graph and its stream of packed tile batches: the workload of
BASELINE.json configs 3 / 4 (``tools/fov_stream.py``; ``bench.py --gpus N`` strong-scaling record).

Restates what ``ISTDataModule`` does around the path (reference ``src/segger/data/data_module.py:155-158,336-384``:
tiles of ~``tile_nodes`` transcripts, batches packed up to ``edges_per_batch`` edges) with the device tile batcher of
:mod:`segger_amd.tiles`."""
from __future__ import annotations

import math
from typing import List, Tuple

import torch

from .synthetic import SyntheticSpec, make_fov
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
