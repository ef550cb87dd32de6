#!/usr/bin/env python3
"""Generates tests/golden/bin_packing.npz from the REFERENCE's own bin packers.  Run from the repo root IN THE BUILD
CONTAINER (it needs /root/reference):

    python tests/golden/make_bin_packing_golden.py

It imports ``/root/reference/src/segger/data/partition/sampler.py`` as a file.  That module's two non-stdlib imports
(``torch_geometric.loader.DataLoader`` at ``:2`` and the sibling ``.dataset.PartitionDataset`` at ``:8``) are only
used by the ``PartitionSampler`` class, never by the three pure functions exercised here
(``best_fit_decreasing`` ``:11-82``, ``harmonic_k`` ``:85-183``, ``first_fit_decreasing_bucketed`` ``:186-289``);
both names are satisfied by empty placeholder modules -- the same way make_golden.py loads ``triplet_loss.py``.
The outputs are genuine reference results; only the data (inputs + bins) is committed.
"""
import importlib.util
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/segger/data/partition/sampler.py"


def load_reference():
    tg, tgl = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.loader")
    tgl.DataLoader = object
    tg.loader = tgl
    sys.modules.setdefault("torch_geometric", tg)
    sys.modules["torch_geometric.loader"] = tgl
    pkg = types.ModuleType("refpartition")
    pkg.__path__ = []                                    # a package, so the relative import resolves
    ds = types.ModuleType("refpartition.dataset")
    ds.PartitionDataset = object
    sys.modules["refpartition"], sys.modules["refpartition.dataset"] = pkg, ds
    spec = importlib.util.spec_from_file_location("refpartition.sampler", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refpartition.sampler"] = mod
    spec.loader.exec_module(mod)
    return mod


def cases():
    rng = random.Random(20260501)
    out = []
    for n, cap, lo, hi in [(1, 10.0, 1, 10), (7, 10.0, 1, 10), (40, 100.0, 1, 100), (200, 1000.0, 5, 700),
                           (64, 1_000_000.0, 20_000, 900_000), (300, 50.0, 1, 50), (25, 7.0, 1, 7)]:
        out.append(([float(rng.randint(lo, hi)) for _ in range(n)], cap, False))
    # equal items, items equal to the capacity, fractional sizes
    out.append(([5.0] * 13, 10.0, False))
    out.append(([10.0, 10.0, 3.0, 10.0, 7.0], 10.0, False))
    out.append(([round(rng.uniform(0.05, 1.0), 3) for _ in range(60)], 1.0, False))
    # skip_too_big: non-positive and oversize items are ignored
    out.append(([12.0, 3.0, 0.0, 9.5, -1.0, 4.0, 25.0, 6.0, 1.0], 10.0, True))
    out.append(([float(rng.randint(1, 1500)) for _ in range(80)], 1000.0, True))
    return out


def main():
    ref = load_reference()
    recs = []
    for items, cap, skip in cases():
        rec = {"items": items, "capacity": cap, "skip_too_big": skip,
               "best_fit_decreasing": ref.best_fit_decreasing(list(items), cap, skip_too_big=skip),
               "first_fit_decreasing": ref.first_fit_decreasing_bucketed(list(items), cap, skip_too_big=skip, n_buckets=None)}
        for k in (2, 3, 6, 10):
            rec[f"harmonic_{k}"] = ref.harmonic_k(list(items), cap, k=k, skip_too_big=skip)
        # the randomised variants with a given random.Random: the draw order is part of the behaviour
        for nb in (1, 3):
            rec[f"ffd_buckets_{nb}_seed7"] = ref.first_fit_decreasing_bucketed(
                list(items), cap, skip_too_big=skip, n_buckets=nb, rng=random.Random(7))
        recs.append(rec)
    errors = {}
    for name, fn in (("best_fit_decreasing", ref.best_fit_decreasing), ("harmonic_k", ref.harmonic_k),
                     ("first_fit_decreasing_bucketed", ref.first_fit_decreasing_bucketed)):
        try:
            fn([3.0, 11.0], 10.0)
        except Exception as e:  # noqa: BLE001
            errors[name] = [type(e).__name__, str(e)]
    try:
        ref.harmonic_k([1.0], 10.0, k=1)
    except Exception as e:  # noqa: BLE001
        errors["harmonic_k_k1"] = [type(e).__name__, str(e)]
    blob = json.dumps({"cases": recs, "errors": errors})
    np.savez_compressed(os.path.join(HERE, "bin_packing.npz"), json=np.frombuffer(blob.encode(), dtype=np.uint8))
    print(f"bin_packing.npz: {len(recs)} cases, errors: {errors}")


if __name__ == "__main__":
    main()
