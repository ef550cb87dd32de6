#!/usr/bin/env python3
"""Edge-AUROC of the HIP path against the CPU oracle AT FOV SCALE (BASELINE.json: "edge-AUROC within 1e-3 of reference
on a 50M-transcript synthetic Xenium-scale graph").  A checker script (it lives under tests/ because it runs the oracle;
not collected by pytest):

    python tests/fov_oracle_auroc.py --n-tx 50000000 --n-bd 500000 --sample-tiles 32 --out profiles/r02_fov_50m_oracle_auroc.json

1. builds the synthetic FOV on the device, partitions it into resident tiles, trains ``--train-batches`` packed
   batches (bf16) so the weights are not the initial ones;
2. draws a seeded random sample of tiles (tiles are independent graphs: training drops inter-tile edges, prediction
   works tile by tile, so a random sample of tiles is an unbiased estimator of the FOV's edge ranking quality);
3. scores the ``tx-neighbors-bd`` candidate edges of every sampled tile with the HIP path in f32 / bf16 / f16 and with
   the oracle (fp32 = what PyG's CPU path computes, same weights) on the host;
4. reports, over the pooled edges of the sample: AUROC per path, |dAUROC| of each HIP dtype vs the oracle, and the
   largest per-edge score difference.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-tx", type=int, default=50_000_000)
    ap.add_argument("--n-bd", type=int, default=500_000)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--edges-per-batch", type=int, default=16_000_000)
    ap.add_argument("--train-batches", type=int, default=64, help="64 >= one epoch of the 50M FOV at 16M-edge batches")
    ap.add_argument("--sample-tiles", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import segger_oracle as O
    from segger_amd import LitISTEncoder, TX_NB_BD, ops
    from segger_amd.fov import build_fov_batches
    from segger_amd.graph import batch_cache, edge_graph
    from segger_amd.metrics import auroc
    from segger_amd.synthetic import SyntheticSpec
    dev = torch.device("cuda")
    log = lambda *a: print(*a, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    spec = SyntheticSpec(n_tx=args.n_tx, n_bd=args.n_bd, k_tx=args.k, seed=args.seed)
    part, batches, aux, tiling = build_fov_batches(spec, dev, edges_per_batch=args.edges_per_batch)
    log(f"[auroc] FOV {args.n_tx} tx: {len(tiling)} tiles, {len(batches)} batches ({time.perf_counter() - t0:.1f}s)")
    torch.manual_seed(0)
    model = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
    model.model._materialize_bd(spec.bd_dim, "cpu")
    model = model.to(dev)
    model.set_similarities(aux["tx_similarity"].to(dev), aux["bd_similarity"].to(dev))
    model._max_epochs_override, model.current_epoch = 20, 10
    model.model.compute_dtype = torch.bfloat16
    model.train()
    opt = model.configure_optimizers()
    for i, ids in enumerate(batches[: args.train_batches]):
        opt.zero_grad(set_to_none=True)
        model.training_step(part.batch(ids), i).backward()
        opt.step()
    torch.cuda.synchronize()
    log(f"[auroc] trained {min(args.train_batches, len(batches))} batches")
    model.eval()
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(args.seed + 1)
    usable = [t for t in range(len(part)) if int(part.node_sizes["bd"][t]) > 1 and int(part.edge_sizes[TX_NB_BD][t]) > 0]
    sample = [usable[i] for i in torch.randperm(len(usable), generator=g)[: args.sample_tiles].tolist()]
    DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
    scores = {k: [] for k in ("oracle",) + tuple(DT)}
    labels = []
    t_or = 0.0
    for n_done, t in enumerate(sample):
        b = part.tile(t)
        ei = b[TX_NB_BD].edge_index
        labels.append((b["bd"]["index"][ei[1]].long() == b["tx"]["cell"][ei[0]]).cpu())
        with torch.no_grad():
            for name, dt in DT.items():
                model.model.compute_dtype = dt
                z = model.forward(b)
                gph = edge_graph(batch_cache(b), TX_NB_BD, ei, b["tx"].num_nodes, b["bd"].num_nodes, need_by_dst=False)
                _, _, _, sim = ops.edge_cos_argmax(gph.by_src, z["tx"], z["bd"], return_sim=True)
                scores[name].append(sim.float().cpu())
        bc = b.to("cpu")
        t1 = time.perf_counter()
        with torch.no_grad():
            zr = O.ist_encoder_forward(sd, bc.x_dict, bc.edge_index_dict, bc.pos_dict, bc.batch_dict, n_heads=2)
        t_or += time.perf_counter() - t1
        scores["oracle"].append(O.edge_scores(zr["tx"], zr["bd"], bc[TX_NB_BD].edge_index).float())
        if n_done % 8 == 0:
            log(f"[auroc] tile {n_done + 1}/{len(sample)} (oracle {t_or:.1f}s so far)")
    lab = torch.cat(labels)
    cat = {k: torch.cat(v) for k, v in scores.items()}
    a_or = auroc(cat["oracle"], lab)
    res = {"workload": f"synthetic FOV {args.n_tx} tx / {args.n_bd} nuclei, k={args.k}; {len(tiling)} tiles; weights after "
                       f"{min(args.train_batches, len(batches))} bf16 training batches of <= {args.edges_per_batch} edges",
           "sample_tiles": len(sample), "sample_edges": int(lab.numel()), "positives": int(lab.sum()),
           "oracle": "oracle/segger_oracle.py fp32 on the host (what PyG's CPU path computes), same weights",
           "auroc_oracle_sample": a_or, "oracle_seconds": t_or, "hip": {}}
    for name in DT:
        a = auroc(cat[name], lab)
        d = (cat[name] - cat["oracle"]).abs()
        res["hip"][name] = {"auroc_hip_sample": a, "abs_diff_vs_oracle": abs(a - a_or),
                            "score_diff_vs_oracle": {"mean": float(d.mean()), "p99": float(d.quantile(0.99)),
                                                     "p99.9": float(d.kthvalue(int(0.999 * d.numel()))[0]),
                                                     "max": float(d.max())}}
    res["target"] = "|dAUROC| <= 1e-3 (BASELINE.json north star)"
    res["met"] = all(v["abs_diff_vs_oracle"] <= 1e-3 for v in res["hip"].values())
    s = json.dumps(res)
    print(s, flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
