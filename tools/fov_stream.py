#!/usr/bin/env python3
"""Full-FOV streaming run (BASELINE.json configs 3 and 5) on ONE MI355X.

Builds a synthetic Xenium-scale field of view on the device (``synthetic.make_fov``: N transcripts, grid kNN
edges), partitions it into square tiles that stay resident in HBM (``tiles.partition_by_tiling``), packs the
tiles into batches of <= ``--edges-per-batch`` edges (``TileBatchSampler``; segger's ``edges_per_batch``,
reference ``data/data_module.py:158``) and

1. trains over the batch stream (fwd + bwd + Adam)               -> tx->cell edges scored / s,
2. scores every ``tx-neighbors-bd`` candidate edge of the FOV with the SAME weights in each requested dtype
   -> edge-AUROC per dtype (label: the candidate is the transcript's true nucleus) and |dAUROC| vs fp32,
3. (``--graphed``) repeats the scoring pass through the hipGraph-captured predictor, one graph per shape
   bucket (config 5: inference only, fp16).

Under ``python -m torch.distributed.run --nproc-per-node N … tools/fov_stream.py`` the training epoch is data
parallel over the packed batches (BASELINE config 4): every rank builds the same FOV from the seed (0.4 s; no data
path collective), takes its share of the batches (``dp.rank_schedule``: balanced edge counts, empty steps where a
rank runs out), one flat RCCL all-reduce per step; scoring / prediction stay on rank 0 (inference needs no
collective -- tiles would simply be split).

Prints one JSON object; progress goes to stderr.  This is a driver around the product path: it never touches
``oracle/`` (the oracle comparison at tile scale lives in tests/test_gpu_fov.py).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Phase:
    def __init__(self, name, out):
        self.name, self.out = name, out

    def __enter__(self):
        torch.cuda.synchronize()
        self.t = time.perf_counter()
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        self.out[self.name] = time.perf_counter() - self.t
        log(f"[fov] {self.name}: {self.out[self.name]:.2f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-tx", type=int, default=50_000_000)
    ap.add_argument("--n-bd", type=int, default=500_000)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--tile-nodes", type=int, default=50_000, help="target transcripts per tile (data_module.py:155)")
    ap.add_argument("--edges-per-batch", type=int, default=1_000_000)
    ap.add_argument("--margin", type=float, default=10.0, help="tile margin excluded from the losses (um)")
    ap.add_argument("--train-batches", type=int, default=0, help="0 = one full epoch")
    ap.add_argument("--train-epochs", type=int, default=1,
                    help="epochs over the batch stream; the LAST one is reported as `train`, all of them in "
                         "`train.epochs_s` (the first epoch builds the per-tile sampler indices)")
    ap.add_argument("--train-dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--graphed-train", action="store_true",
                    help="the whole step (fwd + losses + bwd + Adam) as one hipGraph replay per batch "
                         "(segger_amd.train_step_graph); with several ranks: two graphs around the gradient all-reduce")
    ap.add_argument("--score-dtypes", default="f32,bf16,f16")
    ap.add_argument("--graphed", action="store_true", help="also run the hipGraph-captured predictor (fp16)")
    ap.add_argument("--overlap-predict", action="store_true",
                    help="also run segger's real predict pipeline: overlapping tiles (bbox + margin, predict_mask), "
                         "predict_step per tile, dedup + per-gene thresholds")
    ap.add_argument("--no-slide-csr", action="store_true",
                    help="sort the edges of every batch (5 radix sorts) instead of slicing the once-per-slide CSR views")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None, help="also write the JSON here")
    ap.add_argument("--segmentation-parquet", default=None,
                    help="with --overlap-predict: write the transcript -> cell table (row_index, segger_cell_id, "
                         "segger_similarity, similarity_threshold), the columns of segger_segmentation.parquet")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("fov_stream.py needs an MI355X: there is no CPU fallback")
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from segger_amd import LitISTEncoder, TX_BD, TX_NB_BD, TX_TX, ops
    from segger_amd.dp import FlatGradBucket, broadcast_parameters, rank_schedule, seed_rank
    from segger_amd.graph import batch_cache, edge_graph
    from segger_amd.inference import GraphedPredictor, bucket_sizes
    from segger_amd.metrics import assignment_accuracy, auroc
    from segger_amd.synthetic import SyntheticSpec, make_fov
    from segger_amd.tiles import PredictTileIndex, SquareTiling, TileBatchSampler, partition_by_tiling

    DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
    times: dict = {}
    spec = SyntheticSpec(n_tx=args.n_tx, n_bd=args.n_bd, k_tx=args.k, seed=args.seed)
    with Phase("build_fov_s", times):
        data, aux = make_fov(spec, dev, return_aux=True)
    n_edges = {"__".join(et): int(data[et].edge_index.shape[1]) for et in (TX_TX, TX_BD, TX_NB_BD)}
    log(f"[fov] {args.n_tx} tx, {args.n_bd} nuclei, edges {n_edges}")

    with Phase("partition_s", times):
        L = 10.0 * math.sqrt(args.n_bd)
        side = math.sqrt(args.tile_nodes / (args.n_tx / (L * L)))
        tiling = SquareTiling(data["tx"]["pos"], side)
        part = partition_by_tiling(data, tiling, margin=args.margin)
        part.add_node_attr("tx", "predict_mask", torch.ones(args.n_tx, dtype=torch.bool, device=dev), permuted=True)
        pti = PredictTileIndex(data, tiling, margin=args.margin) if args.overlap_predict else None
        del data
        if not args.no_slide_csr:
            part.build_csr()
        sampler = TileBatchSampler(part, args.edges_per_batch, mode="edge", skip_too_big=True)
        batches = list(sampler)
    kept = {"__".join(et): int(v.sum()) for et, v in part.edge_sizes.items()}
    log(f"[fov] {len(tiling)} tiles of side {side:.1f} um -> {len(batches)} batches; intra-tile edges {kept}")
    torch.cuda.empty_cache()

    torch.manual_seed(0)
    model = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
    model.model._materialize_bd(spec.bd_dim, "cpu")
    model = model.to(dev)
    model.set_similarities(aux["tx_similarity"].to(dev), aux["bd_similarity"].to(dev))
    model._max_epochs_override = 20
    model.current_epoch = 10
    graphed_train = bool(args.graphed_train)
    opt = model.configure_optimizers(capturable=graphed_train)
    broadcast_parameters(model)
    seed_rank(args.seed, rank, model.model)
    bucket = FlatGradBucket(model.parameters())

    # ---- 1. training over the batch stream (data parallel over the packed batches when world > 1) ------
    model.model.compute_dtype = DT[args.train_dtype]
    model.train()
    todo = batches if args.train_batches <= 0 else batches[: args.train_batches]
    w_all = part.weights("edge")
    sched = rank_schedule([sum(w_all[t] for t in ids) for ids in todo], world)[rank]

    trainer = None
    if graphed_train:
        from segger_amd.train_step_graph import GraphedTrainer
        trainer = GraphedTrainer(model, opt, grad_sync=bucket.all_reduce_mean if world > 1 else None)

    def train_step(k, i):
        if trainer is not None:                              # (k None: the empty step of a rank out of batches)
            out = trainer.step(part.batch(todo[k]) if k is not None else None)
            return None if out is None else out[3]
        opt.zero_grad(set_to_none=True)
        loss = None
        if k is not None:                                    # None: this rank ran out of batches (empty step)
            loss = model.training_step(part.batch(todo[k]), i)
            loss.backward()
        bucket.all_reduce_mean()
        opt.step()
        return loss

    for k in sched[:3]:                                      # warm-up: lazy inits, allocator
        train_step(k, 0)
    etb_seen = ett_seen = 0
    eptr_tb, eptr_tt = part.edge_sizes[TX_BD].tolist(), part.edge_sizes[TX_TX].tolist()
    loss = None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    epochs_s = []
    for ep in range(max(1, args.train_epochs)):
        etb_seen = ett_seen = 0
        with Phase("train_s", times):
            for i, k in enumerate(sched):
                l = train_step(k, i)
                if k is not None:
                    loss = l
                    etb_seen += sum(eptr_tb[t] for t in todo[k])
                    ett_seen += sum(eptr_tt[t] for t in todo[k])
                if i % 200 == 0 and loss is not None:
                    log(f"[fov r{rank}] epoch {ep} train step {i}/{len(sched)} loss {float(loss.detach()):.4f}")
            if world > 1:
                dist.barrier()
        epochs_s.append(times["train_s"])
    stats = torch.tensor([times["train_s"], float(etb_seen), float(ett_seen)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = stats[1:].clone(); dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        times["train_s"], etb_seen, ett_seen = float(tmax[0]), float(sums[0]), float(sums[1])
    train = {
        "batches": len(todo), "steps_per_rank": len(sched), "n_gpus": world, "dtype": args.train_dtype,
        "seconds": times["train_s"], "ms_per_step": times["train_s"] / max(len(sched), 1) * 1e3,
        "edges_scored_per_s": 2.0 * etb_seen / times["train_s"],
        "mp_edges_per_s": 4.0 * (ett_seen + etb_seen) / times["train_s"],
        "final_loss": float(loss.detach()) if loss is not None else None,
        "epochs_s": epochs_s, "ms_per_step_by_epoch": [e / max(len(sched), 1) * 1e3 for e in epochs_s],
        "graphed": graphed_train, "graph_buckets": None if trainer is None else [b.sizes for b in trainer.buckets],
    }
    if rank != 0:                                            # scoring / prediction: rank 0 only
        dist.destroy_process_group()
        return

    # ---- 2. score every candidate edge, one pass per dtype, identical weights ------------------------
    model.eval()
    ep_sizes = part.edge_sizes[TX_NB_BD].tolist()
    ep_total = sum(ep_sizes[t] for ids in batches for t in ids)      # tiles over the batch budget are skipped
    labels = torch.empty(ep_total, dtype=torch.bool, device=dev)
    scoring = {}
    ref_scores = None
    for name in [s for s in args.score_dtypes.split(",") if s]:
        model.model.compute_dtype = DT[name]
        scores = torch.empty(ep_total, dtype=torch.float32, device=dev)
        hit = tot = 0.0
        with torch.no_grad(), Phase(f"score_{name}_s", times):
            o = 0
            for ids in batches:
                b = part.batch(ids)
                z = model.forward(b)
                ei = b[TX_NB_BD].edge_index
                g = edge_graph(batch_cache(b), TX_NB_BD, ei, b["tx"].num_nodes, b["bd"].num_nodes, need_by_dst=False)
                _, _, seg, sim = ops.edge_cos_argmax(g.by_src, z["tx"], z["bd"], dst_index=b["bd"]["index"],
                                                     return_sim=True)
                e = int(ei.shape[1])
                scores[o:o + e] = sim
                lab = b["bd"]["index"][ei[1]].long() == b["tx"]["cell"][ei[0]]
                labels[o:o + e] = lab
                has = torch.zeros(b["tx"].num_nodes, dtype=torch.bool, device=dev)
                has[ei[0][lab]] = True
                hit += float(((seg == b["tx"]["cell"]) & has).sum())
                tot += float(has.sum())
                o += e
        a = auroc(scores, labels)
        scoring[name] = {"auroc": a, "assignment_accuracy": hit / max(tot, 1.0), "seconds": times[f"score_{name}_s"],
                         "edges_scored_per_s": ep_total / times[f"score_{name}_s"]}
        if name == "f32":
            ref_scores = scores
        elif ref_scores is not None:
            scoring[name]["max_abs_score_diff_vs_f32"] = float((scores - ref_scores).abs().max())
            scoring[name]["auroc_diff_vs_f32"] = abs(a - scoring["f32"]["auroc"])
        log(f"[fov] {name}: {scoring[name]}")
    del ref_scores

    # ---- 3. hipGraph-captured predictor (config 5) ---------------------------------------------------
    graphed = None
    if args.graphed:
        model.model.compute_dtype = torch.float16
        from segger_amd.inference import GraphedPredictorPool
        pool = GraphedPredictorPool(model, spec.bd_dim)
        n_out = 0
        outs = []
        for warm in (True, False):                           # first sweep captures one graph per bucket
            with Phase("graphed_capture_s" if warm else "graphed_predict_s", times):
                dev_out = [pool.predict_device(part.batch(ids)) for ids in batches]    # no sync inside the loop
                mask = torch.cat([o[4] for o in dev_out])
                outs = [tuple(torch.cat([o[i] for o in dev_out])[mask] for i in range(4))]
                n_out = int(outs[0][0].numel())              # (one compaction + one sync for the whole sweep)
                del dev_out, mask
        graphed = {"dtype": "f16", "buckets": len(pool.buckets), "capture_sweep_s": times["graphed_capture_s"],
                   "seconds": times["graphed_predict_s"], "transcripts_out": n_out,
                   "edges_scored_per_s": ep_total / times["graphed_predict_s"],
                   "note": "predict_device() per batch (one staging launch + graph replay), one mask compaction for the sweep; "
                           "outputs stay on the device for the post-processing"}
        # the step after the path: best row per transcript + per-gene Yen / Li thresholds, on the device
        from segger_amd.postprocess import assign_transcripts_to_cells
        with Phase("postprocess_s", times):
            seg = assign_transcripts_to_cells(outs, device=dev)
        thr = seg["similarity_threshold"]
        graphed["postprocess"] = {
            "seconds": times["postprocess_s"], "rows": int(seg["row_index"].numel()),
            "assigned": int((seg["cell_encoding"] >= 0).sum()),
            "above_threshold": int(((seg["cell_encoding"] >= 0) & (seg["similarity"].double() >= thr)).sum()),
            "global_threshold": seg["global_threshold"], "failed_genes": int(seg["failed_genes"].numel())}
        del outs, seg
        log(f"[fov] graphed: {graphed}")

    # ---- 4. overlapping prediction tiles -> predict_step -> dedup + thresholds ---------------------------
    overlap = None
    if pti is not None:
        from segger_amd.postprocess import assign_transcripts_to_cells
        model.model.compute_dtype = torch.float16
        outs = []
        with Phase("overlap_predict_s", times):
            for i in range(len(pti)):
                outs.append(model.predict_step(pti[i], i))
        rows = sum(int(o[0].numel()) for o in outs)
        # the same sweep through the hipGraph predictor pool: no per-tile sync, one mask compaction at the end
        from segger_amd.inference import GraphedPredictorPool
        opool = GraphedPredictorPool(model, spec.bd_dim)
        for phase in ("overlap_graphed_capture_s", "overlap_graphed_predict_s"):
            with Phase(phase, times):
                dev_out = [opool.predict_device(pti[i]) for i in range(len(pti))]
                mask = torch.cat([o[4] for o in dev_out])
                outs_g = [tuple(torch.cat([o[i] for o in dev_out])[mask] for i in range(4))]
                rows_g = int(outs_g[0][0].numel())
                del dev_out, mask
        same = rows_g == rows and all(
            torch.equal(torch.cat([o[i] for o in outs]).to(dev), outs_g[0][i]) for i in (0, 3))
        del outs_g
        with Phase("overlap_postprocess_s", times):
            seg = assign_transcripts_to_cells(outs, device=dev)
        overlap = {"tiles": len(pti), "margin_um": args.margin, "dtype": "f16", "predict_seconds": times["overlap_predict_s"],
                   "rows_from_tiles": rows, "transcripts": int(seg["row_index"].numel()),
                   "assigned": int((seg["cell_encoding"] >= 0).sum()),
                   "postprocess_seconds": times["overlap_postprocess_s"], "global_threshold": seg["global_threshold"],
                   "transcripts_per_s": int(seg["row_index"].numel()) / times["overlap_predict_s"],
                   "graphed": {"predict_seconds": times["overlap_graphed_predict_s"],
                               "capture_sweep_seconds": times["overlap_graphed_capture_s"], "buckets": len(opool.buckets),
                               "same_rows_as_eager": bool(same),
                               "transcripts_per_s": int(seg["row_index"].numel()) / times["overlap_graphed_predict_s"]}}
        if args.segmentation_parquet:
            from segger_amd.postprocess import to_frame
            with Phase("write_parquet_s", times):
                df = to_frame(seg)
                try:
                    df.to_parquet(args.segmentation_parquet, index=False)
                    overlap["parquet"] = args.segmentation_parquet
                except ImportError as e:                      # no pyarrow / fastparquet on this host
                    alt = os.path.splitext(args.segmentation_parquet)[0] + ".npz"
                    import numpy as np
                    np.savez(alt, **{c: df[c].to_numpy(dtype="float64" if c == "segger_cell_id" else None, na_value=np.nan)
                                     if c == "segger_cell_id" else df[c].to_numpy() for c in df.columns})
                    overlap["parquet"] = f"{alt} (no parquet engine: {type(e).__name__})"
        del outs, seg
        log(f"[fov] overlap predict: {overlap}")

    res = {
        "workload": f"synthetic FOV: {args.n_tx} tx, {args.n_bd} nuclei, k={args.k}; {len(tiling)} square tiles "
                    f"(~{args.tile_nodes} tx), {len(batches)} batches of <= {args.edges_per_batch} edges",
        "edges_total": n_edges, "edges_intra_tile": kept, "phases_s": times,
        "train": train, "scoring": scoring, "graphed_predict": graphed, "overlap_predict": overlap,
        "peak_hbm_gib": torch.cuda.max_memory_allocated() / 2 ** 30,
    }
    if world > 1:
        dist.destroy_process_group()
    s = json.dumps(res)
    print(s, flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
