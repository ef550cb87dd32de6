"""Worker of test_gpu_train_graph.py::test_graphed_trainer_data_parallel: two ranks (gloo) sharing cuda:0, uneven
batch counts, the captured step split around the gradient exchange.  Prints one JSON line per rank."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from segger_amd import tiles as T
    from segger_amd.dp import FlatGradBucket, broadcast_parameters, rank_schedule, seed_rank
    from segger_amd.synthetic import SyntheticSpec
    from segger_amd.train_step_graph import GraphedTrainer
    from tests.test_gpu_model import build
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    spec = SyntheticSpec(n_tx=24000, n_bd=700, k_tx=6, seed=43)
    m, _, b, _ = build(spec, dev, dtype=torch.bfloat16)
    m.train()
    m._max_epochs_override, m.current_epoch = 20, 12
    bg = b.to(dev)
    for nt in ("tx", "bd"):
        del bg[nt]["mask"]
    tiling = T.SquareTiling(torch.cat([bg["tx"].pos, bg["bd"].pos]).cpu(), 60.0)
    part = T.partition_by_tiling(bg, tiling, margin=3.0)
    part.build_csr()
    sampler = T.TileBatchSampler(part, max_num=max(part.weights("edge")) * 3, mode="edge", skip_too_big=True)
    batches = [ids for ids in sampler if all(part.node_sizes["bd"][t] > 1 for t in ids)]
    batches = batches[: 2 * (len(batches) // 2) - 1]              # an odd count: one rank takes an empty step
    w = part.weights("edge")
    sched = rank_schedule([sum(w[t] for t in ids) for ids in batches], world)[rank]
    broadcast_parameters(m)
    seed_rank(7, rank, m.model)
    opt = m.configure_optimizers(capturable=True)
    bucket = FlatGradBucket(m.parameters())
    trainer = GraphedTrainer(m, opt, granularity=1.4, grad_bucket=bucket)
    losses = []
    epochs = 4
    bucket._ensure()
    flat_ptr0 = bucket.flat.data_ptr()
    for ep in range(epochs):
        for k in sched:
            out = trainer.step(part.batch(batches[k]) if k is not None else None)
            if out is not None:
                losses.append(float(out[3]))
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    steps = float(next(iter(opt.state.values()))["step"])
    rec = json.dumps({"rank": rank, "own": sum(k is not None for k in sched), "steps_per_epoch": len(sched),
                      "adam_steps": steps, "epochs": epochs, "captures": trainer.n_captures,
                      "finite": bool(torch.isfinite(flat).all()), "flat_ptr_stable": bucket.flat.data_ptr() == flat_ptr0,
                      "grads_are_views": all(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                                             for p, v in zip(bucket.params, bucket.views)),
                      "max_param_diff": max(float((g - flat).abs().max()) for g in gathered),
                      "first": sum(losses[: len(losses) // epochs]), "last": sum(losses[-(len(losses) // epochs):])})
    out_dir = os.environ.get("DP_WORKER_OUT")       # one file per rank: two ranks' stdout lines can interleave
    if out_dir:
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
            fh.write(rec)
    print(rec, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
