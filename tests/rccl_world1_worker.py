"""Worker of test_gpu_train_graph.py::test_rccl_one_rank_communicator: RCCL itself on a one-GPU box.

A one-rank ``nccl`` process group (``device_id`` given, so the communicator is created eagerly on this GPU) runs
every collective ``bench.py`` / ``dp.py`` issue at N > 1, with the same dtypes and reduce ops, and the captured
training step in its data-parallel form -- forward + backward + pack graph | all-reduce + divide on the persistent
flat buffer, eagerly | Adam graph -- with the collective FORCED (a world-1 mean is the identity).  Nothing touches
the GPU before ``init_process_group``.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.cuda.set_device(dev)
    rec = {"backend": dist.get_backend(), "world": dist.get_world_size()}

    # ---- the collectives of bench.py / dp.py, same dtypes and ops -------------------------------------------------
    t = torch.tensor([1.5, 2.0, 3.0], dtype=torch.float64, device=dev)
    a = t.clone(); dist.all_reduce(a, op=dist.ReduceOp.MAX)                 # bench.py: max over ranks of the step time
    b = t.clone(); dist.all_reduce(b, op=dist.ReduceOp.SUM)                 # bench.py: unit sums
    v = torch.tensor([1], dtype=torch.int32, device=dev); dist.all_reduce(v, op=dist.ReduceOp.MIN)    # dp.all_agree
    c = torch.zeros(1, dtype=torch.int32, device=dev); c[0] = 1; dist.all_reduce(c, op=dist.ReduceOp.SUM)   # dp.rank_census
    f = torch.arange(1000, dtype=torch.float32, device=dev); f0 = f.clone(); dist.broadcast(f, src=0)    # dp.broadcast_parameters
    dist.barrier()
    torch.cuda.synchronize()
    rec["collectives_identity"] = bool(torch.equal(a, t) and torch.equal(b, t) and int(v) == 1 and int(c) == 1
                                       and torch.equal(f, f0))

    from segger_amd import tiles as T
    from segger_amd.dp import FlatGradBucket, restore_training_state, snapshot_training_state
    from segger_amd.synthetic import SyntheticSpec
    from segger_amd.train_step_graph import GraphedTrainer
    from tests.test_gpu_model import build
    spec = SyntheticSpec(n_tx=24000, n_bd=700, k_tx=6, seed=43)
    m, _, bcpu, _ = build(spec, dev, dtype=torch.bfloat16)
    m.train()
    m._max_epochs_override, m.current_epoch = 20, 12
    bg = bcpu.to(dev)
    for nt in ("tx", "bd"):
        del bg[nt]["mask"]
    tiling = T.SquareTiling(torch.cat([bg["tx"].pos, bg["bd"].pos]).cpu(), 60.0)
    part = T.partition_by_tiling(bg, tiling, margin=3.0)
    part.build_csr()
    sampler = T.TileBatchSampler(part, max_num=max(part.weights("edge")) * 3, mode="edge", skip_too_big=True)
    batches = [ids for ids in sampler if all(part.node_sizes["bd"][t] > 1 for t in ids)]
    steps = [batches[i % len(batches)] for i in range(20)]

    # ---- eager step through the forced one-rank all-reduce: gradients come back bit-identical ----------------------
    opt = m.configure_optimizers(capturable=True)
    snap = snapshot_training_state(m, opt)
    step0 = m.model._step_dev.clone()

    class Checked(FlatGradBucket):
        """Records whether a forced exchange ever changed a bit of the flat buffer (it must not: sum over one rank, / 1)."""
        changed, calls = 0, 0

        def all_reduce_mean(self, group=None, packed=False, force=None):
            if not packed:
                self.pack()
            before = self.flat.clone()
            super().all_reduce_mean(group, packed=True, force=force)
            Checked.changed += int((self.flat.view(torch.int32) != before.view(torch.int32)).sum())
            Checked.calls += 1

    bucket = Checked(m.parameters(), force_collective=True)
    opt.zero_grad(set_to_none=True)
    m.training_step(part.batch(steps[0]), 0).backward()
    bucket.all_reduce_mean()
    opt.step()
    torch.cuda.synchronize()
    rec["eager_calls"], rec["eager_bits_changed"] = Checked.calls, Checked.changed
    rec["grads_are_views"] = all(p.grad is not None and p.grad.data_ptr() == v_.data_ptr()
                                 for p, v_ in zip(bucket.params, bucket.views))
    restore_training_state(m, snap, opt)
    with torch.no_grad():
        m.model._step_dev.copy_(step0)

    # ---- 20 captured steps: two graphs around the forced all-reduce vs the single-graph trainer ---------------------
    def run(split: bool):
        restore_training_state(m, snap, opt)
        with torch.no_grad():
            m.model._step_dev.copy_(step0)
        Checked.changed = Checked.calls = 0
        tr = GraphedTrainer(m, opt, granularity=1.4, grad_bucket=Checked(m.parameters(), force_collective=True) if split else None)
        losses = [float(tr.step(part.batch(ids))[3]) for ids in steps]
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).clone()
        kinds = sorted({(b.graph is not None, b.graph_opt is not None) for b in tr.buckets})
        return losses, flat, float(next(iter(opt.state.values()))["step"]), kinds, tr.n_captures
    l_split, p_split, n_split, kinds_split, cap_split = run(True)
    rec["split_calls"], rec["split_bits_changed"] = Checked.calls, Checked.changed
    l_one, p_one, n_one, kinds_one, _ = run(False)
    rec.update({
        "split_two_graphs": kinds_split == [(True, True)], "single_graph": kinds_one == [(True, False)],
        "captures": cap_split, "adam_steps": [n_split - n_one, n_one],
        "finite": bool(torch.isfinite(p_split).all() and torch.isfinite(p_one).all()),
        # the loss kernels add with float atomics (order varies run to run), so two runs of ONE trainer already differ in the
        # last bits; the two routes must agree like two runs do
        "max_param_diff": float((p_split - p_one).abs().max()), "param_scale": float(p_one.abs().max()),
        "max_loss_diff": max(abs(x - y) for x, y in zip(l_split, l_one)),
        "loss_first_last": [l_split[0], l_split[-1]]})
    print(json.dumps(rec), flush=True)
    out = os.environ.get("RCCL_WORKER_OUT")
    if out:
        with open(out, "w") as fh:
            fh.write(json.dumps(rec))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
