"""The N>1 path on the CPU: world_size-2 gloo processes exercise the flat gradient bucket
(all-reduce mean, parameter broadcast) that bench.py / training use over RCCL on the GPUs."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segger_amd import LitISTEncoder
    from segger_amd.dp import FlatGradBucket, broadcast_parameters
    torch.manual_seed(100 + rank)                       # ranks start from different weights
    m = LitISTEncoder(n_genes=12, in_channels=16, hidden_channels=8, out_channels=8)
    m.model._materialize_bd(6, "cpu")
    broadcast_parameters(m, src=0)
    w0 = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    gathered = [torch.empty_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    same_weights = all(torch.equal(gathered[0], g) for g in gathered)
    bucket = FlatGradBucket(m.parameters())
    # stand-in for backward: rank-dependent gradients; one parameter gets no gradient on rank 1
    for i, p in enumerate(bucket.params):
        p.grad = None if (rank == 1 and i == 3) else torch.full_like(p, float((rank + 1) * (i + 1)))
    views_ok = True
    bucket.all_reduce_mean()
    want = sum(r + 1 for r in range(world)) / world
    grads_ok = all(
        torch.allclose(p.grad, torch.full_like(p, (1.0 / world if i == 3 else want) * (i + 1)))
        for i, p in enumerate(bucket.params))
    reattach_ok = bucket.flat.numel() == bucket.numel and all(p.grad is not None for p in bucket.params)
    opt = m.configure_optimizers()
    opt.step()
    w1 = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    dist.all_gather(gathered, w1)
    in_sync = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        out.put((same_weights, views_ok, grads_ok, reattach_ok, in_sync, bucket.flat.numel()))
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    same_weights, views_ok, grads_ok, reattach_ok, in_sync, n = res
    assert same_weights and views_ok and grads_ok and reattach_ok and in_sync and n > 1000


def test_single_process_is_a_noop():
    from segger_amd.dp import FlatGradBucket
    lin = torch.nn.Linear(3, 2)
    b = FlatGradBucket(lin.parameters())
    lin(torch.ones(1, 3)).sum().backward()
    before = [p.grad.clone() for p in lin.parameters()]
    b.all_reduce_mean()                                  # no process group: nothing happens
    assert all(torch.equal(x, p.grad) for x, p in zip(before, lin.parameters())) and b.flat.numel() == 0


def _uneven_worker(rank, world, port, out):
    """5 batches over 2 ranks -> 3 steps; the rank that runs out takes an EMPTY step (zeros into the all-reduce)."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segger_amd.dp import FlatGradBucket, broadcast_parameters, rank_schedule, seed_rank
    torch.manual_seed(5)
    lin = torch.nn.Linear(4, 3)
    broadcast_parameters(lin)
    seed_rank(123, rank)
    draw = torch.rand(1).item()                             # rank-offset RNG stream
    data = [torch.full((2, 4), float(i + 1)) for i in range(5)]
    weights = [10, 9, 8, 7, 6]
    sched = rank_schedule(weights, world)
    bucket = FlatGradBucket(lin.parameters())
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    for b in sched[rank]:
        opt.zero_grad(set_to_none=True)
        if b is not None:
            lin(data[b]).sum().backward()
        bucket.all_reduce_mean()
        opt.step()
    w = torch.cat([p.detach().reshape(-1) for p in lin.parameters()])
    gathered = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    draws = [None] * world
    dist.all_gather_object(draws, draw)
    if rank == 0:
        out.put((sched, torch.equal(gathered[0], gathered[1]), w.tolist(), draws))   # plain lists: no fd hand-over
    dist.destroy_process_group()


def test_uneven_batch_counts_take_empty_steps_world2():
    from segger_amd.dp import rank_schedule
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    sched, in_sync, w, draws = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sched == [[0, 3, 4], [1, 2, None]] == rank_schedule([10, 9, 8, 7, 6], 2)
    assert in_sync and draws[0] != draws[1]
    # single-process replay of the same 3 averaged steps
    torch.manual_seed(5)
    lin = torch.nn.Linear(4, 3)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    data = [torch.full((2, 4), float(i + 1)) for i in range(5)]
    for pair in zip(*sched):
        opt.zero_grad()
        for b in pair:
            if b is not None:
                (lin(data[b]).sum() / 2).backward()          # mean over the world size, empty rank = zeros
        opt.step()
    assert torch.allclose(torch.tensor(w), torch.cat([p.detach().reshape(-1) for p in lin.parameters()]), atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------
# bench.py's strong-scaling record (BASELINE config 4) as a 2-rank gloo dry run: the schedule, the empty steps, the
# rank census and the unit / time reductions of dp.strong_scaling_epoch, with a CPU stand-in for the training step
def _strong_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segger_amd.dp import FlatGradBucket, strong_scaling_epoch
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 3)
    bucket = FlatGradBucket(lin.parameters())
    weights = [5.0, 3.0, 9.0, 1.0, 2.0]                     # 5 batches over 2 ranks: LPT -> {2, 4 | ...}: uneven counts
    ran = []

    def step(k, i):
        lin.zero_grad(set_to_none=True)
        if k is not None:
            ran.append(k)
            x = torch.full((2, 4), float(k + 1))
            lin(x).sum().backward()
        bucket.all_reduce_mean()                            # every rank, every step (empty steps send zeros)

    rec = strong_scaling_epoch(weights, step, lambda k: (weights[k], 1.0), warmup=1)
    gathered = [None] * world
    dist.all_gather_object(gathered, ran)
    if rank == 0:
        out.put((rec, gathered))
    dist.destroy_process_group()


def test_strong_scaling_epoch_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rec, ran = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert rec["scaling"] == "strong" and rec["world_size"] == 2 and rec["n_ranks_seen"] == 2 and rec["census"] == [1, 1]
    assert rec["batches"] == 5 and rec["steps_per_rank"] == 3
    assert rec["units_total"] == [20.0, 5.0]                # every batch counted exactly once over the two ranks
    timed = [r[1:] for r in ran]                            # the first entry of each rank is its warm-up step
    assert sorted(k for r in timed for k in r) == [0, 1, 2, 3, 4]
    assert abs(sum([5, 3, 9, 1, 2][k] for k in timed[0]) - sum([5, 3, 9, 1, 2][k] for k in timed[1])) <= 2   # balanced
    assert rec["epoch_s"] > 0


def test_strong_scaling_epoch_single_process():
    from segger_amd.dp import strong_scaling_epoch
    seen = []
    rec = strong_scaling_epoch([2.0, 1.0, 4.0], lambda k, i: seen.append(k), lambda k: (1.0,))
    assert seen == [2, 0, 1] and rec["units_total"] == [3.0] and rec["n_ranks_seen"] == 1 and rec["steps_per_rank"] == 3


# ---------------------------------------------------------------------------------------------------------------
# round 3: shard residency, the persistent flat gradient buffer, and the collective vote before a phase with collectives
def _round3_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segger_amd import tiles as T
    from segger_amd.dp import FlatGradBucket, all_agree, broadcast_parameters, rank_schedule
    from segger_amd.hetero import TX_BD, TX_TX
    from segger_amd.synthetic import SyntheticSpec, make_graph
    res = {}
    # (a) every rank builds the same graph, partitions it, and keeps only the tiles of its own batches
    g = make_graph(SyntheticSpec(n_tx=4000, n_bd=120, k_tx=6, seed=9))
    for nt in ("tx", "bd"):
        del g[nt]["mask"]
    part = T.partition_by_tiling(g, T.SquareTiling(torch.cat([g["tx"].pos, g["bd"].pos]), 22.0), margin=2.0)
    sampler = T.TileBatchSampler(part, max(part.weights("edge")) * 2, mode="edge", skip_too_big=True)
    batches = list(sampler)
    w = part.weights("edge")
    weights = [float(sum(w[t] for t in ids)) for ids in batches]
    mine = [k for k in rank_schedule(weights, world)[rank] if k is not None]
    tiles_mine = sorted({t for k in mine for t in batches[k]})
    remap = {t: i for i, t in enumerate(tiles_mine)}
    local = part.shard(tiles_mine)
    same = True
    for k in mine:
        a, b = local.batch([remap[t] for t in batches[k]]), part.batch(batches[k])
        same &= all(torch.equal(a[nt][key], b[nt][key]) for nt in ("tx", "bd") for key in ("x", "pos", "index", "mask", "batch"))
        same &= all(torch.equal(a[et].edge_index, b[et].edge_index) for et in (TX_TX, TX_BD))
    res["shard_batches_equal"] = bool(same)
    res["resident_fraction"] = local.resident_bytes() / part.resident_bytes()
    res["tiles"] = tiles_mine
    # (b) persistent flat buffer: same address every step, .grad = its slices, values = the mean over the ranks
    torch.manual_seed(3)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 2))
    broadcast_parameters(lin)
    bucket = FlatGradBucket(lin.parameters())
    ptrs, views_ok, mean_ok = [], True, True
    for step in range(3):
        lin.zero_grad(set_to_none=True)
        x = torch.full((3, 5), float(rank + 1 + step))
        lin(x).sum().backward()
        own = [p.grad.clone() for p in lin.parameters()]
        bucket.all_reduce_mean()
        ptrs.append(bucket.flat.data_ptr())
        views_ok &= all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        gathered = [None] * world
        dist.all_gather_object(gathered, [o.tolist() for o in own])
        want = [sum(torch.tensor(gr[i]) for gr in gathered) / world for i in range(len(own))]
        mean_ok &= all(torch.allclose(p.grad, w_) for p, w_ in zip(lin.parameters(), want))
    bucket.zero()                                           # an empty step: zeros in, .grad still the slices
    res["flat_ptr_stable"] = len(set(ptrs)) == 1 and bucket.flat.data_ptr() == ptrs[0]
    res["grads_are_views"], res["mean_ok"] = bool(views_ok), bool(mean_ok)
    res["zero_ok"] = bool((bucket.flat == 0).all()) and all(p.grad is not None for p in lin.parameters())
    # (c) vote: a rank that failed its collective-free pre-flight makes EVERY rank skip the phase
    try:
        if rank == 1:
            raise RuntimeError("pre-flight failed on this rank")
        ok = True
    except RuntimeError:
        ok = False
    entered = all_agree(ok)
    if entered:                                             # would deadlock if only some ranks came here
        dist.all_reduce(torch.ones(1))
    res["entered_after_no_vote"] = entered
    res["agree_when_all_ok"] = all_agree(True)
    alive = torch.ones(1)
    dist.all_reduce(alive)                                  # both ranks are still in step with each other
    res["alive"] = float(alive)
    allres = [None] * world
    dist.all_gather_object(allres, res)
    if rank == 0:
        out.put(allres)
    dist.destroy_process_group()


def test_shard_residency_persistent_bucket_and_vote_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_round3_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r["shard_batches_equal"] for r in res)
    assert not set(res[0]["tiles"]) & set(res[1]["tiles"])                  # disjoint tile sets ...
    assert abs(res[0]["resident_fraction"] + res[1]["resident_fraction"] - 1.0) < 0.15   # ... that split the bytes
    assert all(0.25 < r["resident_fraction"] < 0.75 for r in res)           # about 1 / world each
    for r in res:
        assert r["flat_ptr_stable"] and r["grads_are_views"] and r["mean_ok"] and r["zero_ok"]
        assert r["entered_after_no_vote"] is False and r["agree_when_all_ok"] is True and r["alive"] == 2.0


def _disagree_worker(rank, world, port, out, same_count=False):
    """Ranks that hold DIFFERENT batch lists (a bug upstream) must fail loudly before the first step, not deadlock in a
    mismatched collective half-way through the epoch."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segger_amd.dp import strong_scaling_epoch
    weights = [5.0, 4.0, 3.0, 2.0] if rank == 0 else [5.0, 4.0, 3.0, 2.0, 1.0, 1.0]      # 2 vs 3 steps per rank
    if same_count:      # same number of batches and steps, different weights (rank-local tile counts that disagree)
        weights = [5.0, 4.0, 3.0, 2.0] if rank == 0 else [5.0, 4.0, 2.0, 3.0]
    steps = []
    try:
        strong_scaling_epoch(weights, lambda k, i: steps.append(k), units=None)
        res = "no error"
    except RuntimeError as e:
        res = str(e)
    out.put((rank, res, len(steps)))
    dist.destroy_process_group()


@pytest.mark.parametrize("same_count", [False, True])
def test_strong_epoch_refuses_ranks_that_disagree_on_the_schedule(same_count):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_disagree_worker, args=(r, 2, port, q, same_count)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res, n_steps in got:
        assert "do not agree on the batch list" in res and n_steps == 0, (rank, res)


def _run_bench(*argv, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus_n_launches_its_own_ranks_world2():
    """`python bench.py --gpus 2` with no launcher environment (the driver's command shape) must start two ranks itself:
    the dry launch joins a gloo group and all-reduces the census; stdout is exactly one JSON line."""
    import json
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["census"] == [1, 1] and rec["n_ranks_seen"] == 2
    assert rec["launched_by"] == "torch.distributed.run"


def test_bench_refuses_to_report_fewer_gpus_than_asked():
    """No GPU here: `--gpus 2` (RCCL) must exit non-zero with a message, never print an n_gpus: 1 line."""
    r = _run_bench("--gpus", "2", timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == "" and "GPU(s) visible" in r.stderr


def test_bench_refuses_a_launcher_whose_world_differs_from_gpus():
    r = _run_bench("--gpus", "4", "--dry-launch", env_extra={"WORLD_SIZE": "1", "RANK": "0"}, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE=1" in r.stderr
