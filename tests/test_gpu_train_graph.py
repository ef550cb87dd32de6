"""Whole-step hipGraph (segger_amd.train_step_graph): the padded, captured step must compute the gradients of the eager
step on the real batch, must not count its warm-up as a training step, and must draw afresh on every replay."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(spec, dev, dtype, **kw):
    from tests.test_gpu_model import build
    m, _, b, _ = build(spec, dev, dtype=dtype, **kw)
    m.train()
    m._max_epochs_override, m.current_epoch = 20, 12
    return m, b.to(dev)


def _fixed_draws(m, bg):
    tx, bd = bg["tx"], bg["bd"]
    from segger_amd.hetero import TX_BD
    pos, neg, _, _ = m.loss_tx.selector.sample_triplets(tx["cluster"], mask=tx["mask"])
    bmask = bd["mask"] & (bd["cluster"] >= 0)
    bpos, bneg, dp, dn = m.loss_bd.selector.sample_triplets(bd["cluster"], mask=bmask)
    dst = bg[TX_BD].edge_index[1]
    n_bd = bd.num_nodes
    dst_neg = (dst + torch.randint(1, n_bd, dst.shape, device=dst.device)) % n_bd
    return dict(tx=(pos, neg), bd=(bpos, bneg, dp, dn), dst_neg=dst_neg, bmask=bmask)


def _eager_grads(m, bg, d):
    """The eager step's loss (lightning_model.get_losses) with the draws given instead of sampled."""
    from segger_amd import ops
    from segger_amd.hetero import TX_BD
    m.zero_grad(set_to_none=True)
    z = m(bg)
    n = bg["tx"].num_nodes
    l_tx = ops.triplet_edge_loss(z["tx"], None, torch.arange(n, device=z["tx"].device), *d["tx"], m.loss_tx.margin,
                                 eps=m.loss_tx.eps) * (n / bg["tx"]["mask"].sum().float())
    l_bd = ops.metric_loss(z["bd"], *d["bd"], d["bmask"].float() / d["bmask"].sum().float())
    l_sg = m._segmentation_loss(z, bg, d["dst_neg"])
    w = m._scheduled_weights(m._w_start, m._w_end)
    loss = float(w[0]) * l_tx + float(w[1]) * l_bd + float(w[2]) * l_sg
    loss.backward()
    return torch.stack([l_tx, l_bd, l_sg, loss]).detach().float(), {k: p.grad.clone() for k, p in m.named_parameters()}


def _pad_draws(step, d):
    s = step.sizes
    def pad(t, n, fill):
        return torch.cat([t, torch.full((n - t.numel(),), fill, dtype=t.dtype, device=t.device)])
    return dict(tx=tuple(pad(t, s["tx"], -1) for t in d["tx"]),
                bd=tuple(pad(t, s["bd"], -1 if i < 2 else 0) for i, t in enumerate(d["bd"])),
                dst_neg=pad(d["dst_neg"], s["e_tb"], -1))


@pytest.mark.parametrize("sg_loss_type", ["triplet", "bce"])
@pytest.mark.parametrize("capture", [False, True])
def test_graphed_step_gradients_equal_eager_step(cuda, capture, sg_loss_type):
    from segger_amd.synthetic import SyntheticSpec
    from segger_amd.train_step_graph import GraphedTrainStep, step_bucket
    spec = SyntheticSpec(n_tx=5000, n_bd=170, k_tx=7, seed=31)
    m, bg = _model(spec, cuda, torch.float32, sg_loss_type=sg_loss_type)
    torch.manual_seed(5)
    d = _fixed_draws(m, bg)
    m2 = copy.deepcopy(m)
    m2.set_similarities(m.loss_tx.selector.similarity, m.loss_bd.selector.similarity)
    ref_losses, ref = _eager_grads(m, bg, d)              # dropout stream: counter 0 -> 256 in both models
    before = {k: p.detach().clone() for k, p in m2.named_parameters()}
    opt = m2.configure_optimizers(capturable=True)
    step = GraphedTrainStep(m2, opt, step_bucket(bg, granularity=1.3), bg)
    step.draws = _pad_draws(step, d)
    out = step.step(bg, capture=capture).clone()
    assert int(m2.model._step_dev) == 256                  # the warm-up run was rolled back: ONE step happened
    assert torch.allclose(out, ref_losses, rtol=1e-4, atol=1e-6)
    moved = 0
    for k, p in m2.named_parameters():
        g, r = p.grad, ref[k]
        assert (g - r).abs().max().item() <= 2e-4 * r.abs().max().item() + 1e-7, k
        # one Adam step from zero state moves every weight with a gradient by ~lr, the others not at all
        delta = (p.detach() - before[k]).abs()
        assert delta.max().item() <= 1.01 * m2.learning_rate
        moved += int((delta > 0).sum())
    assert moved > 0
    st = opt.state[next(iter(m2.parameters()))]
    assert float(st["step"]) == 1.0


def test_graphed_trainer_replays_learn_and_draw_afresh(cuda):
    from segger_amd import tiles as T
    from segger_amd.synthetic import SyntheticSpec
    from segger_amd.train_step_graph import GraphedTrainer
    spec = SyntheticSpec(n_tx=30000, n_bd=900, k_tx=6, seed=37)
    m, bg = _model(spec, cuda, torch.bfloat16)
    for nt in ("tx", "bd"):
        del bg[nt]["mask"]                                  # the tiling assigns its own margin mask
    tiling = T.SquareTiling(torch.cat([bg["tx"].pos, bg["bd"].pos]).cpu(), 60.0)
    part = T.partition_by_tiling(bg, tiling, margin=3.0)
    part.build_csr()
    sampler = T.TileBatchSampler(part, max_num=max(part.weights("edge")) * 4, mode="edge", skip_too_big=True)
    batches = [ids for ids in sampler if all(part.node_sizes["bd"][t] > 1 for t in ids)]
    assert len(batches) >= 3
    opt = m.configure_optimizers(capturable=True)
    # an eager step first, its loss (and with it the autograd graph and the parameters' AccumulateGrad nodes, bound to
    # the default stream) kept alive: the capture must not depend on those nodes
    stale = m.training_step(part.batch(batches[0]), 0)
    stale.backward()
    opt.step()
    trainer = GraphedTrainer(m, opt, granularity=1.5)
    first, last = [], []
    for ep in range(8):
        for ids in batches:
            out = trainer.step(part.batch(ids)).clone()
            assert torch.isfinite(out).all()
            if ep in (0, 7):
                (first if ep == 0 else last).append(float(out[3]))
    assert trainer.n_captures < len(batches) * 8            # replays, not a capture per step
    assert sum(last) < sum(first)          # it learns
    # same batch, same weights cannot be arranged (the step trains); but the sampler stream must move: two replays of
    # one batch give different loss_tx AND the dropout counter advanced once per step
    a = trainer.step(part.batch(batches[0])).clone()
    b = trainer.step(part.batch(batches[0])).clone()
    assert not torch.equal(a, b)
    assert int(m.model._step_dev) == 256 * (8 * len(batches) + 3)
    # eager evaluation after graphed training sees the trained weights (cache invalidated after every replay)
    m.eval()
    with torch.no_grad():
        z = m(part.batch(batches[0]))
    assert torch.isfinite(z["tx"].float()).all()


def test_graphed_training_under_manual_optimisation(cuda):
    """``LitISTEncoder.enable_graphed_training()``: ``training_step`` itself replays the captured step (Lightning's manual
    optimisation: the trainer runs neither backward nor optimizer.step) -- driven here the way ``Trainer.fit`` drives a
    module with ``automatic_optimization = False``; parameters and losses equal a ``GraphedTrainer`` used directly."""
    import types
    from segger_amd.synthetic import SyntheticSpec
    from segger_amd.train_step_graph import GraphedTrainer
    spec = SyntheticSpec(n_tx=20000, n_bd=600, k_tx=6, seed=41)
    m, bg = _model(spec, cuda, torch.bfloat16)
    twin, init = copy.deepcopy(m), copy.deepcopy(m)
    # the reference's flow: Trainer.fit -> configure_optimizers -> training_step per batch
    assert m.automatic_optimization
    m.enable_graphed_training(granularity=1.5)
    assert not m.automatic_optimization
    opt = m.configure_optimizers()
    assert all(g["capturable"] for g in opt.param_groups)
    m.trainer = types.SimpleNamespace(optimizers=[opt], max_epochs=20, datamodule=None)
    direct = GraphedTrainer(twin, twin.configure_optimizers(capturable=True), granularity=1.5)
    for i in range(4):
        loss = m.training_step(bg, i)
        ref = direct.step(bg).clone()
        # the first step is the same arithmetic on the same weights; later ones differ by the order of the loss
        # gradients' atomic adds (and Adam turns a last-bit difference of a near-zero gradient into +-lr)
        same = torch.equal if i == 0 else (lambda x, y: torch.allclose(x, y, rtol=2e-2, atol=1e-4))
        assert loss.requires_grad is False and same(loss, ref[3])
        for j, name in enumerate(("loss_tx", "loss_bd", "loss_sg")):
            assert same(m.logged[f"train:{name}"], ref[j])
    # (two GraphedTrainers on twins differ by ~1e-3 in up to a third of the near-zero-gradient parameters after four
    # Adam steps: compare the distance between the twins with the distance travelled)
    moved = apart = 0.0
    with torch.no_grad():
        for (k, a), (_, b), (_, c) in zip(m.named_parameters(), twin.named_parameters(), init.named_parameters()):
            apart += float((a - b).abs().sum())
            moved += float((a - c).abs().sum())
    assert moved > 0.0 and apart < 0.3 * moved
    assert float(opt.state[next(iter(m.parameters()))]["step"]) == 4.0
    # Lightning's manual optimisation counts global_step through hooks its loop installs on the LightningOptimizer wrapper
    # (lightning/pytorch/loops/optimization/manual.py: _on_before_step / _on_after_step -> optim_step_progress); the
    # replay steps the raw optimizer, so training_step must fire them itself -- else global_step stays 0, ModelCheckpoint
    # never saves and max_steps never ends the fit
    class Progress:
        ready = completed = 0

    class Wrapper:                                          # the slice of LightningOptimizer the path touches
        def __init__(self, optimizer, prog):
            self.optimizer = optimizer
            self._on_before_step = lambda: setattr(prog, "ready", prog.ready + 1)
            self._on_after_step = lambda: setattr(prog, "completed", prog.completed + 1)
    prog = Progress()
    m.trainer = types.SimpleNamespace(optimizers=[Wrapper(opt, prog)], max_epochs=20, datamodule=None)
    for i in range(3):
        m.training_step(bg, i)
    assert (prog.ready, prog.completed) == (3, 3)
    # no hooks on the wrapper: the loop's progress tracker is advanced directly
    class Tracker(Progress):
        def increment_ready(self): self.ready += 1
        def increment_completed(self): self.completed += 1
    trk = Tracker()
    loop = types.SimpleNamespace(epoch_loop=types.SimpleNamespace(manual_optimization=types.SimpleNamespace(optim_step_progress=trk)))
    m.trainer = types.SimpleNamespace(optimizers=[opt], max_epochs=20, datamodule=None, fit_loop=loop)
    m.training_step(bg, 0)
    assert (trk.ready, trk.completed) == (1, 1)
    # switched on too late (the trainer already holds a non-capturable Adam): a clear error, not a capture failure
    late = copy.deepcopy(twin)
    late._graphed_trainer = None
    late.trainer = types.SimpleNamespace(optimizers=[torch.optim.Adam(late.parameters(), lr=1e-3)], max_epochs=20)
    late.enable_graphed_training()
    with pytest.raises(RuntimeError, match="capturable"):
        late.training_step(bg, 0)
    # and off again: the eager step with autograd
    late.enable_graphed_training(False)
    assert late.automatic_optimization and late.training_step(bg, 0).requires_grad


def test_graphed_step_degenerate_batches(cuda):
    """No tx-belongs-bd edge at all, and nothing inside the margin mask: the step must replay, report zero for the
    terms that have no triplets, and leave finite parameters."""
    from segger_amd.hetero import TX_BD
    from segger_amd.synthetic import SyntheticSpec
    from segger_amd.train_step_graph import GraphedTrainer
    spec = SyntheticSpec(n_tx=3000, n_bd=90, k_tx=5, seed=41)
    m, bg = _model(spec, cuda, torch.bfloat16)
    opt = m.configure_optimizers(capturable=True)
    trainer = GraphedTrainer(m, opt, granularity=1.3)
    a = trainer.step(bg).clone()
    assert torch.isfinite(a).all() and float(a[2]) > 0
    bg[TX_BD]["edge_index"] = bg[TX_BD].edge_index[:, :0].contiguous()
    if hasattr(bg, "_segger_amd_cache"):                    # derived structures of the old edge list
        bg._segger_amd_cache.clear()
    b = trainer.step(bg).clone()
    assert torch.isfinite(b).all() and float(b[2]) == 0.0 and float(b[0]) > 0
    bg["tx"]["mask"] = torch.zeros_like(bg["tx"]["mask"])
    bg["bd"]["mask"] = torch.zeros_like(bg["bd"]["mask"])
    if hasattr(bg, "_segger_amd_cache"):
        bg._segger_amd_cache.clear()
    c = trainer.step(bg).clone()
    assert torch.equal(c, torch.zeros_like(c))
    assert all(torch.isfinite(p).all() for p in m.parameters())


def test_graphed_trainer_data_parallel(cuda, tmp_path):
    """Two ranks (gloo, both on this GPU), uneven batch counts: forward + backward graph | one gradient exchange |
    Adam graph.  Replicas must stay bit-identical (same averaged gradients, same Adam state), every rank takes the same
    number of optimizer steps, and the loss goes down."""
    import json
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DP_WORKER_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "tests", "dp_graphed_worker.py")],
                       capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.loads(open(os.path.join(str(tmp_path), f"rank{k}.json")).read()) for k in range(2)]
    assert sorted(x["rank"] for x in recs) == [0, 1]
    assert {x["own"] for x in recs} == {recs[0]["steps_per_epoch"], recs[0]["steps_per_epoch"] - 1}   # one empty step
    for x in recs:
        assert x["finite"] and x["max_param_diff"] == 0.0
        assert x["flat_ptr_stable"] and x["grads_are_views"]        # persistent flat buffer, .grad = its slices
        assert x["adam_steps"] == x["epochs"] * x["steps_per_epoch"]
        assert x["last"] < x["first"]


def test_rccl_one_rank_communicator(cuda, tmp_path):
    """RCCL on this one GPU (task 2a): a one-rank ``nccl`` group runs every collective of bench.py / dp.py
    (same dtypes and reduce ops), an eager step through the forced flat-bucket all-reduce, and 20 captured steps in the
    data-parallel form -- two graphs around the all-reduce -- against the single-graph trainer.  A world-1 mean is the
    identity: the exchange must not change one bit of the flat gradient buffer; the two trainers agree as two runs of
    one trainer do (the loss kernels add with float atomics, so even those differ in the last bits).  The worker is a
    fresh process whose first GPU call is ``init_process_group(device_id=...)``."""
    import json
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(str(tmp_path), "rccl.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", RCCL_WORKER_OUT=out)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_world1_worker.py")],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    x = json.loads(open(out).read())
    assert x["backend"] == "nccl" and x["world"] == 1
    assert x["collectives_identity"]
    assert x["eager_calls"] == 1 and x["eager_bits_changed"] == 0 and x["grads_are_views"]
    assert x["split_calls"] == 20 and x["split_bits_changed"] == 0          # one exchange per step, the identity
    assert x["split_two_graphs"] and x["single_graph"]
    assert x["adam_steps"] == [0.0, 20.0] and x["finite"]
    assert x["max_loss_diff"] < 2e-2 and x["max_param_diff"] < 2e-2 * max(x["param_scale"], 1.0)
    assert x["loss_first_last"][1] < x["loss_first_last"][0]


def test_adam_kernel_matches_torch_adam(cuda):
    """ops.adam_step (segger_adam_step: all tensors in two launches, on torch's own state tensors) against
    torch.optim.Adam(fused, capturable) on the same gradients: parameters and both moments agree to fp32 rounding over
    several steps, the step counters are torch's, odd sizes / an unused parameter / a zero-sized tensor included; an
    optimizer it does not cover is declined."""
    from segger_amd import ops
    g = torch.Generator(device=cuda).manual_seed(5)
    shapes = [(384, 128), (128,), (1, 2, 64), (7,), (3, 5), (0,), (64, 130), (1,)]
    ps_a = [torch.randn(s, device=cuda, generator=g).requires_grad_(True) for s in shapes]
    ps_b = [p.detach().clone().requires_grad_(True) for p in ps_a]
    unused_a, unused_b = (torch.zeros(5, device=cuda, requires_grad=True) for _ in range(2))
    oa = torch.optim.Adam(ps_a + [unused_a], lr=1e-3, fused=True, capturable=True)
    ob = torch.optim.Adam(ps_b + [unused_b], lr=1e-3, fused=True, capturable=True)
    for step in range(6):
        grads = [torch.randn(s, device=cuda, generator=g) * (10.0 ** (step % 3 - 1)) for s in shapes]
        for p, q, gr in zip(ps_a, ps_b, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if step == 0:
            assert not ops.adam_step(oa)              # no state yet: declined, torch creates it
            oa.step()
        else:
            assert ops.adam_step(oa)
        ob.step()
        for p, q in zip(ps_a, ps_b):
            sa, sb = oa.state[p], ob.state[q]
            assert float(sa["step"]) == float(sb["step"]) == step + 1
            for x, y in ((p, q), (sa["exp_avg"], sb["exp_avg"]), (sa["exp_avg_sq"], sb["exp_avg_sq"])):
                # one fp32 ulp of the operands (the lerp cancels: an exp_avg near zero from terms of order 1)
                scale = float(y.abs().max()) if y.numel() else 0.0
                assert torch.allclose(x, y, rtol=2e-6, atol=5e-7 * scale + 1e-12), (step, tuple(p.shape))
        assert unused_a.grad is None and len(oa.state.get(unused_a, {})) == 0
    assert not ops.adam_step(torch.optim.Adam(ps_a, lr=1e-3, weight_decay=0.1, fused=True, capturable=True))
    assert not ops.adam_step(torch.optim.AdamW(ps_a, lr=1e-3, fused=True, capturable=True))
    assert not ops.adam_step(torch.optim.Adam(ps_a, lr=1e-3))          # not capturable: host-side step counters


def test_optim_adam_honours_the_grad_scaler_contract(cuda):
    """torch hands a FUSED optimizer the loss scale and the inf flag instead of unscaling itself
    (``_step_supports_amp_scaling``: ``GradScaler.step`` sets ``optimizer.grad_scale`` / ``found_inf``).  The hand-written
    kernel reads neither, so ``optim.Adam`` must leave such a step to torch: same parameters as
    ``torch.optim.Adam(fused=True)`` under the same scaler, and a step with an inf gradient is skipped."""
    from segger_amd.optim import Adam
    g = torch.Generator(device=cuda).manual_seed(3)
    w0 = torch.randn(48, 32, device=cuda, generator=g)
    x = torch.randn(64, 32, device=cuda, generator=g)
    runs = {}
    for name, cls in (("ours", Adam), ("torch", torch.optim.Adam)):
        w = w0.clone().requires_grad_(True)
        opt = cls([w], lr=1e-2, fused=True, capturable=True)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=1000)
        for step in range(5):
            opt.zero_grad(set_to_none=True)
            loss = (x @ w.t()).square().mean()
            scaler.scale(loss).backward()
            if step == 3:
                w.grad[0, 0] = float("inf")              # the scaler must skip this step and halve the scale
            before = w.detach().clone()
            scaler.step(opt)
            scaler.update()
            if step == 3:
                assert torch.equal(w.detach(), before), name
        runs[name] = (w.detach().clone(), float(opt.state[w]["step"]), scaler.get_scale())
    assert runs["ours"][1] == runs["torch"][1] == 4.0 and runs["ours"][2] == runs["torch"][2] == 512.0
    assert torch.allclose(runs["ours"][0], runs["torch"][0], rtol=1e-6, atol=1e-7)
    assert not torch.equal(runs["ours"][0], w0)


def test_optim_adam_is_a_drop_in_torch_adam(cuda):
    """segger_amd.optim.Adam (what configure_optimizers returns on the GPU): a closure is evaluated once and its loss
    returned (Lightning's automatic optimisation calls step(closure)), the weight-cache hook fires, and the state
    dict moves to and from a plain torch.optim.Adam."""
    from segger_amd.optim import Adam
    g = torch.Generator(device=cuda).manual_seed(9)
    w = torch.randn(64, 32, device=cuda, generator=g).requires_grad_(True)
    x = torch.randn(128, 32, device=cuda, generator=g)
    opt = Adam([w], lr=1e-2, fused=True, capturable=True)
    assert isinstance(opt, torch.optim.Adam)
    calls = []

    def closure():
        calls.append(1)
        opt.zero_grad(set_to_none=True)
        loss = (x @ w.t()).square().mean()
        loss.backward()
        return loss

    losses = [float(opt.step(closure).detach()) for _ in range(5)]  # step 1 through torch (creates the state), then the kernel
    assert len(calls) == 5 and losses[-1] < losses[0]
    assert float(opt.state[w]["step"]) == 5.0
    # state dict -> plain torch Adam on a copy: the next step agrees
    w2 = w.detach().clone().requires_grad_(True)
    ref = torch.optim.Adam([w2], lr=1e-2, fused=True, capturable=True)
    ref.load_state_dict(copy.deepcopy(opt.state_dict()))     # (load_state_dict keeps tensors that already fit: no aliasing wanted here)
    gr = torch.randn(64, 32, device=cuda, generator=g)
    w.grad, w2.grad = gr.clone(), gr.clone()
    opt.step(); ref.step()
    assert torch.allclose(w, w2, rtol=2e-6, atol=2e-6) and float(ref.state[w2]["step"]) == 6.0     # (one ulp of values ~ 3)
    # and back
    opt2 = Adam([w], lr=1e-2, fused=True, capturable=True)
    opt2.load_state_dict(copy.deepcopy(ref.state_dict()))
    assert float(opt2.state[w]["step"]) == 6.0 and torch.equal(opt2.state[w]["exp_avg"], ref.state[w2]["exp_avg"])


def test_merged_draws_step_equals_separate_draws_step(cuda, monkeypatch):
    """A captured step that makes all its draws in its first launch (ops.step_draws, Adam's counters advanced there, the
    dropout counter inside Adam's launch at the END of the step) against one that advances first and draws in four launches:
    same masks, same triplets -- the same loss trajectory up to the order of the loss kernels' float atomics -- and the same
    counters afterwards."""
    from segger_amd import tiles as T
    from segger_amd import train_step_graph as tsg
    from segger_amd.synthetic import SyntheticSpec
    spec = SyntheticSpec(n_tx=20000, n_bd=600, k_tx=6, seed=41)
    runs = {}
    for merged in (True, False):
        monkeypatch.setattr(tsg, "MERGED_DRAWS", merged)
        m, bg = _model(spec, cuda, torch.float32)
        for nt in ("tx", "bd"):
            del bg[nt]["mask"]
        tiling = T.SquareTiling(torch.cat([bg["tx"].pos, bg["bd"].pos]).cpu(), 80.0)
        part = T.partition_by_tiling(bg, tiling, margin=3.0)
        part.build_csr()
        batches = [[t] for t in range(len(part)) if part.node_sizes["bd"][t] > 1][:3]
        opt = m.configure_optimizers(capturable=True)
        trainer = tsg.GraphedTrainer(m, opt, granularity=1.5)
        losses = [trainer.step(part.batch(ids)).clone() for _ in range(3) for ids in batches]
        st = opt.state[next(iter(m.model.conv_layers[1].parameters()))]
        runs[merged] = (torch.stack(losses), int(m.model._step_dev), float(st["step"]),
                        torch.cat([p.detach().reshape(-1) for p in m.parameters()]))
    a, b = runs[True], runs[False]
    assert a[1] == b[1] == 256 * 9 and a[2] == b[2] == 9.0
    assert torch.allclose(a[0], b[0], rtol=2e-3, atol=1e-5), (a[0] - b[0]).abs().max()
    assert (a[3] - b[3]).abs().max().item() <= 2e-3 * b[3].abs().max().item()


def test_captured_step_follows_a_changed_learning_rate(cuda, monkeypatch):
    """An LR scheduler changes ``param_groups[0]["lr"]`` between steps.  On the Adam-kernel route the captured step reads
    (lr, betas, eps) from a device array staged with every batch (segger_adam_step_dev): the SAME graph is replayed -- with
    lr = 0 the parameters stop moving, with the rate restored they move again, and no capture is added.  With torch's own
    optimizer step inside the capture (Python floats baked into its kernel arguments) the bucket is re-captured instead."""
    from segger_amd import tiles as T
    from segger_amd import train_step_graph as G
    from segger_amd.synthetic import SyntheticSpec
    from segger_amd.train_step_graph import GraphedTrainer
    spec = SyntheticSpec(n_tx=12000, n_bd=300, k_tx=6, seed=43)
    flat = lambda m: torch.cat([p.detach().reshape(-1).clone() for p in m.parameters()])
    for kernel in (True, False):
        monkeypatch.setattr(G, "USE_ADAM_KERNEL", kernel)
        m, bg = _model(spec, cuda, torch.float32)
        opt = m.configure_optimizers(capturable=True)
        trainer = GraphedTrainer(m, opt, granularity=1.5)
        trainer.step(bg); trainer.step(bg)
        assert trainer.n_captures == 1 and trainer.buckets[0].graph is not None
        before = flat(m)
        first_graph = trainer.buckets[0].graph
        for g in opt.param_groups:
            g["lr"] = 0.0
        trainer.step(bg)
        assert (trainer.buckets[0].graph is first_graph) == kernel      # same graph replayed | re-captured
        assert torch.equal(before, flat(m))                              # and the new rate was used
        for g in opt.param_groups:
            g["lr"] = 1e-3
        trainer.step(bg)
        assert (trainer.buckets[0].graph is first_graph) == kernel
        assert not torch.equal(before, flat(m))


def test_adam_step_dev_equals_adam_step(cuda):
    """segger_adam_step_dev (hyper-parameters from a device array) == segger_adam_step_ex (by value), bit for bit; and a
    schedule of rates applied through the device array equals torch's capturable Adam stepping through the same schedule."""
    from segger_amd import ops
    g = torch.Generator().manual_seed(3)
    shapes = [(128, 64), (64,), (3, 5, 7), (1,)]
    def make():
        ps = [torch.nn.Parameter(torch.randn(*s_, generator=g).to(cuda)) for s_ in shapes]
        return ps
    torch.manual_seed(0)
    pa = make(); g.manual_seed(3); pb = make(); g.manual_seed(3); pc = make()
    oa = torch.optim.Adam(pa, lr=1e-2, fused=True, capturable=True)
    ob = torch.optim.Adam(pb, lr=1e-2, fused=True, capturable=True)
    oc = torch.optim.Adam(pc, lr=1e-2, fused=True, capturable=True)
    hyper = torch.zeros(4, dtype=torch.float64, device=cuda)
    rates = [1e-2, 5e-3, 0.0, 2e-2, 1e-3]
    for it, lr in enumerate(rates):
        gg = [torch.randn(*s_, generator=g).to(cuda) for s_ in shapes]
        for ps in (pa, pb, pc):
            for p, x in zip(ps, gg):
                p.grad = x.clone()
        for o in (oa, ob, oc):
            o.param_groups[0]["lr"] = lr
        if it == 0:                                    # creates the state (the kernels run on existing state only)
            oa.step(); ob.step(); oc.step()
            continue
        hyper.copy_(torch.tensor([lr, 0.9, 0.999, 1e-8], dtype=torch.float64))
        assert ops.adam_step(oa)
        assert ops.adam_step(ob, hyper_dev=hyper)
        oc.step()
    for a, b, c in zip(pa, pb, pc):
        assert torch.equal(a, b)
        assert torch.allclose(a, c, rtol=2e-6, atol=1e-7)
