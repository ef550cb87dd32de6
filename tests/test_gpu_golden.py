"""The committed golden fixtures replayed ON THE GPU (through the C ABI), so that a drift of the oracle and a drift
of the kernels in the same direction cannot pass unnoticed:

* ``encoder_small.npz``  -- inputs, weights under the reference's state-dict names and STORED float64 answers
  (embeddings, layer-0 attention, predict 4-tuple, segmentation loss, parameter gradients); the oracle is not
  run here at all.  Tolerances as in test_gpu_model.py: fp32 5e-5 on unit-norm embeddings, bf16 3e-2.
* ``triplet_selector.npz`` -- genuine outputs of the reference's own ``models/triplet_loss.py`` (made by
  tests/golden/make_golden.py): the device selector must reproduce positives / negatives bit-exactly from the same
  four uniform draws, ``loss_tx`` (through ``segger_triplet_*``) and ``loss_bd`` within 1e-6
  (reference ``triplet_loss.py:83-125,144-160,176-204``).
* prediction post-processing on the device against the numpy oracle (reference ``data/writer.py:186-241``).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_encoder_golden():
    from segger_amd.hetero import HeteroBatch
    z = np.load(os.path.join(GOLD, "encoder_small.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    b = HeteroBatch(num_graphs=4)
    for k in z.files:
        if k.startswith("in::edge::"):
            b[tuple(k[len("in::edge::"):].split("__"))]["edge_index"] = torch.from_numpy(z[k])
        elif k.startswith("in::tx::") or k.startswith("in::bd::"):
            _, nt, a = k.split("::")
            b[nt][a] = torch.from_numpy(z[k])
    return z, sd, b


def golden_model(sd, dev, dtype):
    from segger_amd import LitISTEncoder
    n_genes, d = sd["model.lin_first.tx.weight"].shape
    hid = sd["model.lin_last.lins.tx.weight"].shape[0]
    m = LitISTEncoder(n_genes=n_genes, in_channels=d, hidden_channels=hid, out_channels=hid, n_mid_layers=2, n_heads=2)
    # reference-shaped checkpoint, strict: the boundary projection is materialised from the incoming weight
    m.load_state_dict(sd, strict=True)
    m.model.compute_dtype = dtype
    return m.to(dev)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_golden_embeddings_attention_and_predict(cuda, dtype):
    from segger_amd import TX_TX
    z, sd, b = load_encoder_golden()
    m = golden_model(sd, cuda, dtype).eval()
    m.model.conv_layers[0].store_attention = True
    bg = b.to(cuda)
    with torch.no_grad():
        out = m(bg)
    tol = 5e-5 if dtype == torch.float32 else 3e-2
    for k in ("tx", "bd"):
        err = np.abs(out[k].double().cpu().numpy() - z[f"out::z_{k}"]).max()
        assert err < tol, f"z_{k}: {err}"
    alpha = m.model.conv_layers[0].attention_weights[TX_TX].double().cpu().numpy()
    assert alpha.shape == z["out::alpha0_tx_tx"].shape
    assert np.abs(alpha - z["out::alpha0_tx_tx"]).max() < (2e-5 if dtype == torch.float32 else 2e-2)
    pred = m.predict_step(bg, 0)
    assert np.array_equal(pred[0].numpy(), z["out::pred_tx_index"])
    assert np.array_equal(pred[3].numpy(), z["out::pred_gene"])
    assert np.abs(pred[2].double().numpy() - z["out::pred_sim"]).max() < tol
    agree = (pred[1].numpy() == z["out::pred_seg"]).mean()
    # an assignment can only differ where the two best candidates are within rounding of each other
    assert agree > (0.995 if dtype == torch.float32 else 0.95), agree
    if dtype == torch.float32:
        diff = pred[1].numpy() != z["out::pred_seg"]
        assert np.abs(pred[2].double().numpy() - z["out::pred_sim"])[diff].max(initial=0.0) < 5e-5


def _layer_of(name: str) -> str:
    """model.conv_layers.2.conv... -> 'conv_layers.2'; other parameters group by their module (lin_first, pos_emb, lin_last)."""
    parts = name.split(".")
    return ".".join(parts[1:3]) if parts[1] == "conv_layers" else parts[1]


def _golden_grads(cuda, dtype):
    z, sd, b = load_encoder_golden()
    m = golden_model(sd, cuda, dtype).eval()              # dropout off, gradients flow
    bg = b.to(cuda)
    loss = m._segmentation_loss(m(bg), bg, torch.from_numpy(z["in::neg"]).to(cuda))
    loss.backward()
    return float(loss), {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("dtype,u,flow_coef", [(torch.bfloat16, 2.0 ** -8, 2.0), (torch.float16, 2.0 ** -11, 8.0)])
def test_16bit_gradients_follow_the_fp32_hip_gradients_per_layer(cuda, dtype, u, flow_coef):
    """Every parameter gradient of the 16-bit step against the SAME kernels at fp32 storage (the exact-fp32 MFMA
    projections, fp32 aggregation), per tensor, in the Euclidean norm:
        || g16 - g32 ||  <=  16 u || g32 ||  +  2 u G_layer ,     G_layer = max over the tensors of the same layer of || g32 ||
    u = the storage format's unit in the last place (bf16 2^-8, f16 2^-11; one rounding costs u / 2).  Between a layer's
    parameters and the loss lie at most 4 layers x (projection, pre-activation, activation) stored forward values and as
    many stored gradient matrices, ~ 30 roundings of u / 2 on the longest path: 16 u.  The additive term is what a tensor
    whose own gradient nearly cancels (lin_r / att of the late layers, the tx-belongs-bd convs of the early ones: 10^2-10^6
    times smaller than their layer's largest tensor) inherits from the rounding of the activations it is summed over:
    proportional to the layer's gradient flow, not to its own size (coefficient 2; 8 for f16, whose stored gradient
    matrices -- entries of 1e-6 .. 1e-8 on this batch -- sit in the format's subnormal range, spacing 2^-24, and lose
    absolute rather than relative precision).  Measured (tools output in DESIGN.md 1): the large
    tensors differ by 1.2-2.5 % (bf16) / 0.2-0.6 % (f16), the loss by < 1e-3.  This replaces the 12 %-of-max bound the
    bf16 run was held to against the stored oracle gradients."""
    loss32, g32 = _golden_grads(cuda, torch.float32)
    loss16, g16 = _golden_grads(cuda, dtype)
    assert abs(loss16 - loss32) < (4e-3 if dtype == torch.bfloat16 else 5e-4)
    assert set(g16) == set(g32) and len(g32) >= 40
    layer_max: dict = {}
    for k, g in g32.items():
        layer_max[_layer_of(k)] = max(layer_max.get(_layer_of(k), 0.0), float(g.norm()))
    worst = 0.0
    for k, g in g32.items():
        err, own, flow = float((g16[k] - g).norm()), float(g.norm()), layer_max[_layer_of(k)]
        assert err <= 16 * u * own + flow_coef * u * flow, f"{k}: |dg| {err:.3e} vs |g| {own:.3e} (layer {flow:.3e})"
        if own > 0.5 * flow:
            worst = max(worst, err / own)
    assert worst < 8 * u                               # the tensors that carry a layer's gradient: well inside the bound


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_golden_loss_and_parameter_gradients(cuda, dtype):
    from segger_amd import TX_BD
    z, sd, b = load_encoder_golden()
    m = golden_model(sd, cuda, dtype).eval()              # dropout off, gradients flow
    bg = b.to(cuda)
    emb = m(bg)
    loss = m._segmentation_loss(emb, bg, torch.from_numpy(z["in::neg"]).to(cuda))
    loss.backward()
    ref_loss = float(z["out::loss_sg"])
    assert abs(loss.item() - ref_loss) < (2e-5 if dtype == torch.float32 else 2e-2)
    named = dict(m.named_parameters())
    checked = 0
    # fp32: every tensor within 2e-3 of its own largest entry of the STORED oracle gradient.  bf16 against the oracle:
    # the per-layer norm bound of test_16bit_gradients_follow_the_fp32_hip_gradients_per_layer (16 u |g| + 2 u G_layer,
    # u = 2^-8), widened by the fp32 run's own distance to the oracle
    if dtype != torch.float32:
        u = 2.0 ** -8
        ref = {k: torch.from_numpy(z[f"grad::{k}"].astype(np.float64)) for k in sd if k in named and named[k].grad is not None}
        layer_max: dict = {}
        for k, g in ref.items():
            layer_max[_layer_of(k)] = max(layer_max.get(_layer_of(k), 0.0), float(g.norm()))
        for k, g in ref.items():
            err = float((named[k].grad.double().cpu() - g).norm())
            assert err <= (16 * u + 1e-3) * float(g.norm()) + 2 * u * layer_max[_layer_of(k)], f"{k}: {err}"
        assert len(ref) >= 40
        return
    rel, floor = 2e-3, 1e-7
    for k in sd:
        ref = z[f"grad::{k}"].astype(np.float64)
        if k not in named or named[k].grad is None:
            assert not np.any(ref), k
            continue
        got = named[k].grad.double().cpu().numpy()
        scale = max(np.abs(ref).max(), 1e-8)
        err = np.abs(got - ref).max()
        assert err < rel * scale + floor, f"{k}: grad err {err} (scale {scale})"
        checked += 1
    assert checked >= 40
    assert bg[TX_BD].edge_index.shape[1] == z["in::neg"].shape[0]


def test_reference_triplet_vectors_on_device(cuda):
    """N1: FastTripletSelector / TripletLoss / MetricLoss on the MI355X against outputs of the reference's own file."""
    from segger_amd.triplet_loss import FastTripletSelector, MetricLoss, TripletLoss
    from segger_amd import ops
    z = np.load(os.path.join(GOLD, "triplet_selector.npz"))
    sim, labels = torch.from_numpy(z["similarity"]), torch.from_numpy(z["labels"])
    emb = torch.from_numpy(z["embeddings"])
    n = labels.numel()
    torch.manual_seed(int(z["seed"]))
    u = tuple(torch.rand(n) for _ in range(4))            # the reference's four CPU draws, in its order
    ud = tuple(t.to(cuda) for t in u)
    sel = FastTripletSelector(sim.to(cuda))
    # (i) everything on the device, its own sort: the reference's default argsort leaves the order of equal labels
    # implementation-defined (triplet_loss.py:41), so WHICH member of the drawn cluster comes back may differ from the
    # CPU-made vectors; the drawn clusters and hence the distances must not
    p_d, n_d, dp, dn = sel.sample_triplets(labels.to(cuda), uniforms=ud)
    assert p_d.is_cuda and n_d.is_cuda
    assert np.array_equal(labels[p_d.cpu()].numpy(), labels.numpy()[z["positives"]])
    assert np.array_equal(labels[n_d.cpu()].numpy(), labels.numpy()[z["negatives"]])
    assert np.array_equal(dp.cpu().numpy(), z["dists_pos"]) and np.array_equal(dn.cpu().numpy(), z["dists_neg"])
    # (ii) with the member order of the CPU sort (the one the vectors were made with) handed to the device: bit-exact
    cpu_index = FastTripletSelector(sim).build_index(labels)
    index = sel.build_index(labels.to(cuda))
    index["members"] = cpu_index["members"].to(cuda)
    pos, neg, dp, dn = sel.sample_triplets(labels.to(cuda), uniforms=ud, index=index)
    assert np.array_equal(pos.cpu().numpy(), z["positives"]) and np.array_equal(neg.cpu().numpy(), z["negatives"])
    assert np.array_equal(dp.cpu().numpy(), z["dists_pos"]) and np.array_equal(dn.cpu().numpy(), z["dists_neg"])

    # loss_tx: the fused HIP triplet kernel over the sampled triplets (fp32 embeddings)
    e = emb.to(cuda)
    idx = torch.arange(n, device=cuda)
    lt = ops.triplet_edge_loss(e, None, idx, pos, neg, float(z["margin"]), eps=1e-6)
    assert abs(lt.item() - float(z["triplet_loss"])) < 1e-6
    # ... and through the module, with torch's device RNG replaced by the reference's draws
    tl = TripletLoss(sim.to(cuda), margin=float(z["margin"]))
    ml = MetricLoss(sim.to(cuda))
    tl.selector.build_index = ml.selector.build_index = lambda labels_, mask_=None: index
    assert abs(float(tl.forward(e, labels.to(cuda), uniforms=ud)) - float(z["triplet_loss"])) < 1e-6
    mask = torch.ones(n, dtype=torch.bool, device=cuda)
    assert abs(float(tl.forward_masked(e, labels.to(cuda), mask, {}, uniforms=ud)) - float(z["triplet_loss"])) < 1e-6
    assert abs(float(ml.forward(e, labels.to(cuda), uniforms=ud)) - float(z["metric_loss"])) < 1e-6

    # gradient of loss_tx through segger_triplet_bwd == autograd through torch's TripletMarginLoss on the same triplets
    e1 = e.clone().requires_grad_(True)
    ops.triplet_edge_loss(e1, None, idx, pos, neg, float(z["margin"]), eps=1e-6).backward()
    e2 = e.clone().requires_grad_(True)
    torch.nn.TripletMarginLoss(margin=float(z["margin"]))(e2, e2[pos], e2[neg]).backward()
    assert torch.allclose(e1.grad, e2.grad, atol=1e-6)


@pytest.mark.parametrize("seed", [0, 3])
def test_postprocess_on_device_matches_numpy_oracle(cuda, seed):
    """N4: the writer's dedup + per-gene Yen/Li thresholds computed on the MI355X (what consumes predict_step)."""
    import postprocess_oracle as po
    from segger_amd import postprocess as pp
    from test_postprocess import fake_predictions
    preds = fake_predictions(seed, n_tx=20000, n_genes=24, n_batches=6)
    got = pp.assign_transcripts_to_cells([[t.to(cuda) for t in p] for p in preds])
    assert got["row_index"].is_cuda and got["similarity_threshold"].is_cuda
    got2 = pp.assign_transcripts_to_cells(preds, device="cuda")            # CPU tuples (predict_step's output), device run
    ref = po.assign_transcripts_to_cells([[t.numpy() for t in p] for p in preds])
    for g in (got, got2):
        assert np.array_equal(g["row_index"].cpu().numpy(), ref["row_index"])
        assert np.array_equal(g["cell_encoding"].cpu().numpy(), ref["cell_encoding"])
        assert np.array_equal(g["similarity"].cpu().numpy(), ref["similarity"])
        assert np.array_equal(g["gene"].cpu().numpy(), ref["gene"])
        assert np.allclose(g["similarity_threshold"].cpu().numpy(), ref["similarity_threshold"], atol=1e-9, equal_nan=True)
        assert abs(g["global_threshold"] - ref["global_threshold"]) < 1e-9
        assert np.array_equal(g["failed_genes"].cpu().numpy(), ref["failed_genes"])


def test_postprocess_consumes_predict_step_output(cuda):
    """predict_step 4-tuples of overlapping prediction tiles -> device post-processing == oracle post-processing."""
    import postprocess_oracle as po
    from segger_amd import postprocess as pp
    z, sd, b = load_encoder_golden()
    m = golden_model(sd, cuda, torch.float32).eval()
    bg = b.to(cuda)
    outs = []
    g = torch.Generator().manual_seed(5)
    for _ in range(3):                                    # three overlapping "tiles": different predict masks
        bg["tx"]["predict_mask"] = (torch.rand(b["tx"].num_nodes, generator=g) < 0.6).to(cuda)
        outs.append(m.predict_step(bg, 0))
    got = pp.assign_transcripts_to_cells(outs, device="cuda")
    ref = po.assign_transcripts_to_cells([[t.numpy() for t in p] for p in outs])
    assert np.array_equal(got["row_index"].cpu().numpy(), ref["row_index"])
    assert np.array_equal(got["cell_encoding"].cpu().numpy(), ref["cell_encoding"])
    assert np.allclose(got["similarity_threshold"].cpu().numpy(), ref["similarity_threshold"], atol=1e-9, equal_nan=True)


# ---------------------------------------------------------------- reference_heads.npz: genuine reference outputs
def _heads():
    return np.load(os.path.join(GOLD, "reference_heads.npz"))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_positional_embedder_matches_reference_outputs(cuda, dtype):
    """Row a3 on the MI355X (segment min/max + sinusoid kernels + MFMA projections) against outputs of the
    reference's own ``Positional2dEmbedder`` (ist_encoder.py:33-79).  fp32: the fp32 rounding of positions ~5e3
    before normalisation bounds the agreement (2e-5, as for the oracle); bf16 activations: 3e-2."""
    from segger_amd.ist_encoder import Positional2dEmbedder
    z = _heads()
    emb = Positional2dEmbedder(32)
    emb.load_state_dict({k[len("pe::w::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("pe::w::")})
    emb = emb.to(cuda)
    pos, batch = torch.from_numpy(z["pe::pos"]).to(cuda), torch.from_numpy(z["pe::batch"]).to(cuda)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    with torch.no_grad():
        for key, b in (("pe::out_batched", batch), ("pe::out_one_graph", torch.zeros_like(batch)), ("pe::out_unbatched", None)):
            got = emb(pos, b, dtype=dtype).float().cpu().numpy()
            assert got.shape == z[key].shape and np.abs(got - z[key]).max() < tol, key


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_get_losses_matches_reference_outputs(cuda, dtype):
    """Rows a9 / a10 / N1 on the MI355X: ``LitISTEncoder.get_losses`` (fused triplet kernels, device selector, masked
    losses, schedule) on GIVEN embeddings against the values the reference's own ``get_losses``
    (lightning_model.py:151-213, with its TripletLoss / MetricLoss) returned for the same embeddings, masks, clusters,
    edges and random draws.  fp32 embeddings: 1e-5; bf16 embeddings (rounded inputs): 2e-2."""
    from test_oracle import replay_reference_draws
    from segger_amd import LitISTEncoder, TX_BD
    from segger_amd.hetero import HeteroBatch
    z = _heads()
    n_tx, n_bd = z["loss::z_tx"].shape[0], z["loss::z_bd"].shape[0]
    b = HeteroBatch(num_graphs=1)
    b["tx"]["mask"], b["tx"]["cluster"] = torch.from_numpy(z["loss::tx_mask"]), torch.from_numpy(z["loss::tx_cluster"])
    b["bd"]["mask"], b["bd"]["cluster"] = torch.from_numpy(z["loss::bd_mask"]), torch.from_numpy(z["loss::bd_cluster"])
    b["tx"]["x"], b["bd"]["x"] = torch.zeros(n_tx, dtype=torch.int32), torch.zeros(n_bd, 4)
    b[TX_BD]["edge_index"] = torch.from_numpy(z["loss::edge_index"])
    bg = b.to(cuda)
    emb = {"tx": torch.from_numpy(z["loss::z_tx"]).to(cuda).to(dtype), "bd": torch.from_numpy(z["loss::z_bd"]).to(cuda).to(dtype)}
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for row in z["loss::results"]:
        m = LitISTEncoder(n_genes=4, in_channels=16, sg_loss_type="triplet" if row[0] == 0 else "bce",
                          tx_margin=0.3, sg_margin=0.4).to(cuda)
        m.set_similarities(torch.from_numpy(z["loss::tx_sim"]).to(cuda), torch.from_numpy(z["loss::bd_sim"]).to(cuda))
        m._max_epochs_override, m.current_epoch = 20, int(row[2])
        tx_mask, bd_mask, u_tx, u_bd, dst_neg = replay_reference_draws(z, row)

        def per_node(us, mask, n):          # the reference draws one number per MASKED node; the product per node
            out = []
            for u in us:
                full = torch.zeros(n)
                full[mask] = u
                out.append(full.to(cuda))
            return tuple(out)
        # WHICH member of a drawn cluster comes back depends on the order ``torch.argsort`` leaves equal labels in
        # (triplet_loss.py:41) -- implementation-defined, and different between torch's CPU sort (which made the
        # vectors) and its GPU sort: hand the device selectors the CPU member order, as test_reference_triplet_vectors does
        from segger_amd.graph import batch_cache

        def pin_members(loss, labels, mask):
            ix = loss.selector.build_index(labels.to(cuda), mask.to(cuda))
            idx = mask.nonzero().squeeze(1)
            ix["members"][: idx.numel()] = idx[torch.argsort(labels[mask])].to(cuda)
            return ix
        cache = batch_cache(bg)
        cache.clear()
        cache["persistent"] = {"tx_triplet_index": pin_members(m.loss_tx, b["tx"]["cluster"], tx_mask),
                               "bd_metric_index": pin_members(m.loss_bd, b["bd"]["cluster"], bd_mask)}
        losses = m.get_losses(bg, dst_neg=dst_neg.to(cuda), embeddings=emb,
                              uniforms=(per_node(u_tx, tx_mask, n_tx), per_node(u_bd, bd_mask, n_bd)))
        got = [float(v) for v in losses]
        assert np.allclose(got, row[3:7], atol=tol), (row.tolist(), got)
