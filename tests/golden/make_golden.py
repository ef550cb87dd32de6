#!/usr/bin/env python3
"""Generates the committed golden fixtures.  Run from the repo root IN THE BUILD
CONTAINER (it needs /root/reference for part 2):

    python tests/golden/make_golden.py

1. ``encoder_small.npz`` -- inputs (a 4-tile HeteroData-contract batch), weights under the
   reference's state-dict key names, and float64 ORACLE outputs (embeddings, attention of
   layer 0, predict_step 4-tuple, segmentation loss with given negatives, parameter-gradient
   checksums).  The reference itself cannot produce these: its arithmetic needs
   torch_geometric / torch_scatter, which are not installable here -- these
   vectors pin the oracle against drift and the HIP path against the oracle ("parity unpinned").

2. ``triplet_selector.npz`` -- produced by IMPORTING THE REFERENCE file
   ``/root/reference/src/segger/models/triplet_loss.py`` (its only non-torch import,
   ``torch_geometric.data.Data`` at ``:3``, is unused and satisfied by an empty placeholder
   module): labels, similarity matrix, torch seed -> positives, negatives, dists_pos, dists_neg,
   TripletLoss and MetricLoss values.  These are genuine reference outputs.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import segger_oracle as O  # noqa: E402

H, D, HID, OUT, N_GENES, BD_DIM, N_LAYERS = 2, 32, 32, 32, 40, 12, 4


def make_state_dict(seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).float()
    sd = {
        "model.lin_first.tx.weight": r(N_GENES, D),
        "model.lin_first.bd.weight": r(D, BD_DIM, scale=BD_DIM ** -0.5),
        "model.lin_first.bd.bias": r(D, scale=0.1),
        "model.pos_emb.mlp.0.weight": r(D // 2, 256, scale=1 / 16),
        "model.pos_emb.mlp.0.bias": r(D // 2, scale=0.1),
        "model.pos_emb.mlp.2.weight": r(D // 2, D // 2, scale=0.25),
        "model.pos_emb.mlp.2.bias": r(D // 2, scale=0.1),
    }
    fin = 2 * D
    for li in range(N_LAYERS):
        c = OUT if li == N_LAYERS - 1 else HID
        for et in (O.TX_TX, O.TX_BD):
            p = f"model.conv_layers.{li}.conv.convs.{O.pyg_key(et)}."
            sd[p + "lin_l.weight"] = r(H * c, fin, scale=fin ** -0.5)
            sd[p + "lin_l.bias"] = r(H * c, scale=0.1)
            sd[p + "lin_r.weight"] = r(H * c, fin, scale=fin ** -0.5)
            sd[p + "lin_r.bias"] = r(H * c, scale=0.1)
            sd[p + "att"] = r(1, H, c, scale=0.5)
            sd[p + "bias"] = r(H * c, scale=0.1)
        fin = H * c
    for k in ("tx", "bd"):
        sd[f"model.lin_last.lins.{k}.weight"] = r(OUT, fin, scale=fin ** -0.5)
        sd[f"model.lin_last.lins.{k}.bias"] = r(OUT, scale=0.1)
    return sd


def encoder_fixture():
    from segger_amd.synthetic import SyntheticSpec, make_graph
    spec = SyntheticSpec(n_tx=400, n_bd=36, k_tx=4, n_genes=N_GENES, bd_dim=BD_DIM, n_graphs=4, seed=42)
    b = make_graph(spec)
    g = torch.Generator().manual_seed(1)
    b["tx"]["predict_mask"] = torch.rand(spec.n_tx, generator=g) < 0.8
    b["tx"]["index"] = torch.randperm(5000, generator=g)[: spec.n_tx]
    b["bd"]["index"] = (torch.randperm(spec.n_bd, generator=g) + 100).to(torch.int32)
    sd32 = make_state_dict()
    sd = {k: v.double().requires_grad_(True) for k, v in sd32.items()}
    z, attn = O.ist_encoder_forward(sd, b.x_dict, b.edge_index_dict, b.pos_dict, b.batch_dict, n_heads=H,
                                    return_attention=True)
    ei_tb = b[O.TX_BD].edge_index
    neg = (ei_tb[1] + torch.randint(1, spec.n_bd, (ei_tb.shape[1],), generator=g)) % spec.n_bd
    loss = O.segmentation_loss(z["tx"], z["bd"], ei_tb, neg, "triplet", 0.4)
    loss.backward()
    pred = O.predict_step({k: v.detach() for k, v in sd.items()}, b, n_heads=H)
    out = {f"w::{k}": v.numpy() for k, v in sd32.items()}
    for nt in ("tx", "bd"):
        for a, v in b[nt].items():
            out[f"in::{nt}::{a}"] = v.numpy()
    for et in b.edge_types:
        out["in::edge::" + "__".join(et)] = b[et].edge_index.numpy()
    out["in::neg"] = neg.numpy()
    out["out::z_tx"] = z["tx"].detach().numpy()
    out["out::z_bd"] = z["bd"].detach().numpy()
    out["out::alpha0_tx_tx"] = attn[(0, O.TX_TX)].detach().numpy()
    out["out::loss_sg"] = np.float64(loss.item())
    out["out::pred_tx_index"], out["out::pred_seg"], out["out::pred_sim"], out["out::pred_gene"] = (t.numpy() for t in pred)
    for k, v in sd.items():
        out[f"grad::{k}"] = v.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "encoder_small.npz"), **out)
    print("encoder_small.npz:", len(out), "arrays; loss_sg", loss.item())


def triplet_fixture():
    tg, tgd = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.data")
    tgd.Data = object
    sys.modules["torch_geometric"], sys.modules["torch_geometric.data"] = tg, tgd
    spec = importlib.util.spec_from_file_location(
        "ref_triplet_loss", "/root/reference/src/segger/models/triplet_loss.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    g = torch.Generator().manual_seed(3)
    n_clusters, n, c = 7, 500, 16
    a = torch.randn(n_clusters, 5, generator=g)
    a = a / a.norm(dim=1, keepdim=True)
    sim = (a @ a.T).float()
    labels = torch.randint(0, n_clusters, (n,), generator=g)
    labels[labels == 4] = 5                      # one cluster absent
    emb = torch.nn.functional.normalize(torch.randn(n, c, generator=g), dim=-1)
    seed = 1234
    torch.manual_seed(seed)
    pos, neg, dpos, dneg = ref.FastTripletSelector(sim.clone()).sample_triplets(labels)
    torch.manual_seed(seed)
    lt = ref.TripletLoss(sim.clone(), margin=0.3).forward(emb, labels)
    torch.manual_seed(seed)
    lm = ref.MetricLoss(sim.clone()).forward(emb, labels)
    np.savez_compressed(
        os.path.join(HERE, "triplet_selector.npz"),
        similarity=sim.numpy(), labels=labels.numpy(), embeddings=emb.numpy(), seed=np.int64(seed),
        positives=pos.numpy(), negatives=neg.numpy(), dists_pos=dpos.numpy(), dists_neg=dneg.numpy(),
        triplet_loss=np.float64(float(lt)), metric_loss=np.float64(float(lm)), margin=np.float64(0.3))
    print("triplet_selector.npz: triplet", float(lt), "metric", float(lm))


if __name__ == "__main__":
    encoder_fixture()
    if os.path.exists("/root/reference/src/segger/models/triplet_loss.py"):
        triplet_fixture()
    else:
        print("reference not present: triplet_selector.npz not regenerated")
