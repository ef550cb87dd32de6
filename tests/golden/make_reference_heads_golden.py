#!/usr/bin/env python3
"""Generates tests/golden/reference_heads.npz by RUNNING the reference's own torch-only code.  Run from the repo root
IN THE BUILD CONTAINER (needs /root/reference):

    python tests/golden/make_reference_heads_golden.py

The reference's model files are loaded as files under a private package name.  Their third-party imports that the
exercised functions never touch are satisfied by empty placeholder modules (the technique make_golden.py uses for
``triplet_loss.py``): ``torch_geometric`` (``ist_encoder.py:1``, ``lightning_model.py:2``), ``lightning``
(``:3``; ``LightningModule`` = ``torch.nn.Module``), ``torch_scatter`` (``:4``), ``polars`` (``:7``) and the sibling
packages ``..io.fields`` / ``..data.data_module`` (``:15-16``).  What runs is reference code on torch alone:

* ``sinusoidal_embedding`` (ist_encoder.py:22-31) and ``Positional2dEmbedder.forward`` (``:57-79``), batched (three
  graphs) and unbatched, with given MLP weights                                              -> row a3
* ``LitISTEncoder._scheduled_weights`` (lightning_model.py:136-149) over an epoch sweep        -> row a10
* ``LitISTEncoder.get_losses`` (``:151-213``), called unbound on a stand-in ``self`` whose ``forward`` returns
  GIVEN embeddings: the reference's own TripletLoss / MetricLoss (``triplet_loss.py``), its negative sampling and
  TripletMarginLoss / BCEWithLogitsLoss, its loss combination                                  -> rows a9, a10, N1
  (the encoder itself and ``predict_step`` need PyG / torch_scatter arithmetic and stay out of reach).

Only data is committed: inputs, the torch seed, and the reference's outputs.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/segger"


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    class _Stub:                                         # constructible, never used by what runs here
        def __init__(self, *a, **k):
            pass
    _placeholder("torch_geometric")
    _placeholder("torch_geometric.nn", GATv2Conv=_Stub, Linear=_Stub, HeteroDictLinear=_Stub, HeteroConv=_Stub)
    _placeholder("torch_geometric.data", Batch=_Stub, Data=_Stub)
    _placeholder("lightning", LightningModule=torch.nn.Module)
    _placeholder("torch_scatter", scatter_max=None)
    _placeholder("polars")
    for pkg in ("refsegger", "refsegger.models", "refsegger.io", "refsegger.data"):
        _placeholder(pkg).__path__ = []
    _placeholder("refsegger.io.fields", StandardTranscriptFields=_Stub)
    _placeholder("refsegger.data.data_module", ISTDataModule=_Stub)
    mods = {}
    for name in ("triplet_loss", "ist_encoder", "lightning_model"):
        spec = importlib.util.spec_from_file_location(f"refsegger.models.{name}", f"{REF}/models/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"refsegger.models.{name}"] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods


class _Store(dict):
    """batch['tx'] / batch['tx', 'belongs', 'bd'] as get_losses reads them."""
    def __getattr__(self, k):
        return self[k]


def main():
    ref = load_reference()
    enc, lit = ref["ist_encoder"], ref["lightning_model"]
    out = {}
    g = torch.Generator().manual_seed(11)

    # ---- a3: positional embedding ------------------------------------------------------------------------
    x = torch.cat([torch.rand(37, generator=g), torch.tensor([0.0, 1.0, 0.5])])
    out["sin::x"] = x.numpy()
    out["sin::dim256_p10000"] = enc.sinusoidal_embedding(x, 256, max_period=10000).numpy()
    out["sin::dim7_p1000"] = enc.sinusoidal_embedding(x, 7).numpy()
    emb = enc.Positional2dEmbedder(32)                   # hidden 32 -> dim 16, frequency size 256
    with torch.no_grad():
        for p in emb.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.06 if p.dim() == 2 else 0.1))
    for k, v in emb.state_dict().items():
        out[f"pe::w::{k}"] = v.numpy()
    n = 150
    pos = torch.rand(n, 2, generator=g) * torch.tensor([800.0, 300.0]) + torch.tensor([5000.0, -200.0])
    batch = torch.sort(torch.randint(0, 3, (n,), generator=g)).values
    out["pe::pos"], out["pe::batch"] = pos.numpy(), batch.numpy()
    with torch.no_grad():
        out["pe::out_batched"] = emb(pos, batch).numpy()
        out["pe::out_unbatched"] = emb(pos).numpy()
        one = torch.zeros(n, dtype=torch.long)
        out["pe::out_one_graph"] = emb(pos, one).numpy()

    # ---- a10: loss-weight schedule ------------------------------------------------------------------------
    w_start, w_end = torch.tensor([1.0, 1.0, 0.0]), torch.tensor([1.0, 1.0, 0.5])
    sched = []
    for max_epochs in (1, 2, 20):
        for epoch in (0, 1, 7, 19, 40):
            me = types.SimpleNamespace(trainer=types.SimpleNamespace(max_epochs=max_epochs), current_epoch=epoch,
                                       device=torch.device("cpu"))
            w = lit.LitISTEncoder._scheduled_weights(me, w_start.clone(), w_end.clone())
            wn = lit.LitISTEncoder._scheduled_weights(me, w_start.clone(), w_end.clone(), normalize=False)
            sched.append([max_epochs, epoch] + w.tolist() + wn.tolist())
    out["sched::w_start"], out["sched::w_end"] = w_start.numpy(), w_end.numpy()
    out["sched::table"] = np.asarray(sched, dtype=np.float64)

    # ---- a9 / N1: get_losses on given embeddings ----------------------------------------------------------------
    n_tx, n_bd, c, n_cl = 600, 40, 16, 6
    a = torch.randn(n_cl, 4, generator=g); a = a / a.norm(dim=1, keepdim=True)
    tx_sim = (a @ a.T).float()
    b = torch.randn(n_cl, 3, generator=g); b = b / b.norm(dim=1, keepdim=True)
    bd_sim = (b @ b.T).float()
    z_tx = torch.nn.functional.normalize(torch.randn(n_tx, c, generator=g), dim=-1)
    z_bd = torch.nn.functional.normalize(torch.randn(n_bd, c, generator=g), dim=-1)
    tx_cluster = torch.randint(0, n_cl, (n_tx,), generator=g); tx_cluster[tx_cluster == 2] = 1
    bd_cluster = torch.randint(-1, n_cl, (n_bd,), generator=g)              # -1 = no cluster
    tx_mask = torch.rand(n_tx, generator=g) < 0.85
    bd_mask = torch.rand(n_bd, generator=g) < 0.9
    src = torch.randperm(n_tx, generator=g)[:250]
    ei = torch.stack([src, torch.randint(0, n_bd, (250,), generator=g)])
    for k, v in dict(tx_sim=tx_sim, bd_sim=bd_sim, z_tx=z_tx, z_bd=z_bd, tx_cluster=tx_cluster, bd_cluster=bd_cluster,
                     tx_mask=tx_mask, bd_mask=bd_mask, edge_index=ei).items():
        out[f"loss::{k}"] = v.numpy()
    tl = ref["triplet_loss"]
    rows = []
    for sg_type in ("triplet", "bce"):
        for seed, epoch in ((5, 0), (6, 10), (7, 19)):
            me = types.SimpleNamespace(
                forward=lambda batch: {"tx": z_tx, "bd": z_bd},
                loss_tx=tl.TripletLoss(tx_sim.clone(), margin=0.3), loss_bd=tl.MetricLoss(bd_sim.clone()),
                loss_sg=(torch.nn.TripletMarginLoss(margin=0.4) if sg_type == "triplet" else torch.nn.BCEWithLogitsLoss()),
                _sg_loss_type=sg_type, _w_start=w_start.clone(), _w_end=w_end.clone(),
                trainer=types.SimpleNamespace(max_epochs=20), current_epoch=epoch, device=torch.device("cpu"))
            me._scheduled_weights = lambda ws, we, normalize=True, _me=me: lit.LitISTEncoder._scheduled_weights(_me, ws, we, normalize)
            batch_obj = _Store({"tx": _Store(mask=tx_mask, cluster=tx_cluster), "bd": _Store(mask=bd_mask, cluster=bd_cluster),
                                ("tx", "belongs", "bd"): _Store(edge_index=ei)})
            torch.manual_seed(seed)
            l_tx, l_bd, l_sg, l = lit.LitISTEncoder.get_losses(me, batch_obj)
            rows.append([0.0 if sg_type == "triplet" else 1.0, seed, epoch, float(l_tx), float(l_bd), float(l_sg), float(l)])
    out["loss::results"] = np.asarray(rows, dtype=np.float64)    # [sg_type, seed, epoch, loss_tx, loss_bd, loss_sg, loss]
    np.savez_compressed(os.path.join(HERE, "reference_heads.npz"), **out)
    print("reference_heads.npz:", len(out), "arrays;", "loss rows:", rows[0], rows[-1])


if __name__ == "__main__":
    main()
