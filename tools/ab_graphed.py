"""A/B of the captured small-batch training step (segger's default 1M-edge batches of a 10M-tx FOV) under integer switches
of segger_amd.ops:   VARIANTS="base:;packed:_CONTRIB_MIN_EDGES=4096" python tools/ab_graphed.py
Per variant: fresh model + trainer, one warm epoch (captures), then the timed epoch; prints ms/step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import LitISTEncoder, ops
from segger_amd.fov import build_fov_batches
from segger_amd.synthetic import SyntheticSpec
from segger_amd.train_step_graph import GraphedTrainer
from segger_amd import train_step_graph as tsg

dev = torch.device("cuda:0")
spec = SyntheticSpec(n_tx=int(os.environ.get("N_TX", 10_000_000)), n_bd=int(os.environ.get("N_BD", 100_000)), k_tx=15, seed=0)
part, batches, aux, _ = build_fov_batches(spec, dev)
batches = [b for b in batches if len(b) == 1][: int(os.environ.get("BATCHES", 150))]
defaults = {}
variants = []
for item in os.environ.get("VARIANTS", "base:").split(";"):
    name, _, fl = item.partition(":")
    variants.append((name, dict(kv.split("=") for kv in fl.split(",") if kv)))
for rnd in range(int(os.environ.get("ROUNDS", 2))):
    for name, fl in variants:
        for (mod, k), v in defaults.items():
            setattr(mod, k, v)
        for k, v in fl.items():
            mod = ops if hasattr(ops, k) else tsg          # (switches of segger_amd.ops or of segger_amd.train_step_graph)
            defaults.setdefault((mod, k), getattr(mod, k))
            setattr(mod, k, type(getattr(mod, k))(int(v)))
        torch.manual_seed(0)
        m = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
        m.model._materialize_bd(spec.bd_dim, "cpu")
        m = m.to(dev)
        m.set_similarities(aux["tx_similarity"].to(dev), aux["bd_similarity"].to(dev))
        m._max_epochs_override, m.current_epoch = 20, 10
        m.model.compute_dtype = torch.bfloat16
        m.train()
        tr = GraphedTrainer(m, m.configure_optimizers(capturable=True))
        for ids in batches:
            tr.step(part.batch(ids))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ids in batches:
            out = tr.step(part.batch(ids))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(batches)
        print(f"round {rnd} {name:10s} {fl}: {dt * 1e3:.3f} ms/step  loss {float(out[3]):.4f}  buckets {len(tr.buckets)}", flush=True)
        del tr, m
