"""A/B of the C2 training step (bench.py's timed region) under switches of segger_amd.ops:
    FUSED_DX=0|1 (one-pass projection backward), any other integer attribute of segger_amd.ops by name;   STEPS, WARMUP, N_TX
Alternates the variants in one process (same box, same clocks): VARIANTS="base:FUSED_DX=0;new:FUSED_DX=1"."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import LitISTEncoder, ops
from segger_amd.synthetic import SyntheticSpec, make_graph

dev = torch.device("cuda")
n_tx = int(os.environ.get("N_TX", 1_000_000))
spec = SyntheticSpec(n_tx=n_tx, n_bd=max(n_tx // 100, 10), k_tx=15, seed=0)
b, aux = make_graph(spec, return_aux=True)
batch = b.to(dev)
torch.manual_seed(0)
model = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
model.model._materialize_bd(spec.bd_dim, "cpu")
model.model.compute_dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[os.environ.get('DTYPE', 'bf16')]
model = model.to(dev)
model.set_similarities(aux["tx_similarity"].to(dev), aux["bd_similarity"].to(dev))
model._max_epochs_override, model.current_epoch = 20, 10
model.train()
opt = model.configure_optimizers()


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss


def apply(flags):
    ops.FUSED_WGRAD_DX = bool(int(flags.get("FUSED_DX", 1)))
    if "ROW_ORDER" in flags:                  # window of the by-destination visiting order (graph.ROW_ORDER_WINDOW_DST); 0 = off
        from segger_amd import graph
        graph.ROW_ORDER_WINDOW_DST = int(flags["ROW_ORDER"])
        graph.ROW_ORDER_FORWARD = bool(int(flags.get("ORDER_FWD", 0)))
        graph.batch_cache(batch).clear()      # the sorted views are cached on the batch: rebuild them under the new setting
    for k, v in flags.items():
        if k not in ("FUSED_DX", "ROW_ORDER", "ORDER_FWD") and hasattr(ops, k):
            setattr(ops, k, type(getattr(ops, k))(int(v)))


variants = []
for item in os.environ.get("VARIANTS", "base:FUSED_DX=0;fused:FUSED_DX=1").split(";"):
    name, _, fl = item.partition(":")
    variants.append((name, dict(kv.split("=") for kv in fl.split(",") if kv)))
steps, warm = int(os.environ.get("STEPS", 15)), int(os.environ.get("WARMUP", 3))
for rnd in range(int(os.environ.get("ROUNDS", 2))):
    for name, fl in variants:
        apply(fl)
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        print(f"round {rnd} {name:8s} {fl}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step  loss {float(loss):.4f}", flush=True)
