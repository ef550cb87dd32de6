"""Host-side time per phase of the graphed training step (batch assembly, bucket pick, staging, replay), no device sync
inside the loop, followed by a cProfile of the same loop:  python tools/host_graphed.py"""
import sys, time
sys.path.insert(0, '.')
import torch
from segger_amd import LitISTEncoder
from segger_amd.fov import build_fov_batches
from segger_amd.synthetic import SyntheticSpec
from segger_amd.train_step_graph import GraphedTrainer

dev = torch.device("cuda:0")
spec = SyntheticSpec(n_tx=10_000_000, n_bd=100_000, k_tx=15, seed=0)
part, batches, aux, _ = build_fov_batches(spec, dev)
torch.manual_seed(0)
m = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
m.model._materialize_bd(spec.bd_dim, "cpu")
m = m.to(dev)
m.set_similarities(aux["tx_similarity"].to(dev), aux["bd_similarity"].to(dev))
m._max_epochs_override, m.current_epoch = 20, 10
m.model.compute_dtype = torch.bfloat16
m.train()
opt = m.configure_optimizers(capturable=True)
tr = GraphedTrainer(m, opt)
for ids in batches:                      # epoch 0: captures + caches
    tr.step(part.batch(ids))
torch.cuda.synchronize()
T = {"batch": 0.0, "pick": 0.0, "stage": 0.0, "replay": 0.0, "inval": 0.0}
from segger_amd import ops
t_all = time.perf_counter()
for ids in batches:
    t0 = time.perf_counter(); b = part.batch(ids)
    t1 = time.perf_counter()
    fit = [x for x in tr.buckets if x.fits(b)]; best = min(fit, key=lambda x: x.waste(b))
    t2 = time.perf_counter(); best.stage(b)
    t3 = time.perf_counter(); best.graph.replay()
    t4 = time.perf_counter(); ops.invalidate_weight_cache()
    t5 = time.perf_counter()
    for k, d in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        T[k] += d
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
n = len(batches)
print({k: round(v / n * 1e3, 3) for k, v in T.items()}, "host ms/step; wall", round(wall / n * 1e3, 3), "ms/step", n, "steps")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for ids in batches:
    b = part.batch(ids)
    fit = [x for x in tr.buckets if x.fits(b)]; best = min(fit, key=lambda x: x.waste(b))
    best.stage(b); best.graph.replay()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
