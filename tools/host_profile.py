"""Host-side profile of the small-batch training regime (segger's default edges_per_batch = 1M): where the Python /
launch time of a step goes (cProfile), and how many kernels a step launches."""
import cProfile, os, pstats, sys, time, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import LitISTEncoder, TX_BD
from segger_amd.dp import FlatGradBucket
from segger_amd.fov import build_fov_batches
from segger_amd.synthetic import SyntheticSpec
dev = torch.device('cuda')
spec = SyntheticSpec(n_tx=int(os.environ.get('N_TX', 3_000_000)), n_bd=int(os.environ.get('N_BD', 30_000)), k_tx=15, seed=0)
part, batches, aux, tiling = build_fov_batches(spec, dev, edges_per_batch=int(os.environ.get('EPB', 1_000_000)))
torch.manual_seed(0)
model = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
model.model._materialize_bd(spec.bd_dim, "cpu")
model.model.compute_dtype = torch.bfloat16
model = model.to(dev)
model.set_similarities(aux["tx_similarity"].to(dev), aux["bd_similarity"].to(dev))
model._max_epochs_override, model.current_epoch = 20, 10
model.train()
opt = model.configure_optimizers()
def step(k, i):
    opt.zero_grad(set_to_none=True)
    model.training_step(part.batch(batches[k]), i).backward()
    opt.step()
n = min(len(batches), 60)
for k in range(3): step(k, 0)
for epoch in range(2):                                   # the second epoch finds the per-tile sampler indices cached
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(n): step(k, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"{len(batches)} batches; epoch {epoch}: {dt * 1e3:.2f} ms / step (wall)")
if os.environ.get('NO_CPROFILE'):
    sys.exit(0)
# device-only time of the same steps
a, e = torch.cuda.Event(True), torch.cuda.Event(True)
pr = cProfile.Profile()
pr.enable()
for k in range(n): step(k, k)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
