"""Training step at segger's default batch size (~1M edges): eager vs hipGraph-captured encoder."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import LitISTEncoder
from segger_amd.inference import bucket_sizes
from segger_amd.train_graph import GraphedEncoder
from segger_amd.synthetic import SyntheticSpec, make_graph
dev = torch.device('cuda')
spec = SyntheticSpec(n_tx=66_000, n_bd=660, k_tx=15, seed=0)
b, aux = make_graph(spec, return_aux=True)
bg = b.to(dev)
torch.manual_seed(0)
m = LitISTEncoder(n_genes=spec.n_genes, in_channels=128); m.model._materialize_bd(spec.bd_dim, 'cpu')
m.model.compute_dtype = torch.bfloat16
m = m.to(dev); m.set_similarities(aux['tx_similarity'].to(dev), aux['bd_similarity'].to(dev))
m._max_epochs_override, m.current_epoch = 20, 10
m.train()
opt = m.configure_optimizers()
def eager():
    opt.zero_grad(set_to_none=True); loss = m.training_step(bg, 0); loss.backward(); opt.step(); return loss
ge = GraphedEncoder(m, bucket_sizes(bg), bd_dim=spec.bd_dim, max_graphs=1)
def graphed():
    opt.zero_grad(set_to_none=True)
    z = ge(bg)
    loss = m.get_losses(bg, embeddings=z)[3]
    loss.backward(); opt.step(); return loss
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): l = float(fn().detach()) if _ == n - 1 else fn() and 0
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, l
print('eager   %.2f ms/step loss %.4f' % t(eager))
import gc; gc.collect(); torch.cuda.synchronize()
print('graphed %.2f ms/step loss %.4f' % t(graphed))
import time
def tm(name, fn, n=10):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(name, '%.2f ms' % ((time.perf_counter()-t0)/n*1e3), flush=True)
from segger_amd.inference import pad_batch
tm('pad_batch', lambda: pad_batch(bg, ge.sizes))
pb = pad_batch(bg, ge.sizes)
tm('stage', lambda: ge._stage(pb))
tm('replay fwd only', lambda: ge.callable(*ge.inp))
def fb():
    z = ge.callable(*ge.inp); (z[0].float().sum()).backward()
tm('replay fwd+bwd', fb)
def lossonly():
    with torch.no_grad(): z = {'tx': torch.nn.functional.normalize(torch.randn(bg['tx'].num_nodes, 64, device=dev)).bfloat16(), 'bd': torch.nn.functional.normalize(torch.randn(bg['bd'].num_nodes, 64, device=dev)).bfloat16()}
    z = {k: v.requires_grad_(True) for k, v in z.items()}
    m.get_losses(bg, embeddings=z)[3].backward()
tm('losses fwd+bwd', lossonly)
