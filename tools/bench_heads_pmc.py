"""A few launches of the round-3 front-end / loss kernels at C2 size for the PMC passes of tools/pmc.sh: the positional
embedder forward and its one-kernel backward, the grouped segmentation-triplet backward and the loss_tx backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
from segger_amd.graph import csr_from_coo
from segger_amd.ist_encoder import Positional2dEmbedder
dev = torch.device('cuda')
n, nbd, etb = 1_000_000, 10_000, 394_000
g = torch.Generator(device=dev).manual_seed(0)
pos = torch.rand(n, 2, device=dev, generator=g) * 1000
batch = torch.zeros(n, dtype=torch.int64, device=dev)
emb = Positional2dEmbedder(128).to(dev)
gy = torch.randn(n, 128, device=dev, generator=g).bfloat16()
z = torch.nn.functional.normalize(torch.randn(n, 64, device=dev, generator=g), dim=-1).bfloat16().requires_grad_(True)
zb = torch.nn.functional.normalize(torch.randn(nbd, 64, device=dev, generator=g), dim=-1).bfloat16().requires_grad_(True)
src = torch.randperm(n, device=dev, generator=g)[:etb]
dstp = torch.randint(0, nbd, (etb,), device=dev, generator=g)
dneg = (dstp + torch.randint(1, nbd, (etb,), device=dev, generator=g)) % nbd
groups = csr_from_coo(dstp, src, nbd, n, validate=False)
anchors = torch.arange(n, device=dev)
p_ = torch.randint(0, n, (n,), device=dev, generator=g)
q_ = torch.randint(0, n, (n,), device=dev, generator=g)
for _ in range(3):
    pe = emb(pos, batch, num_graphs=1, dtype=torch.bfloat16)
    pe.backward(gy)
    z.grad = None; zb.grad = None
    ops.triplet_edge_loss(z, zb, src, dstp, dneg, 0.4, pos_groups=groups, anchors_unique=True).backward()
    ops.triplet_edge_loss(z, None, anchors, p_, q_, 0.3).backward()
torch.cuda.synchronize()
