"""Times the positional embedder's backward at C2 size (1M transcripts = 2M coordinate rows, bf16): the one-pass kernel
(segger_posmlp_bwd) against the three kernels it replaces (weight gradient 64x64, data gradient with the SiLU' epilogue,
generated-operand weight gradient 64x256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
from segger_amd.ist_encoder import Positional2dEmbedder
dev = torch.device('cuda')
n = int(os.environ.get('N', 1_000_000))
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); e = torch.cuda.Event(True); a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it
g = torch.Generator(device=dev).manual_seed(0)
pos = torch.rand(n, 2, device=dev, generator=g) * 1000
batch = torch.zeros(n, dtype=torch.int64, device=dev)
emb = Positional2dEmbedder(128).to(dev)
gy = torch.randn(n, 128, device=dev, generator=g).bfloat16()
for fused in (False, True):
    ops.FUSED_POSMLP_BWD = fused
    pe = emb(pos, batch, num_graphs=1, dtype=torch.bfloat16)
    ms_f = t(lambda: emb(pos, batch, num_graphs=1, dtype=torch.bfloat16))
    ms_b = t(lambda: pe.backward(gy, retain_graph=True))
    print(f"fused_backward={fused}: forward {ms_f:.3f} ms  backward {ms_b:.3f} ms "
          f"(algorithmic bytes of the one-pass form {n * 2 * (64 * 2 * 2 + 4) / 1e6:.0f} MB)", flush=True)
