"""Times the positional embedder at C2 size (1M nodes, bf16): fused kernel (inference / training variant) against the
unfused route (posfreq + linear + SiLU + linear)."""
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
pos = torch.rand(n, 2, device=dev) * 500; batch = torch.zeros(n, dtype=torch.long, device=dev)
emb = Positional2dEmbedder(128).to(dev)
dt = torch.bfloat16
mins, maxs = ops.segment_minmax(pos, batch, 1)
l0, l2 = emb.mlp[0], emb.mlp[2]
def unfused():
    freq = ops.posfreq(pos, batch, mins, maxs, 256, dt)
    h = torch.nn.functional.silu(ops.linear(freq, l0.weight, l0.bias))
    return ops.linear(h, l2.weight, l2.bias).flatten(-2)
with torch.no_grad():
    print("inference: fused %.3f ms   unfused %.3f ms" % (t(lambda: ops.posmlp(pos, batch, mins, maxs, l0.weight, l0.bias, l2.weight, l2.bias, dt)), t(unfused)))
print("training forward: fused %.3f ms   unfused %.3f ms" % (t(lambda: ops.posmlp(pos, batch, mins, maxs, l0.weight, l0.bias, l2.weight, l2.bias, dt)), t(unfused)))
gy = torch.randn(n, 128, device=dev).to(dt)
def fb(f):
    emb.zero_grad(set_to_none=True); f().backward(gy)
print("training fwd+bwd: fused %.3f ms   unfused %.3f ms" % (t(lambda: fb(lambda: ops.posmlp(pos, batch, mins, maxs, l0.weight, l0.bias, l2.weight, l2.bias, dt))), t(lambda: fb(unfused))))
