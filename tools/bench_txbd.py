import torch, sys, time
sys.path.insert(0, '.')
from segger_amd import ops, TX_BD
from segger_amd.graph import build_edge_graph
from segger_amd.synthetic import SyntheticSpec, make_graph
dev = torch.device('cuda')
b = make_graph(SyntheticSpec(n_tx=1_000_000, n_bd=10_000, k_tx=15, seed=0))
ei = b[TX_BD].edge_index.to(dev)
n, nb = 1_000_000, 10_000
H, C = 2, 64; hc = 128
gen = torch.Generator(device=dev).manual_seed(0)
xp = torch.randn(n, 3 * hc, device=dev, generator=gen).bfloat16()
xb = torch.randn(nb, hc, device=dev, generator=gen).bfloat16()
att = torch.randn(hc, device=dev, generator=gen) * 0.3
bias = torch.zeros(hc, device=dev)
out = torch.empty(nb, hc, dtype=torch.bfloat16, device=dev); pre = torch.empty_like(out); lse = torch.empty(nb, H, device=dev)
gy = torch.randn(nb, hc, device=dev, generator=gen).bfloat16(); gxp = torch.empty_like(xp); gxb = torch.empty_like(xb)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); e = torch.cuda.Event(True); a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it
for name, g in (("lazy", build_edge_graph(ei, n, nb, need_by_src="lazy")), ("full", build_edge_graph(ei, n, nb))):
    fwd = lambda: ops.gatv2_fwd_launch(g.by_dst, xp[:, 2*hc:], xb, att, bias, H, C, out, pre=pre, lse=lse, apply_gelu=True, dropout_p=0.2, seed=5)
    print(name, "fwd %.3f ms" % t(fwd))
    bwd = lambda: ops.gatv2_bwd_launch(g, xp[:, 2*hc:], xb, att, bias, H, C, gy, pre, lse, gxp[:, 2*hc:], gxb, apply_gelu=True, dropout_p=0.2, seed=5)
    print(name, "unique", g.src_unique(), "bwd %.3f ms" % t(bwd))
z = torch.empty_like(gxp)
print("memset2d-like torch zero of slice %.3f ms" % t(lambda: gxp[:, 2*hc:].zero_()))
