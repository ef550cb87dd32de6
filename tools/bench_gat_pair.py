"""tx-neighbors-tx + tx-belongs-bd forward of one layer at C2 size: two launches against segger_gatv2_fwd_pair (the
tx-belongs-bd blocks ride in the tx-neighbors-tx launch); also checks that the outputs are identical."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops, TX_BD, TX_TX
from segger_amd.graph import build_edge_graph
from segger_amd.synthetic import SyntheticSpec, make_graph
dev = torch.device('cuda')
n, nb = int(os.environ.get('N_TX', 1_000_000)), int(os.environ.get('N_BD', 10_000))
b = make_graph(SyntheticSpec(n_tx=n, n_bd=nb, k_tx=15, seed=0))
g_tt = build_edge_graph(b[TX_TX].edge_index.to(dev), n, n)
g_tb = build_edge_graph(b[TX_BD].edge_index.to(dev), n, nb, need_by_src="lazy")
H, C = 2, 64; hc = 128
gen = torch.Generator(device=dev).manual_seed(0)
xp = torch.randn(n, 3 * hc, device=dev, generator=gen).bfloat16()
xb = torch.randn(nb, hc, device=dev, generator=gen).bfloat16()
att = torch.randn(2, hc, device=dev, generator=gen) * 0.3
bias = torch.randn(2, hc, device=dev, generator=gen) * 0.1
step = torch.zeros(1, dtype=torch.int64, device=dev)
bits = ops.dropout_bits_many([(g_tt.by_dst, [0]), (g_tb.by_dst, [1])], H, 0.2, step)
def outs():
    return dict(y_tx=torch.empty(n, hc, dtype=torch.bfloat16, device=dev), p_tx=torch.empty(n, hc, dtype=torch.bfloat16, device=dev),
                l_tx=torch.empty(n, H, device=dev), y_bd=torch.empty(nb, hc, dtype=torch.bfloat16, device=dev),
                p_bd=torch.empty(nb, hc, dtype=torch.bfloat16, device=dev), l_bd=torch.empty(nb, H, device=dev))
def args(o):
    a = dict(by_dst=g_tt.by_dst, xl=xp[:, :hc], xr=xp[:, hc:2 * hc], att=att[0], bias=bias[0], heads=H, channels=C, out=o["y_tx"],
             pre=o["p_tx"], lse=o["l_tx"], apply_gelu=True, dropout_p=0.2, seed=(0, step), keep_bits=bits[0][0])
    c = dict(by_dst=g_tb.by_dst, xl=xp[:, 2 * hc:], xr=xb, att=att[1], bias=bias[1], heads=H, channels=C, out=o["y_bd"],
             pre=o["p_bd"], lse=o["l_bd"], apply_gelu=True, dropout_p=0.2, seed=(1, step), keep_bits=bits[1][0])
    return a, c
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); e = torch.cuda.Event(True); a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it
o1, o2 = outs(), outs()
a1, c1 = args(o1); a2, c2 = args(o2)
def two():
    ops.gatv2_fwd_launch(**a1); ops.gatv2_fwd_launch(**c1)
def one():
    ops.gatv2_fwd_pair_launch(a2, c2)
two(); one(); torch.cuda.synchronize()
same = all(torch.equal(o1[k], o2[k]) for k in o1)
for r in range(3):
    print(f"round {r}: tt alone {t(lambda: ops.gatv2_fwd_launch(**a1)):.3f} ms  tb alone {t(lambda: ops.gatv2_fwd_launch(**c1)):.3f} ms  "
          f"two launches {t(two):.3f} ms  one launch {t(one):.3f} ms  identical outputs: {same}", flush=True)
