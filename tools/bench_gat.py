"""Times the three GATv2 kernels on the C2 tile (tx-neighbors-tx, H=2, C=64; DTYPE=bf16|f16|f32)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops, TX_TX
from segger_amd.graph import build_edge_graph
from segger_amd.synthetic import SyntheticSpec, make_graph
dev = torch.device('cuda')
n = int(os.environ.get('N_TX', 1_000_000))
b = make_graph(SyntheticSpec(n_tx=n, n_bd=n // 100, k_tx=15, seed=0))
ei = b[TX_TX].edge_index.to(dev)
mode = os.environ.get('GRAPH', 'knn')            # locality experiments: same degrees, other neighbours
if mode == 'band':                               # neighbours = the adjacent rows: every gather hits L1 / L2
    dst = ei[1]
    off = torch.arange(ei.shape[1], device=dev) % 15 - 7
    ei = torch.stack([(dst + off).clamp_(0, n - 1), dst])
elif mode == 'random':                           # uniformly random neighbours: every gather misses
    ei = torch.stack([torch.randint(0, n, (ei.shape[1],), device=dev), ei[1]])
g = build_edge_graph(ei, n, n)
if os.environ.get('ORDER', '1') == '0':          # A/B of the degree-balanced visiting order
    g.by_dst.order = g.by_src.order = None
if os.environ.get('LDSG', '0') == '1':           # block tables: the forward gathers from LDS
    g.by_dst.block_tables()
    print('block tables:', g.by_dst.tables is not None, flush=True)
H, C = 2, 64; hc = H * C
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[os.environ.get('DTYPE', 'bf16')]
gen = torch.Generator(device=dev).manual_seed(0)
xp = torch.randn(n, 3 * hc, device=dev, generator=gen).to(dt)
att = torch.randn(hc, device=dev, generator=gen) * 0.3
bias = torch.zeros(hc, device=dev)
out = torch.empty(n, hc, dtype=dt, device=dev); pre = torch.empty_like(out)
lse = torch.empty(n, H, device=dev)
gy = torch.randn(n, hc, device=dev, generator=gen).to(dt); gxp = torch.empty_like(xp)
p = float(os.environ.get('DROP', 0.0))
bits = None
if os.environ.get('BITS', '0') == '1' and p > 0:    # dropout mask as precomputed bit planes (ops.dropout_bits)
    bits = (ops.dropout_bits(g.by_dst, H, p, [5])[0], ops.dropout_bits(g.by_src, H, p, [5])[0])
fwd = lambda: ops.gatv2_fwd_launch(g.by_dst, xp[:, :hc], xp[:, hc:2*hc], att, bias, H, C, out, pre=pre, lse=lse, apply_gelu=True, dropout_p=p, seed=5, keep_bits=None if bits is None else bits[0])
bwd = lambda: ops.gatv2_bwd_launch(g, xp[:, :hc], xp[:, hc:2*hc], att, bias, H, C, gy, pre, lse, gxp[:, :hc], gxp[:, hc:2*hc], apply_gelu=True, dropout_p=p, seed=5, keep_bits=bits)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); e = torch.cuda.Event(True); a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it
print(os.environ.get('DTYPE', 'bf16'), mode, os.path.basename(os.environ.get('SEGGER_AMD_LIB', 'default')), 'bits', os.environ.get('BITS', '0'), 'drop', p, 'fwd %.3f ms  bwd(dst+src) %.3f ms' % (t(fwd), t(bwd)), flush=True)
