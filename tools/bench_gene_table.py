"""Times segger_gene_table_fwd / _bwd at segger's sizes (G genes, D = 128, three 128-row weights)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import _lib, ops
dev = torch.device("cuda"); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
G, D = int(os.environ.get("G", 256)), 128
table = torch.randn(G, D, device=dev, generator=g)
ws = [torch.randn(128, 2 * D, device=dev, generator=g) for _ in range(3)]
bs = [torch.randn(128, device=dev, generator=g) for _ in range(3)]
a = ops._gene_table_args(table, ws, bs, torch.bfloat16)
M = 384
tab = torch.empty(G, M, dtype=torch.bfloat16, device=dev); wc = torch.empty(M, D, dtype=torch.bfloat16, device=dev); wct = torch.empty(D, M, dtype=torch.bfloat16, device=dev)
a.tab, a.ld_tab, a.wc, a.wc_t = tab.data_ptr(), M, wc.data_ptr(), wct.data_ptr()
gt = torch.randn(G, M, device=dev, generator=g); gw = torch.randn(M, D, device=dev, generator=g)
gtab = torch.empty_like(table); gws = [torch.empty(128, 2 * D, device=dev) for _ in range(3)]; gbs = [torch.empty(128, device=dev) for _ in range(3)]
a.g_tab, a.g_wc, a.g_table = gt.data_ptr(), gw.data_ptr(), gtab.data_ptr()
for i in range(3): a.g_w[i], a.g_b[i] = gws[i].data_ptr(), gbs[i].data_ptr()
st = torch.cuda.current_stream().cuda_stream
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(True), torch.cuda.Event(True); s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3
print(f"G={G}: gene_table_fwd {t(lambda: lib.segger_gene_table_fwd(C.byref(a), st)):.1f} us, gene_table_bwd {t(lambda: lib.segger_gene_table_bwd(C.byref(a), st)):.1f} us")
W = torch.cat(ws, 0); ref = torch.nn.functional.gelu(table) @ W[:, :D].t() + torch.cat(bs)
print("fwd max err", float((tab.float() - ref).abs().max()), "dE err", float((gtab - (gt @ W[:, :D]) * (torch.autograd.functional.jacobian(lambda x: torch.nn.functional.gelu(x).sum(), table))).abs().max()))
