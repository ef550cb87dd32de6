"""dW = gy^T x via batched slab GEMMs: sweep the slab count for the step's shapes."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = 'cuda'
n = 1_000_000
def t(f, it=15):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e3
for M, K in ((384, 128), (384, 256), (128, 128), (64, 128), (64, 256)):
    rows = n if not (M == 64 and K == 256) else 2 * n
    gy = torch.randn(rows, M, device=dev).bfloat16(); x = torch.randn(rows, K, device=dev).bfloat16()
    res = []
    for s in (32, 64, 128, 256, 512, 1024):
        nn = rows // s * s
        f = lambda: torch.bmm(gy[:nn].reshape(s, nn // s, M).transpose(1, 2), x[:nn].reshape(s, nn // s, K)).sum(0, dtype=torch.float32)
        res.append((s, round(t(f))))
    gb = rows * (M + K) * 2 / 1e9
    print(f"dW M={M} K={K} rows={rows} ({gb:.2f} GB): " + "  ".join(f"s={s}:{us}us" for s, us in res), flush=True)
