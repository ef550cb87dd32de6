"""Micro-benchmarks of the row-wise helper kernels at C2 size: embedding-table gradient, column sums."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segger_amd import ops
from segger_amd.synthetic import SyntheticSpec, make_fov

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


n = 1_000_000
data = make_fov(SyntheticSpec(n_tx=n, n_bd=10_000, k_tx=4), dev)
ids = data["tx"]["x"]
table = torch.randn(256, 128, device=dev, requires_grad=True)
pe = torch.randn(n, 128, device=dev).to(torch.bfloat16).requires_grad_(True)
g = torch.randn(n, 256, device=dev).to(torch.bfloat16)


def emb():
    table.grad = None; pe.grad = None
    ops.embed_gelu(table, ids, pe).backward(g)


ms_all = timeit(emb)
with torch.no_grad():
    ms_fwd = timeit(lambda: ops.embed_gelu(table, ids, pe))
print(f"embed_gelu fwd {ms_fwd*1e3:.0f} us, fwd+bwd {ms_all*1e3:.0f} us")
rid = torch.randint(0, 256, (n,), device=dev, dtype=torch.int32)
ids_keep = ids
ids = rid
print(f"  with uniformly random ids: fwd+bwd {timeit(emb)*1e3:.0f} us")
ids = ids_keep

for rows, cols in ((n, 384), (2 * n, 64), (n, 128), (n, 64)):
    x = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
    t1 = timeit(lambda: ops.colsum(x))
    t2 = timeit(lambda: x.sum(0, dtype=torch.float32))
    gb = rows * cols * 2 / 1e9
    print(f"colsum [{rows},{cols}] bf16: hip {t1*1e3:.0f} us ({gb/t1:.0f} GB/s... x1e3)  torch {t2*1e3:.0f} us")

x = torch.randn(n, 384, device=dev).to(torch.bfloat16)
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x))
print(f"copy 768 MB: {t*1e3:.0f} us -> {2*x.numel()*2/t/1e9:.0f} GB/s (r+w)")
t = timeit(lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32))
print(f"sum-all 768 MB: {t*1e3:.0f} us -> {x.numel()*2/t/1e9:.0f} GB/s (read)")
