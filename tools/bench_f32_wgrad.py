"""Times the fp32 weight-gradient kernels (exact-fp32 MFMA vs bf16x3 split) at C2 size:  python tools/bench_f32_wgrad.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
n = int(os.environ.get("N", 1_000_000))
def t(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); e = torch.cuda.Event(True); a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it
for m, k in ((384, 128), (128, 128), (64, 256), (64, 128), (64, 64), (128, 256)):
    gy = torch.randn(n, m, device=dev, generator=g)
    x = torch.randn(n, k, device=dev, generator=g)
    res = {}
    for split in (False, True):
        ops.F32_SPLIT = split
        res[split] = (t(lambda: ops.linear_wgrad_launch(gy, x)), ops.linear_wgrad_launch(gy, x)[0])
    ref = gy.double().t() @ x.double()
    bound = gy.double().abs().t() @ x.double().abs()
    err = {s: ((res[s][1].double() - ref).abs() / bound).max().item() for s in res}
    print(f"{n} x ({m}, {k}): exact-fp32 MFMA {res[False][0]:.3f} ms, bf16x3 split {res[True][0]:.3f} ms; max |err| / sum|dy||x|: "
          f"exact {err[False]:.2e}, split {err[True]:.2e}", flush=True)
