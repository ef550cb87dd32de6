"""Times the projection kernels of one encoder layer at C2 size (1M rows, bf16): forward / dX (segger_linear_fwd),
dW + db (segger_linear_wgrad) and the vendor-GEMM + column-sum route it replaces."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
dev = torch.device('cuda')
n = int(os.environ.get('N', 1_000_000))
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); e = torch.cuda.Event(True); a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it
g = torch.Generator(device=dev).manual_seed(0)
for m, k in ((384, 256), (384, 128), (64, 128), (64, 256), (64, 64), (128, 128)):
    rows = n * (2 if (m, k) in ((64, 256), (64, 64)) else 1)
    x = torch.randn(rows, k, device=dev, generator=g).bfloat16()
    gy = torch.randn(rows, m, device=dev, generator=g).bfloat16()
    w = torch.randn(m, k, device=dev, generator=g).bfloat16()
    wt = w.t().contiguous()
    gb = (rows * (m + k) * 2) / 1e9
    ms_f = t(lambda: ops.linear_fwd_launch(x, w, None))
    ms_dx = t(lambda: ops.linear_fwd_launch(gy, wt, None)) if ops.linear_supported(m, k, torch.bfloat16) else float('nan')
    ms_w = t(lambda: ops.linear_wgrad_launch(gy, x))
    ms_fused = (t(lambda: ops.linear_wgrad_dx_launch(gy, x, wt)) if ops.linear_wgrad_dx_supported(m, k, torch.bfloat16)
                else float('nan'))
    gb_fused = (rows * (m + 2 * k) * 2) / 1e9
    ms_old = t(lambda: ((gy.t() @ x).float(), ops.colsum(gy)))
    print(f"rows {rows} M {m} K {k}: fwd {ms_f:.3f} ms ({gb / ms_f:.2f} TB/s)  dX {ms_dx:.3f} ms  "
          f"wgrad {ms_w:.3f} ms ({gb / ms_w:.2f} TB/s, {2 * rows * m * k / ms_w / 1e9:.0f} TFLOP/s)  "
          f"dX+dW+db in one pass {ms_fused:.3f} ms ({gb_fused / ms_fused:.2f} TB/s; separate {ms_dx + ms_w:.3f})  "
          f"gemm+colsum {ms_old:.3f} ms", flush=True)

# fp32 storage: exact-fp32 MFMA kernels (csrc/linear_f32.hip) against the vendor GEMM they replace
for m, k in ((384, 256), (384, 128), (64, 128), (64, 256)):
    x = torch.randn(n, k, device=dev, generator=g)
    gy = torch.randn(n, m, device=dev, generator=g)
    w = torch.randn(m, k, device=dev, generator=g)
    flops = 2 * n * m * k
    ms_f = t(lambda: ops.linear_fwd_launch(x, w, None), it=10)
    ms_w = t(lambda: ops.linear_wgrad_launch(gy, x), it=10)
    ms_vf = t(lambda: torch.nn.functional.linear(x, w), it=10)
    ms_vw = t(lambda: (gy.t() @ x, gy.sum(0)), it=10)
    print(f"f32 rows {n} M {m} K {k}: fwd {ms_f:.3f} ms ({flops / ms_f / 1e9:.0f} TFLOP/s, {n * (m + k) * 4 / ms_f / 1e9:.2f} TB/s; "
          f"vendor {ms_vf:.3f})  wgrad {ms_w:.3f} ms ({flops / ms_w / 1e9:.0f} TFLOP/s; vendor gemm + sum {ms_vw:.3f})", flush=True)
