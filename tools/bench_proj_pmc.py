"""A handful of launches of the projection kernels at C2 size for rocprofv3 --pmc passes (tools/pmc.sh <dir> tools/bench_proj_pmc.py):
bf16 forward / one-pass backward, fp32 forward / dX / weight gradient."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
dev = torch.device("cuda")
n = int(os.environ.get("N", 1_000_000))
g = torch.Generator(device=dev).manual_seed(0)
m, k = 384, 128
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(n, k, device=dev, generator=g).to(dt)
    gy = torch.randn(n, m, device=dev, generator=g).to(dt)
    w = (torch.randn(m, k, device=dev, generator=g) / k ** 0.5).to(dt)
    wt = w.t().contiguous()
    for _ in range(4):
        ops.linear_fwd_launch(x, w, None)
        ops.linear_fwd_launch(gy, wt, None)
        ops.linear_wgrad_launch(gy, x)
        if dt == torch.bfloat16:
            ops.linear_wgrad_dx_launch(gy, x, wt)
    torch.cuda.synchronize()
