"""A few launches of the fp32 split weight gradient (384, 128) at 1M rows (for stamp / counter builds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
gy = torch.randn(n, 384, device=dev, generator=g)
x = torch.randn(n, 128, device=dev, generator=g)
for _ in range(2):
    ops.linear_wgrad_launch(gy, x)
torch.cuda.synchronize()
