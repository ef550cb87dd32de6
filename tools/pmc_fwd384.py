"""Two launches of the fp32 stacked forward 128 -> 384 at 1M rows (for stamp / counter builds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
x = torch.randn(n, 128, device=dev, generator=g)
w = torch.randn(384, 128, device=dev, generator=g) / 128 ** 0.5
b = torch.randn(384, device=dev, generator=g)
w3 = ops.f32_split_planes(w)
for _ in range(2):
    ops.linear_f32_split_launch(x, w3, b)
torch.cuda.synchronize()
