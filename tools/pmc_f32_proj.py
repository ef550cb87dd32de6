"""Script for rocprofv3 --pmc passes over the fp32 projection kernels of one layer at C2 size (W-resident forward, gated data
gradient, pipelined weight gradient): a few launches each."""
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
gy = torch.randn(n, 384, device=dev, generator=g)
gate = torch.randn(n, 128, device=dev, generator=g)
for _ in range(3):
    ops.linear_f32_split_launch(x, w3, b)
    ops.linear_f32_gate_launch(gy, w.t().contiguous(), gate, "gelu")
    ops.linear_wgrad_launch(gy, x)
torch.cuda.synchronize()
