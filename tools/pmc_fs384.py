"""One script for tools/pmc_sq.sh: a few launches of the 384 -> 128 fp32 data gradient with its GELU gate (1M rows)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
x = torch.randn(n, 384, device=dev, generator=g)
w = torch.randn(128, 384, device=dev, generator=g) / 384 ** 0.5
gate = torch.randn(n, 128, device=dev, generator=g)
for _ in range(4):
    ops.linear_f32_gate_launch(x, w, gate, "gelu")
torch.cuda.synchronize()
