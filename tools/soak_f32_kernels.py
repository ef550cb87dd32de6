"""Run-to-run determinism soak of the round-5 fp32 projection kernels: the same inputs ITERS times, every output compared bit for
bit with the first (the W-resident kernels rotate two LDS buffers behind one barrier per tile, the pipelined weight gradient
writes the next stage's planes under the current stage's reads: a missing ordering would show as a rare mismatch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
dev = torch.device("cuda")
iters = int(os.environ.get("ITERS", 300))
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
for n in (1_000_003, 40_037, 8192):
    x = torch.randn(n, 128, device=dev, generator=g)
    w = torch.randn(384, 128, device=dev, generator=g) / 128 ** 0.5
    b = torch.randn(384, device=dev, generator=g)
    w3 = ops.f32_split_planes(w)
    gy = torch.randn(n, 384, device=dev, generator=g)
    gate = torch.randn(n, 128, device=dev, generator=g)
    wt = w.t().contiguous()
    cases = {
        "forward 128->384": lambda: ops.linear_f32_split_launch(x, w3, b),
        "dX 384->128 gelu gate": lambda: ops.linear_f32_gate_launch(gy, wt, gate, "gelu"),
        "dX 384->128 silu gate": lambda: ops.linear_f32_gate_launch(gy, wt, gate, "silu"),
        "dX 384->128": lambda: ops.linear_f32_split_launch(gy, ops.f32_split_planes(wt), None),
        "dW (384,128)": lambda: torch.cat([t.flatten() for t in ops.linear_wgrad_launch(gy, x)]),
    }
    for name, fn in cases.items():
        ref = fn().clone()
        k = max(20, iters if n < 100_000 else iters // 6)
        mism = 0
        for _ in range(k):
            if not torch.equal(fn(), ref):
                mism += 1
        bad += mism
        print(f"n={n:8d} {name:24s}: {k} runs, {mism} differ", flush=True)
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
