"""fp32-storage projections: the exact-fp32 MFMA kernel (segger_linear_fwd, dtype f32) against the three-way bf16 split
(segger_linear_fwd_f32_split) -- time at C2 size and error against fp64, for the forward (128 -> 384) and the data
gradient (384 -> 128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
dev = torch.device("cuda")
n = int(os.environ.get("N", 1_000_000))
g = torch.Generator(device=dev).manual_seed(0)


def t(fn, it=10):
    for _ in range(3): fn()
    a, e = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / it


only = int(os.environ.get("ONLY", 0))            # ONLY=384: just the data-gradient shape
for k, m in ((128, 384), (384, 128), (128, 64), (128, 128)):
    if only and k != only:
        continue
    x = torch.randn(n, k, device=dev, generator=g) * torch.rand(n, 1, device=dev, generator=g) * 3
    w = torch.randn(m, k, device=dev, generator=g) / k ** 0.5
    b = torch.randn(m, device=dev, generator=g)
    w3 = ops.f32_split_planes(w)
    y_exact = ops.linear_fwd_launch(x, w, b)
    y_split = ops.linear_f32_split_launch(x, w3, b)
    idx = torch.randint(0, n, (20000,), device=dev, generator=g)
    ref = x[idx].double() @ w.double().t() + b.double()
    bound = (x[idx].double().abs() @ w.double().abs().t() + b.double().abs())          # sum |x||w|: what rounding scales with
    e_exact = ((y_exact[idx].double() - ref).abs() / bound).max().item()
    e_split = ((y_split[idx].double() - ref).abs() / bound).max().item()
    e_torch = (((x[idx] @ w.t() + b).double() - ref).abs() / bound).max().item()
    t_exact = t(lambda: ops.linear_fwd_launch(x, w, b))
    t_split = t(lambda: ops.linear_f32_split_launch(x, w3, b))
    if k == 384:                                  # the data gradient through the layer input's GELU (gate epilogue)
        gate = torch.randn(n, m, device=dev, generator=g)
        if os.environ.get("GATE_ONLY"):
            print(f"  gate epilogue: gelu {t(lambda: ops.linear_f32_gate_launch(x, w, gate, 'gelu')):.3f} ms, "
                  f"silu {t(lambda: ops.linear_f32_gate_launch(x, w, gate, 'silu')):.3f} ms", flush=True)
            continue
        yg = ops.linear_f32_gate_launch(x, w, gate, "gelu")
        refg = (x[idx].double() @ w.double().t()) * torch.ops.aten.gelu_backward(torch.ones(len(idx), m, device=dev, dtype=torch.float64), gate[idx].double())
        e_g = ((yg[idx].double() - refg).abs() / bound).max().item()
        print(f"  with the GELU gate epilogue: {t(lambda: ops.linear_f32_gate_launch(x, w, gate, 'gelu')):.3f} ms (incl. the plane split of w), "
              f"err {e_g:.2e}; silu {t(lambda: ops.linear_f32_gate_launch(x, w, gate, 'silu')):.3f} ms", flush=True)
    print(f"{n} x {k} -> {m}: exact-fp32 MFMA {t_exact:.3f} ms, bf16x3 split {t_split:.3f} ms; "
          f"max |err| / sum|x||w|: exact {e_exact:.2e}, split {e_split:.2e}, torch fp32 matmul {e_torch:.2e}  (2^-24 = 6.0e-08)", flush=True)
