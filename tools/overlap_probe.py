"""Bounding experiment (round-5 review, item 5): do a VALU-bound aggregation kernel and a memory-bound projection kernel
overlap when launched on two HIP streams?  C2 layer sizes, bf16.  For every pair (A, B): t_A and t_B alone, then A on stream
1 and B on stream 2 at the same time, REPS rounds; wall per round against max(t_A, t_B) and t_A + t_B.  Also A on the FIRST
half of the rows beside B on the SECOND half (the two-chunk pipeline's inner stage)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops, TX_TX
from segger_amd.graph import build_edge_graph
from segger_amd.synthetic import SyntheticSpec, make_graph

dev = torch.device("cuda")
n = int(os.environ.get("N_TX", 1_000_000)); H, C = 2, 64; hc = H * C; dt = torch.bfloat16
b = make_graph(SyntheticSpec(n_tx=n, n_bd=n // 100, k_tx=15, seed=0))
g = build_edge_graph(b[TX_TX].edge_index.to(dev), n, n)
gen = torch.Generator(device=dev).manual_seed(0)
xp = torch.randn(n, 3 * hc, device=dev, generator=gen).to(dt)
att = torch.randn(hc, device=dev, generator=gen) * 0.3; bias = torch.zeros(hc, device=dev)
out = torch.empty(n, hc, dtype=dt, device=dev); pre = torch.empty_like(out); lse = torch.empty(n, H, device=dev)
gy = torch.randn(n, hc, device=dev, generator=gen).to(dt); gxp = torch.empty_like(xp)
bits = (ops.dropout_bits(g.by_dst, H, 0.2, [5])[0], ops.dropout_bits(g.by_src, H, 0.2, [5])[0])
x2 = torch.randn(n, hc, device=dev, generator=gen).to(dt)
w = (torch.randn(3 * hc, hc, device=dev, generator=gen) * 0.05).to(dt); wt = w.t().contiguous()
y2 = torch.empty(n, 3 * hc, dtype=dt, device=dev)
gy2 = torch.randn(n, 3 * hc, device=dev, generator=gen).to(dt)

fwd = lambda: ops.gatv2_fwd_launch(g.by_dst, xp[:, :hc], xp[:, hc:2*hc], att, bias, H, C, out, pre=pre, lse=lse, apply_gelu=True,
                                   dropout_p=0.2, seed=5, keep_bits=bits[0])
fwd()
sc = None
def bwd(p):
    global sc
    ops.gatv2_bwd_launch(g, xp[:, :hc], xp[:, hc:2*hc], att, bias, H, C, gy, pre, lse, gxp[:, :hc], gxp[:, hc:2*hc], apply_gelu=True,
                         dropout_p=0.2, seed=5, keep_bits=bits, passes=p, scratch=sc)
    sc = ops.gatv2_bwd_launch.scratch
bwd(1)
lin = lambda: ops.linear_fwd_launch(x2, w, None, out=y2)
wg = lambda: ops.linear_wgrad_dx_launch(gy2, x2, wt)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
REPS = 10


def alone(fn, stream):
    with torch.cuda.stream(stream):
        for _ in range(3): fn()
        a, e = torch.cuda.Event(True), torch.cuda.Event(True); a.record()
        for _ in range(REPS): fn()
        e.record()
    torch.cuda.synchronize(); return a.elapsed_time(e) / REPS


def both(fa, fb):
    torch.cuda.synchronize()
    st = torch.cuda.Event(True); st.record()
    s1.wait_event(st); s2.wait_event(st)
    with torch.cuda.stream(s1):
        for _ in range(REPS): fa()
        e1 = torch.cuda.Event(True); e1.record()
    with torch.cuda.stream(s2):
        for _ in range(REPS): fb()
        e2 = torch.cuda.Event(True); e2.record()
    torch.cuda.synchronize()
    return max(st.elapsed_time(e1), st.elapsed_time(e2)) / REPS, st.elapsed_time(e1) / REPS, st.elapsed_time(e2) / REPS


for name, fa, fb in (("gatv2_fwd || linear_res 128->384", fwd, lin), ("gatv2_bwd_dst || wgrad_dx (384,128)", lambda: bwd(1), wg),
                     ("gatv2_bwd_src || linear_res 128->384", lambda: bwd(2), lin), ("gatv2_fwd || gatv2_fwd (control)", fwd, fwd)):
    ta, tb = alone(fa, s1), alone(fb, s2)
    both(fa, fb)
    wall, w1, w2 = both(fa, fb)
    print(f"{name:40s} t_A {ta:.3f}  t_B {tb:.3f}  sum {ta + tb:.3f}  max {max(ta, tb):.3f}  co-run wall {wall:.3f} "
          f"(A done {w1:.3f}, B done {w2:.3f})  wall/max {wall / max(ta, tb):.2f}  wall/sum {wall / (ta + tb):.2f}", flush=True)
