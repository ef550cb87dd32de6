"""The loss head alone at C2 size (1M transcripts, 10k boundaries, 394k segmentation triplets): forward / backward times of
the kernel-by-kernel head and of the one-launch head (forced: ops routes batches of more than LOSS_HEAD_ONE_LAUNCH_MAX_ROWS
transcripts to the former).   N, NB, E override the sizes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops
from segger_amd.graph import csr_from_coo

dev = torch.device("cuda")
n, nb, e, C = int(os.environ.get("N", 1_000_000)), int(os.environ.get("NB", 10_000)), int(os.environ.get("E", 394_000)), 64
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
y0 = torch.randn(n, C, device=dev, generator=g).to(dt)
yb0 = torch.randn(nb, C, device=dev, generator=g).to(dt)
pos = torch.randint(0, n, (n,), device=dev, generator=g); neg = torch.randint(0, n, (n,), device=dev, generator=g)
bpos = torch.randint(0, nb, (nb,), device=dev, generator=g); bneg = torch.randint(0, nb, (nb,), device=dev, generator=g)
dp, dn = torch.rand(nb, device=dev, generator=g), torch.rand(nb, device=dev, generator=g)
w = torch.full((nb,), 1.0 / nb, device=dev)
src = torch.randperm(n, device=dev, generator=g)[:e]
dst = torch.randint(0, nb, (e,), device=dev, generator=g)
dneg = (dst + torch.randint(1, nb, (e,), device=dev, generator=g)) % nb
groups = csr_from_coo(dst, src, nb, n, validate=False)
a, b = torch.ones(3, device=dev), torch.tensor([0.5, 0.2, 0.3], device=dev)
hint = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev)
iota = torch.arange(n, device=dev)


def run(reps=10):
    tf = tb = 0.0
    for i in range(reps + 2):
        y, yb = y0.clone().requires_grad_(True), yb0.clone().requires_grad_(True)
        zs = ops.l2_normalize_many({"tx": y, "bd": yb})
        spec = ops.LossHeadSpec((iota, pos, neg, 0.3, 1e-6), (bpos, bneg, dp, dn, w, 1e-8),
                                (src, dst, dneg, 0.4, 1e-6, groups, True), tx_anchors_are_rows=True, grad_out_hint=hint)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out = ops.loss_head(zs["tx"], zs["bd"], a, b, spec)
        e1.record()
        out.backward(hint)
        e2.record()
        torch.cuda.synchronize()
        if i >= 2:
            tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
    return tf / reps, tb / reps, float(out[3])


for name, fl in [("kernel-by-kernel", dict(ONE_LAUNCH_LOSS_HEAD=False)),
                 ("one-launch (chains)", dict(LOSS_HEAD_ONE_LAUNCH_MAX_ROWS=1 << 40))]:
    keep = {k: getattr(ops, k) for k in fl}
    for k, v in fl.items():
        setattr(ops, k, v)
    f, bw, loss = run()
    print(f"{name:24s} forward {f * 1e3:8.1f} us   backward (incl. l2norm) {bw * 1e3:8.1f} us   loss {loss:.5f}", flush=True)
    for k, v in keep.items():
        setattr(ops, k, v)
