"""Where the HOST time of an eager small-batch training step goes (segger's default 1M-edge batches; the step is host-bound):
per phase (batch assembly, forward, losses, backward, optimizer) and per autograd Function of segger_amd.ops, forward and
backward separately -- cProfile does not see the autograd engine's worker thread, where every backward runs.
Wall-clock spans on the issuing thread, no synchronisation inside a step.   N_TX, N_BD, STEPS"""
import collections, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import LitISTEncoder, ops
from segger_amd.fov import build_fov_batches
from segger_amd.synthetic import SyntheticSpec

dev = torch.device('cuda')
spec = SyntheticSpec(n_tx=int(os.environ.get('N_TX', 3_000_000)), n_bd=int(os.environ.get('N_BD', 30_000)), k_tx=15, seed=0)
part, batches, aux, tiling = build_fov_batches(spec, dev)
torch.manual_seed(0)
model = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
model.model._materialize_bd(spec.bd_dim, "cpu")
model.model.compute_dtype = {'bf16': torch.bfloat16, 'f32': torch.float32}[os.environ.get('DTYPE', 'bf16')]
model = model.to(dev)
model.set_similarities(aux["tx_similarity"].to(dev), aux["bd_similarity"].to(dev))
model._max_epochs_override, model.current_epoch = 20, 10
model.train()
opt = model.configure_optimizers()

acc = collections.defaultdict(lambda: [0, 0.0])
on = [False]


def timed(name, fn):
    def w(*a, **k):
        if not on[0]:
            return fn(*a, **k)
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[name]; e[0] += 1; e[1] += time.perf_counter() - t
    return w


for k, v in list(vars(ops).items()):
    if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
        v.forward = staticmethod(timed(f"{k}.forward", v.forward))
        v.backward = staticmethod(timed(f"{k}.backward", v.backward))
for name in ("adam_step", "_refresh_stale_packs", "stage", "triplet_sample", "sample_negatives", "dropout_bits_many", "step_advance",
             "rows_by_id", "segment_minmax"):
    setattr(ops, name, timed(name, getattr(ops, name)))

# PROFILE_FN=_HeteroGatLayer.backward : cProfile INSIDE that function (on whichever thread runs it), printed at the end
prof = None
if os.environ.get("PROFILE_FN"):
    import cProfile, pstats, io
    cname, mname = os.environ["PROFILE_FN"].split(".")
    cls = getattr(ops, cname)
    inner = getattr(cls, mname)
    prof = cProfile.Profile()

    def profiled(*a, **k):
        if not on[0]:
            return inner(*a, **k)
        prof.enable()
        try:
            return inner(*a, **k)
        finally:
            prof.disable()
    setattr(cls, mname, staticmethod(profiled))

phases = collections.defaultdict(float)


def step(k, i):
    t0 = time.perf_counter()
    b = part.batch(batches[k])
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(b, i)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    if on[0]:
        phases["batch"] += t1 - t0; phases["forward+losses"] += t2 - t1; phases["backward"] += t3 - t2; phases["optimizer"] += t4 - t3


n = min(len(batches), int(os.environ.get("STEPS", 60)))
for ep in range(2):
    for k in range(n): step(k, k)
torch.cuda.synchronize()
on[0] = True
t = time.perf_counter()
for k in range(n): step(k, k)
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / n
print(f"{n} steps, {wall * 1e3:.3f} ms / step wall")
for k, v in phases.items():
    print(f"  {k:16s} {v / n * 1e6:8.1f} us / step")
print("per function (host, us / step; calls / step):")
for k, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:40s} {s / n * 1e6:8.1f}   {c / n:5.1f}")
if prof is not None:
    st = io.StringIO()
    pstats.Stats(prof, stream=st).sort_stats("tottime").print_stats(25)
    print("\n".join(l[:140] for l in st.getvalue().splitlines()))
