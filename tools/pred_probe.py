import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from segger_amd import LitISTEncoder
from segger_amd.synthetic import SyntheticSpec, make_graph
dev = torch.device("cuda")
spec = SyntheticSpec(n_tx=1_000_000, n_bd=10_000, k_tx=15, seed=0)
b, aux = make_graph(spec, return_aux=True)
batch = b.to(dev)
torch.manual_seed(0)
m = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
m.model._materialize_bd(spec.bd_dim, "cpu")
m.model.compute_dtype = torch.bfloat16
m = m.to(dev).eval()
with torch.no_grad():
    for _ in range(3): m.predict_step(batch, 0)
    torch.cuda.synchronize()
    for tag in ("full", "fwd"):
        t0 = time.perf_counter()
        for _ in range(10):
            if tag == "full": m.predict_step(batch, 0)
            else: m(batch)
        torch.cuda.synchronize()
        print(tag, (time.perf_counter() - t0) / 10 * 1e3, "ms")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3): m.predict_step(batch, 0)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
