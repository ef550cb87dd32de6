"""Where the aggregation kernels' VALU instructions go: per-row vs per-edge-batch coefficients and the padding share.

    MODE=time  python tools/valu_budget.py          20 launches per kernel and configuration, HIP events
    MODE=count rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ... -- python3 tools/valu_budget.py
                                                    ONE launch per kernel and configuration (dispatch order = config order)

Configurations: the C2 tile (1M transcripts, 10k nuclei, bf16, dropout 0.2 as bit planes) with k in {5, 10, 15, 30};
k = 15 also without GELU and with the degree-balanced row order at windows 16 / 64.  For every configuration the script
derives from the graph itself what the kernels' loops do: a wave owns 4 consecutive (or order[]-consecutive) rows, one per
16-lane group, and walks 4-edge batches until its LONGEST row is done, so it issues  iters(wave) = max_g ceil(deg_g / 4)
batch iterations of 16 edge slots each.  Written to gpurun_out/valu_budget/<MODE>.json; tools/valu_budget_fit.py joins the
two with the counter CSV and fits  VALU = P * waves + B * sum_iters.
"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import ops, TX_TX
from segger_amd.graph import build_edge_graph
from segger_amd.synthetic import SyntheticSpec, make_graph

dev = torch.device("cuda")
mode = os.environ.get("MODE", "time")
n = int(os.environ.get("N_TX", 1_000_000))
H, C = 2, 64; hc = H * C
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("DTYPE", "bf16")]
drop = float(os.environ.get("DROP", 0.2))
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "valu_budget"); os.makedirs(out_dir, exist_ok=True)


def slots(csr):
    """(waves, sum over waves of batch iterations, sum over rows of ceil(deg/4)) for the visiting order of this view."""
    deg = (csr.indptr[1:] - csr.indptr[:-1]).to(torch.int64)
    if csr.order is not None:
        deg = deg[csr.order.long()]
    it = (deg + 3) // 4
    pad = (-it.numel()) % 4
    itp = torch.cat([it, it.new_zeros(pad)]).view(-1, 4)
    return int(itp.shape[0]), int(itp.max(1).values.sum()), int(it.sum()), deg


def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); e = torch.cuda.Event(True); a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it


configs = [dict(k=5), dict(k=10), dict(k=15), dict(k=30), dict(k=15, gelu=0), dict(k=15, window=16), dict(k=15, window=64),
           dict(k=15, drop=0.0)]
gen = torch.Generator(device=dev).manual_seed(0)
xp = torch.randn(n, 3 * hc, device=dev, generator=gen).to(dt)
att = torch.randn(hc, device=dev, generator=gen) * 0.3
bias = torch.zeros(hc, device=dev)
out = torch.empty(n, hc, dtype=dt, device=dev); pre = torch.empty_like(out)
lse = torch.empty(n, H, device=dev)
gy = torch.randn(n, hc, device=dev, generator=gen).to(dt); gxp = torch.empty_like(xp)
records, graphs = [], {}
for cfg in configs:
    k, gelu, window, p = cfg["k"], bool(cfg.get("gelu", 1)), cfg.get("window", 0), cfg.get("drop", drop)
    if k not in graphs:
        b = make_graph(SyntheticSpec(n_tx=n, n_bd=n // 100, k_tx=k, seed=0))
        graphs[k] = b[TX_TX].edge_index.to(dev)
    g = build_edge_graph(graphs[k], n, n)
    g.by_dst.order = g.by_src.order = None
    if window:
        g.by_dst.balanced_order(window); g.by_src.balanced_order(window)
    bits = None
    if p > 0:
        bits = (ops.dropout_bits(g.by_dst, H, p, [5])[0], ops.dropout_bits(g.by_src, H, p, [5])[0])
    fwd = lambda: ops.gatv2_fwd_launch(g.by_dst, xp[:, :hc], xp[:, hc:2*hc], att, bias, H, C, out, pre=pre, lse=lse,
                                       apply_gelu=gelu, dropout_p=p, seed=5, keep_bits=None if bits is None else bits[0])
    fwd(); scratch = None
    def bwd(passes, scratch=None):
        ops.gatv2_bwd_launch(g, xp[:, :hc], xp[:, hc:2*hc], att, bias, H, C, gy, pre if gelu else out, lse, gxp[:, :hc], gxp[:, hc:2*hc],
                             apply_gelu=gelu, dropout_p=p, seed=5, keep_bits=bits, passes=passes, scratch=scratch)
        return ops.gatv2_bwd_launch.scratch
    rec = dict(cfg, gelu=int(gelu), window=window, drop=p, n_rows=n, n_edges=int(graphs[k].shape[1]))
    for name, view in (("dst", g.by_dst), ("src", g.by_src)):
        w, si, sr, deg = slots(view)
        rec[name] = {"waves": w, "sum_wave_iters": si, "sum_row_iters": sr,
                     "lane_efficiency": rec["n_edges"] / (16.0 * si), "rounding_share": 1 - rec["n_edges"] / (4.0 * sr),
                     "divergence_share": 1 - sr / (4.0 * si),
                     "deg_mean": float(deg.float().mean()), "deg_std": float(deg.float().std()), "deg_max": int(deg.max())}
    if mode == "count":
        torch.cuda.synchronize()
        fwd(); sc = bwd(1); bwd(2, sc)                    # dispatch order per configuration: fwd, (zero / dst / reduce), src
        torch.cuda.synchronize()
    else:
        sc = bwd(1)
        rec["ms"] = {"fwd": t(fwd), "dst": t(lambda: bwd(1, sc)), "src": t(lambda: bwd(2, sc))}
    print(rec, flush=True)
    records.append(rec)
    del g, bits
json.dump(records, open(os.path.join(out_dir, f"{mode}.json"), "w"), indent=1)
