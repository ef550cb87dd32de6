"""Joins gpurun_out/valu_budget/{time,count}.json with the rocprofv3 counter CSV(s) under gpurun_out/valu_budget/pmc and
prints the per-wave (row prologue + epilogue) and per-batch-iteration VALU coefficients of the three aggregation kernels."""
import csv, glob, json, os, re, sys, collections
import numpy as np
base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/valu_budget"
cnt = json.load(open(os.path.join(base, "count.json")))
tim = {json.dumps({k: r[k] for k in ("k", "gelu", "window", "drop")}): r for r in json.load(open(os.path.join(base, "time.json")))} \
    if os.path.exists(os.path.join(base, "time.json")) else {}
rows = collections.defaultdict(list)            # kernel class -> [(dispatch id, {counter: value})]
by_disp = collections.defaultdict(dict)
for f in glob.glob(base + "/pmc*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        m = re.search(r"(gatv2_fwd_kernel|gatv2_bwd_dst_kernel|gatv2_bwd_src_kernel)", kn)
        if not m: continue
        by_disp[(m.group(1), int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
for (kname, disp), c in sorted(by_disp.items(), key=lambda x: x[0][1]):
    rows[kname].append(c)
names = {"fwd": "gatv2_fwd_kernel", "dst": "gatv2_bwd_dst_kernel", "src": "gatv2_bwd_src_kernel"}
# the script launches fwd once before the counted sequence (to fill pre / lse) and, in count mode, fwd / dst / src once more
per_cfg = {"fwd": 2, "dst": 1, "src": 1}
print(f"{'config':34s} {'kernel':4s} {'VALU (M)':>9s} {'waves':>8s} {'sum_iters':>10s} {'lane_eff':>8s} {'round':>6s} {'diverg':>6s} {'ms':>7s} {'VALU/edge-slot4':>8s}")
fit = collections.defaultdict(list)
for i, r in enumerate(cnt):
    key = json.dumps({k: r[k] for k in ("k", "gelu", "window", "drop")})
    for short, kn in names.items():
        lst = rows.get(kn, [])
        j = i * per_cfg[short] + (per_cfg[short] - 1)
        if j >= len(lst): continue
        valu = lst[j].get("SQ_INSTS_VALU")
        view = r["dst" if short != "src" else "src"]
        ms = tim.get(key, {}).get("ms", {}).get(short)
        print(f"{key[1:-1].replace(chr(34), ''):34s} {short:4s} {valu/1e6:9.1f} {view['waves']:8d} {view['sum_wave_iters']:10d} "
              f"{view['lane_efficiency']:8.3f} {view['rounding_share']:6.3f} {view['divergence_share']:6.3f} "
              f"{(ms if ms is not None else float('nan')):7.3f} {valu / (r['n_edges'] / 4):8.1f}")
        if r["gelu"] == 1 and r["window"] == 0 and r["drop"] > 0:
            fit[short].append((view["waves"], view["sum_wave_iters"], valu, r["k"], r["n_edges"]))
print()
for short, pts in fit.items():
    if len(pts) < 2: continue
    A = np.array([[p[0], p[1]] for p in pts], float); y = np.array([p[2] for p in pts], float)
    (P, B), res, *_ = np.linalg.lstsq(A, y, rcond=None)
    print(f"{short}: VALU = {P:7.1f} per wave (4 rows: prologue + epilogue) + {B:7.1f} per 4-edge batch iteration of a wave (16 edge slots)"
          f"   [fit over k = {[p[3] for p in pts]}, max rel. residual {np.max(np.abs(A @ [P, B] - y) / y):.3f}]")
    for p in pts:
        if p[3] == 15:
            e = p[4]
            print(f"    k=15: per edge  {p[2] * 4 / e:6.1f} wave-instr x 4 edges/instr = total;  row part {P * p[0] * 4 / e:5.1f}, "
                  f"edge loop {B * p[1] * 4 / e:5.1f} of which useful (no padding) {B * (e / 16) * 4 / e:5.1f}, "
                  f"padding {(B * p[1] - B * e / 16) * 4 / e:5.1f}")
