"""profiles/hbm_traffic.json from the PMC passes of tools/pmc.sh:  python tools/pmc_to_traffic.py gpurun_out/<dir> <out.json>
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters; FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950, which tallies 128-B requests at 64 B)."""
import collections, csv, glob, json, re, sys
base, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'segger::(?:\(anonymous namespace\)::)?(gatv2_(?:fwd|bwd_dst|bwd_src))_kernel', r['Kernel_Name'])
        if m and r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
res = {"_comment": "rocprofv3 --pmc passes (separate runs, kernel-trace only: DROP=0.2 BITS=1 tools/pmc.sh) over "
                   "tools/bench_gat.py: C2 tx-neighbors-tx layer, bf16, H=2 C=64, attention dropout 0.2 as bit planes "
                   "(the training configuration); means over the launches of the run. FETCH_SIZE / WRITE_SIZE are KiB; "
                   "FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B).",
       "workload": {"n_tx": 1000000, "k": 15, "dtype": "bf16", "dropout": 0.2}}
for k, d in sorted(agg.items()):
    fs, ws = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']), sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
    res[f"{k}_FETCH_SIZE_KiB"], res[f"{k}_WRITE_SIZE_KiB"] = round(fs, 1), round(ws, 1)
    res[f"{k}_bytes_per_launch"] = int((2 * fs + ws) * 1024)
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
