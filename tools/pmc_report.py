import csv, glob, sys, collections, re
base = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'segger' not in k: continue
        m = re.search(r'segger::(?:\(anonymous namespace\)::)?(\w+)', k); short = m.group(1) if m else k[:60]
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('==', k)
    for c, v in sorted(d.items()):
        print(f'   {c:24s} n={len(v):3d} mean={sum(v)/len(v):16.1f}')
