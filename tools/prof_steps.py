import csv,glob,re,collections,sys
d0=sys.argv[1]; steps=int(sys.argv[2]); top=int(sys.argv[3]) if len(sys.argv)>3 else 30
f=glob.glob(d0+'/**/*_kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
cut=len(rows); run=0
for i,n in enumerate(names):
    if 'gatv2_fwd_kernel' in n:
        run+=1
        if run==5: cut=i-4; break
    else: run=0
rows=rows[:cut]
d=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    n=r['Kernel_Name']; n=re.sub(r'at::native::','',n); n=re.sub(r'\(anonymous namespace\)::','',n)
    d[n][0]+=1; d[n][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
tot=sum(v[1] for v in d.values())
print('kernels in steps (warm-up steps included: the first one also sorts the edge stores): busy %.2f ms/step, %d launches/step'%(tot/steps, len(rows)/steps))
for n,(c,t) in sorted(d.items(), key=lambda kv:-kv[1][1])[:top]:
    print(f"{t/steps:7.3f} ms/step calls/step={c/steps:6.1f} avg={t/c*1e3:8.1f}us {n[:105]}")
