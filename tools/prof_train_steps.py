"""Per-step kernel breakdown of a training run traced with rocprofv3 --kernel-trace: the last N optimizer steps
(delimited by the fused-Adam kernel).   python tools/prof_train_steps.py <trace dir> <N> [top]"""
import collections, csv, glob, re, sys
d0 = sys.argv[1]; steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = glob.glob(d0 + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name'].lower()]
# a fused Adam step may be several launches: group launches closer than 20 kernels apart
ends = [adam[0]]
for i in adam[1:]:
    if i - ends[-1] < 20: ends[-1] = i
    else: ends.append(i)
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
rows = rows[lo:hi]
d = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = re.sub(r'at::native::|\(anonymous namespace\)::', '', r['Kernel_Name'])
    d[n][0] += 1; d[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
tot = sum(v[1] for v in d.values())
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6
print('last %d steps: busy %.3f ms/step, span %.3f ms/step, %.1f launches/step' % (steps, tot / steps, span / steps, len(rows) / steps))
for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t/steps:7.3f} ms/step calls/step={c/steps:6.1f} avg={t/c*1e3:8.1f}us {n[:110]}")
if len(sys.argv) > 4:                                   # the launch sequence of the last step
    seq = rows[-(len(rows) // steps):]
    t0 = int(seq[0]['Start_Timestamp'])
    with open(sys.argv[4], 'w') as fh:
        prev_end = t0
        for r in seq:
            n = re.sub(r'at::native::|\(anonymous namespace\)::', '', r['Kernel_Name'])
            s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            fh.write(f"{(s_ - t0) / 1e3:9.1f}us gap={(s_ - prev_end) / 1e3:6.1f} dur={(e_ - s_) / 1e3:7.1f} grid={r.get('Grid_Size_X', r.get('Grid_Size', '?'))} {n[:120]}\n")
            prev_end = e_
