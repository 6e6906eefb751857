"""Gaps on the GPU timeline of one training step from a rocprofv3 --kernel-trace CSV: python scratch/timeline.py <kernel_trace.csv>"""
import csv, re, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: re.sub(r'\(.*', '', r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''))[:34]
# take the last full step: between the last two adam kernels
idx = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
pairs = [(x, y) for x, y in zip(idx, idx[1:]) if y - x > 200]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(pairs) // 2      # a step from the middle of the run (graph replays)
a, b = pairs[which]
print("%d steps found, analysing step %d" % (len(pairs), which))
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
print("step: %d kernels, %.3f ms wall" % (len(step), (t1 - t0) / 1e6))
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
print("sum of kernel durations %.3f ms" % (busy / 1e6))
# union busy
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
cur_s, cur_e, union = ev[0][0], ev[0][1], 0
gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("GPU busy (union) %.3f ms, idle %.3f ms in %d gaps" % (union / 1e6, (t1 - t0 - union) / 1e6, len(gaps)))
hist = collections.Counter()
for g, _ in gaps:
    hist[min(int(g / 1000), 20)] += 1
print("gap histogram (us: count):", sorted(hist.items()))
# which kernels precede the gaps (by total gap time)
ends = {int(r['End_Timestamp']): name(r) for r in step}
by = collections.defaultdict(lambda: [0, 0])
for g, at in gaps:
    k = ends.get(at, '?')
    by[k][0] += g; by[k][1] += 1
for k, (g, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  after %-36s %4d gaps %7.3f ms" % (k, n, g / 1e6))
