"""One step of the rocprofv3 kernel trace as a table (start offset us, duration us, gap to the previous end on any stream, queue, name):
python scratch/trace_step.py <kernel_trace.csv> [step] > out.txt"""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: re.sub(r'\(.*', '', r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''))[:48]
idx = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
pairs = [(x, y) for x, y in zip(idx, idx[1:]) if y - x > 200]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(pairs) // 2
a, b = pairs[which]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
last_end = t0
print("# %d steps; step %d: %d kernels, %.3f ms" % (len(pairs), which, len(step), (int(step[-1]['End_Timestamp']) - t0) / 1e6))
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%9.1f %8.1f %7.1f q%-3s %s  g=%s wg=%s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3, r.get('Queue_Id', '?')[-3:], name(r),
                                                   r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))))
    last_end = max(last_end, e)
