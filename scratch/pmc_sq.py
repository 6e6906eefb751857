"""Per-kernel means of the SQ counters of one rocprofv3 --pmc pass: python scratch/pmc_sq.py <counter_collection.csv>"""
import csv, collections, sys
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44] + ' g' + r['Grid_Size']
    d[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[(k, r['Counter_Name'])] += 1
names = sorted({c for v in d.values() for c in v})
print('%-56s' % 'kernel', ' '.join('%14s' % c.replace('SQ_', '')[:14] for c in names))
for k, v in d.items():
    if 'ring' in k or 'strip' in k or 'pw_' in k:
        print('%-56s' % k, ' '.join('%14.3g' % (v[c] / max(n[(k, c)], 1)) for c in names))
