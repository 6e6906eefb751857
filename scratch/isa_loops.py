# instruction mix of every loop of one kernel in a -S listing: python scratch/isa_loops.py file.s <substring of the kernel symbol>
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if sys.argv[2] in l and re.match(r'^_Z\w+:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
def kind(x):
    return ('mfma' if 'mfma' in x else 'exp' if 'v_exp' in x else 'ds' if x.startswith('ds_') else 'valu' if x.startswith('v_') else 'wait' if 'waitcnt' in x else 'bar' if 'barrier' in x
            else 'salu' if x.startswith('s_') else 'vmem' if x.startswith(('global', 'buffer', 'scratch')) else 'other')
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i]
        ins = [x.strip().split()[0] for x in seg if x.strip() and not x.strip().startswith(('.', ';')) and not x.strip().endswith(':')]
        c = Counter(kind(x) for x in ins)
        print(m.group(1), 'lines', labels[m.group(1)], i, 'instr', len(ins), dict(c))
        if len(sys.argv) > 3 and sys.argv[3] == m.group(1):
            v = Counter(x for x in ins if kind(x) == 'valu')
            print(v.most_common(40))
