"""Order of vector-memory events in a kernel's ISA: L = global load, S = global store, Wn = s_waitcnt vmcnt(n), m = MFMA, |B| = barrier,
br = branch.  A `S ... Wk` with small k inside a loop means a wait for a store's acknowledgement (vmcnt counts loads and stores in order).
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast --cuda-device-only -S file.hip -o /tmp/f.s
  python scratch/isa_events.py /tmp/f.s <substring of the mangled kernel name>"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_Z\w*%s\w*):[ \t]*(?:;.*)?\n' % re.escape(pat), s, re.M):
    i = m.end(); j = s.index('s_endpgm', i)
    ev = []
    for l in s[i:j].split('\n'):
        t = l.strip()
        if t.startswith('global_load') or t.startswith('buffer_load'): ev.append('L')
        elif t.startswith('global_store') or t.startswith('buffer_store'): ev.append('S')
        elif t.startswith('global_atomic'): ev.append('A')
        elif 'vmcnt' in t: ev.append('W' + re.search(r'vmcnt\((\d+)\)', t).group(1))
        elif t.startswith('s_barrier'): ev.append('|B|')
        elif t.startswith('v_mfma'): ev.append('m')
        elif t.startswith('s_cbranch'): ev.append('br')
    out, prev, cnt = [], None, 0
    for e in ev + [None]:
        if e == prev: cnt += 1
        else:
            if prev: out.append(prev + (str(cnt) if cnt > 1 else ''))
            prev, cnt = e, 1
    print(m.group(1)[:90]); print('  ' + ' '.join(out))
