"""Scan the gfx950 assembly of every kernel for global loads that are waited on (vmcnt(0)) within a few instructions, the
signature of a load the compiler sank under a bounds branch: python scratch/scan_sunk_loads.py [file.hip ...]"""
import subprocess, sys, re, glob, os
src = sys.argv[1:] or sorted(glob.glob('chexpert_amd/csrc/*.hip'))
for f in src:
    s = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-Iinclude', '-S', '--cuda-device-only',
                        f, '-o', '/tmp/scan.s'], capture_output=True, text=True)
    lines = open('/tmp/scan.s').read().split('\n')
    kern, hits = None, {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            kern = m.group(1)
        if kern and re.match(r'\s+(global_load|buffer_load)', l):
            n = 0
            for j in range(i + 1, min(i + 12, len(lines))):
                t = lines[j].strip()
                if not t or t.startswith(';') or t.startswith('.'):
                    continue
                n += 1
                if re.match(r'(global_load|buffer_load)', t):
                    break
                if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
                    hits.setdefault(kern, []).append(i + 1)
                    break
                if n >= 4:
                    break
    for k, v in hits.items():
        d = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()[:90]
        print("%-28s %-92s %3d  lines %s" % (os.path.basename(f), d, len(v), v[:8]))
