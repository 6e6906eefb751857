"""Fill / refresh the per-configuration numbers of DESIGN.md section 5 from bench JSON lines:
   python scratch/fill_design.py aad=<json> rn=<json> ef=<json>"""
import json, re, sys
s = open('DESIGN.md').read()
names = {'aad': 'AAD', 'rn': 'RN', 'ef': 'EF'}
rows = {'aad': '| aadensenet121 bs=128 |', 'rn': '| resnet152 bs=128 |', 'ef': '| efficientnet-b4@380 bs=64 |'}
for a in sys.argv[1:]:
    k, f = a.split('=')
    d = json.load(open(f))
    if k == 'dn':                       # the committed DenseNet121 line: last row of the stage table
        i = s.index('| + dense corrected gradient slices'); j = s.index('\n', i)
        cells = s[i:j].rstrip(' |').split(' | ')
        cells[-3:] = ['**%d**' % round(d['value']), '**%.2f**' % d['ms_per_step'], '**%.1f %%**' % (100 * d['config']['model_hbm_roofline_frac'])]
        s = s[:i] + ' | '.join(cells) + ' |' + s[j:]
        continue
    img, ms, fr = "%d" % round(d['value']), "%.1f" % d['ms_per_step'], "%.1f %%" % (100 * d['config']['model_hbm_roofline_frac'])
    i = s.index(rows[k]); j = s.index('\n', i)
    line = s[i:j]
    cells = line.split(' | ')
    cells[1] = re.sub(r'\*\*[^*]+\*\*', '**' + img + '**', cells[1], count=1)
    cells[2], cells[3] = ms, fr
    s = s[:i] + ' | '.join(cells) + s[j:]
open('DESIGN.md', 'w').write(s)
