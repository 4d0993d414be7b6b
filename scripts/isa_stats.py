"""Instruction histogram of one kernel in a gfx950 .s file: python scripts/isa_stats.py file.s kernel_substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
# function bodies start at "<name>:" and end at s_endpgm
for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ops = [l.split()[0] for l in body.splitlines()
           if l.strip() and not l.strip().startswith((';', '.', '//')) and not l.strip().endswith(':')]
    c = collections.Counter(ops)
    print(name, 'total', sum(c.values()),
          'VALU', sum(n for k, n in c.items() if k.startswith('v_')),
          'SALU', sum(n for k, n in c.items() if k.startswith('s_')),
          'VMEM', sum(n for k, n in c.items() if k.startswith(('global', 'flat', 'buffer', 'scratch'))),
          'LDS', sum(n for k, n in c.items() if k.startswith('ds_')))
    print('  ' + ', '.join(f'{k}:{n}' for k, n in c.most_common(36)))
    break
