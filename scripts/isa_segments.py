"""Per-segment instruction counts of one kernel in a hipcc -S listing; segments are delimited by the
compiler fences (empty asm statements) of the source.

    python scripts/isa_segments.py listing.s 'beam_fat_kernelILi6ELi17ELi1E'
"""
import re
import sys

txt = open(sys.argv[1]).read()
m = re.search(r'^(_Z\w*' + re.escape(sys.argv[2]) + r'\w*):[^\n]*\n(.*?)\n\.Lfunc_end', txt, re.S | re.M)
seg, cur = [], []
for l in m.group(2).split('\n'):
    t = l.strip()
    if t.startswith(';;#ASMSTART'):
        seg.append(cur)
        cur = []
        continue
    t = t.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':'):
        continue
    cur.append(t.split()[0])
seg.append(cur)
tot = dict(f64=0, vo=0, lds=0, salu=0, vmem=0)
for i, s in enumerate(seg):
    if not s:
        continue
    c = dict(f64=sum(1 for x in s if x.startswith('v_') and 'f64' in x),
             vo=sum(1 for x in s if x.startswith('v_') and 'f64' not in x),
             lds=sum(1 for x in s if x.startswith('ds_')),
             salu=sum(1 for x in s if x.startswith('s_')),
             vmem=sum(1 for x in s if x.startswith(('buffer', 'global', 'scratch', 'flat'))))
    print(i, len(s), c)
