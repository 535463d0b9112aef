"""Instruction histogram of one kernel in a hipcc -S listing (static counts, both RZ branches included).

    python scripts/isa_hist.py listing.s 'beam_solve_kernelILi16ELi7ELb1ELb1E'
"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    pat = sys.argv[2]
    m = re.search(r'^(_Z\w*' + re.escape(pat) + r'\w*):[^\n]*\n(.*?)\n\.Lfunc_end', txt, re.S | re.M)
    if not m:
        raise SystemExit("kernel not found")
    body = m.group(2)
    lines = [l.strip() for l in body.split('\n')]
    lines = [l.split(';')[0].strip() for l in lines]
    lines = [l for l in lines if l and not l.startswith(';') and not l.startswith('.') and not l.endswith(':')]
    ins = [l.split()[0] for l in lines]
    c = collections.Counter(ins)
    groups = collections.Counter()
    for k, v in c.items():
        if k.startswith('v_') and 'f64' in k:
            groups['valu_f64'] += v
        elif k.startswith('v_'):
            groups['valu_other'] += v
        elif k.startswith('s_'):
            groups['salu'] += v
        elif k.startswith('ds_'):
            groups['lds'] += v
        elif k.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
            groups['vmem'] += v
        else:
            groups['other'] += v
    print(m.group(1), len(ins), dict(groups))
    for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
        print(f"  {k:28s} {v}")
    for key in ('num_vgpr', 'private_seg_size'):
        mm = re.search(re.escape(m.group(1)) + r'\.' + key + r', (\d+)', txt)
        if mm:
            print(key, mm.group(1))


main()
