#!/usr/bin/env python3
"""Ordered kernel list (start, duration, name) of one captured training step from a rocprofv3 --kernel-trace CSV: usage: trace_step_sequence.py <kernel_trace.csv>"""
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'FusedOptimizer' in r['Kernel_Name'] or 'flat_adam_kernel' in r['Kernel_Name'] or 'flat_adam_tiles_kernel' in r['Kernel_Name']]
a,b=idx[-40],idx[-39]
t0=int(rows[a]['Start_Timestamp'])
import re
def short(n):
    n=re.sub(r'at::native::','',n); n=re.sub(r'\(anonymous namespace\)::','',n); n=n.replace('void ','')
    return n[:95]
for r in rows[a+1:b+1]:
    print("%8.1f %6.1f  %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, short(r['Kernel_Name'])))
print("kernels", b-a)
