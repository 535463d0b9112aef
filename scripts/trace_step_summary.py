#!/usr/bin/env python3
"""Per-training-step kernel statistics from a rocprofv3 --kernel-trace CSV (steps are delimited by the optimiser kernel): usage: trace_step_summary.py <kernel_trace.csv>"""
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'FusedOptimizer' in r['Kernel_Name'] or 'flat_adam_kernel' in r['Kernel_Name'] or 'flat_adam_tiles_kernel' in r['Kernel_Name']]
print("adam calls", len(idx))
# steady-state steps: last 150 adam intervals
n_iv=min(150,max(1,len(idx)-20))
ivs=[(idx[k],idx[k+1]) for k in range(len(idx)-10-n_iv,len(idx)-10)]
cnt=[b-a for a,b in ivs]; wall=[int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']) for a,b in ivs]
busy=[sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[a:b]) for a,b in ivs]
import statistics as st
print("kernels/step median", st.median(cnt), "wall us median", st.median(wall)/1e3, "busy us median", st.median(busy)/1e3)
a,b=ivs[len(ivs)//3]
c=collections.Counter(); d=collections.Counter()
for r in rows[a:b]:
    c[r['Kernel_Name'][:70]]+=1; d[r['Kernel_Name'][:70]]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,v in d.most_common(25): print("%7.1f us %3d x %s"%(v/1e3,c[k],k))
