"""Per-wave phase timeline of the fat-wave beam kernel (library built with -DOPS_AMD_TRACE):
    OPS_AMD_LIB=$PWD/ab/lib_fat_trace.so python scripts/trace_fat.py 10000 6 [sets]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
B = int(sys.argv[1]); til = int(sys.argv[2]); nsets = int(sys.argv[3]) if len(sys.argv) > 3 else 1
bpw = 64 // (til & 0xff)
buf = torch.zeros(8 * (B // bpw + 4), dtype=torch.int64, device='cuda')
os.environ["OPS_AMD_TRACE_PTR"] = str(buf.data_ptr())
import bench, openpystruct_amd as oa
base = bench.synth_inputs(B, 0, torch.device('cuda'), 'trajectory')
sets = [base] + [dict(base, I=base["I"].roll(k, 0).contiguous(), Fy=base["Fy"].roll(k, 0).contiguous()) for k in range(1, nsets)]
outs = [oa.beam_solve(**s, tiling=til) for s in sets]
for r in range(3):
    for s, o in zip(sets, outs):
        oa.beam_solve(**s, tiling=til, out=o)
torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
oa.beam_solve(**sets[0], tiling=til, out=outs[0]); torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print("waves", len(t), "kernel", oa.kernel_name(B, 100, til), "sets", nsets)
def col(k): return (t[:, k] - t0) / 100.0
names = [("entry", 0), ("loads issued", 5), ("staged", 1), ("condensed", 6), ("interface done", 7), ("solved", 2), ("end", 3)]
prev = None
for nm, k in names:
    c = col(k)
    d = "" if prev is None else "  (+%.2f med since previous)" % np.median(c - prev)
    print("%-15s min %.2f p10 %.2f med %.2f p90 %.2f max %.2f us%s" % (nm, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max(), d))
    prev = c
hw = t[:, 4]; wid = hw & 15
import collections
print("wave_id hist", sorted(collections.Counter(wid.tolist()).items()))
for w in sorted(set(wid.tolist())):
    m = wid == w
    print("slot", w, "n", int(m.sum()), " ".join("%s %.2f" % (nm, np.median(col(k)[m])) for nm, k in names))
