#!/usr/bin/env python3
"""Where a stalled graph replay spends its time: from a rocprofv3 kernel / memory-copy / HIP-API trace directory, list every device-side
interval > 5 ms in which no kernel ran, every kernel > 5 ms, every copy > 1 ms, and the HIP calls that overlap them.
usage: stall_trace_summary.py <dir with *_kernel_trace.csv ...>"""
import csv, glob, sys, os
d = sys.argv[1]
def load(pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    rows = []
    for f in fs:
        rows += list(csv.DictReader(open(f)))
    return rows
k = load("*_kernel_trace.csv")
for r in k:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
k.sort(key=lambda r: r["s"])
print("kernels", len(k))
if not k:
    sys.exit(0)
t0 = k[0]["s"]
import collections
dur = collections.defaultdict(list)
for r in k:
    dur[r["Kernel_Name"][:60]].append((r["e"] - r["s"]) / 1e3)
for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v.sort()
    print("  %-60s n %5d  median %8.1f us  p99 %8.1f  max %9.1f" % (name, len(v), v[len(v) // 2], v[int(len(v) * 0.99)], v[-1]))
long_k = [r for r in k if r["e"] - r["s"] > 5e6]
print("kernels longer than 5 ms:", len(long_k))
for r in long_k[:20]:
    print("  %.3f ms at +%.3f ms  %s" % ((r["e"] - r["s"]) / 1e6, (r["s"] - t0) / 1e6, r["Kernel_Name"][:80]))
gaps = []
end = k[0]["e"]
for a in k[1:]:
    if a["s"] - end > 5e6:
        gaps.append((end, a["s"]))
    end = max(end, a["e"])
print("device idle gaps longer than 5 ms:", len(gaps))
m = load("*_memory_copy_trace.csv")
for r in m:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
h = load("*_hip_api_trace.csv")
for r in h:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
for g0, g1 in gaps[:40]:
    # what ran last before the gap and first after it; copies and HIP calls overlapping it
    before = [r for r in k if r["e"] <= g0][-1]
    after = [r for r in k if r["s"] >= g1][0]
    print("  gap %.2f ms at +%.3f ms: after %s -> before %s" % ((g1 - g0) / 1e6, (g0 - t0) / 1e6, before["Kernel_Name"][:40], after["Kernel_Name"][:40]))
    for r in m:
        if r["e"] > g0 and r["s"] < g1:
            print("      copy %s %s bytes %.3f ms (+%.3f .. +%.3f ms into the gap)" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")), (r["e"] - r["s"]) / 1e6, (r["s"] - g0) / 1e6, (r["e"] - g0) / 1e6))
    for r in h:
        if r["e"] > g0 and r["s"] < g1 and r["e"] - r["s"] > 2e5:
            print("      hip %s %.3f ms (+%.3f .. +%.3f ms into the gap)" % (r.get("Function", "?"), (r["e"] - r["s"]) / 1e6, (r["s"] - g0) / 1e6, (r["e"] - g0) / 1e6))
long_m = [r for r in m if r["e"] - r["s"] > 1e6]
print("copies longer than 1 ms:", len(long_m), "of", len(m))
for r in long_m[:20]:
    print("  %.3f ms at +%.3f ms %s %s" % ((r["e"] - r["s"]) / 1e6, (r["s"] - t0) / 1e6, r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))))
long_h = sorted((r for r in h if r["e"] - r["s"] > 5e6), key=lambda r: r["s"])
print("HIP calls longer than 5 ms:", len(long_h), "of", len(h))
for r in long_h[:40]:
    print("  %.3f ms at +%.3f ms %s" % ((r["e"] - r["s"]) / 1e6, (r["s"] - t0) / 1e6, r.get("Function", "?")))
# every long hipGraphLaunch: the HIP calls of all threads from 3 ms before its start to its end
print("---- HIP calls around every hipGraphLaunch longer than 5 ms")
h.sort(key=lambda r: r["s"])
for L in [r for r in long_h if r.get("Function") == "hipGraphLaunch"][:6]:
    print("hipGraphLaunch %.3f ms, thread %s, start +%.3f ms" % ((L["e"] - L["s"]) / 1e6, L.get("Thread_Id", "?"), (L["s"] - t0) / 1e6))
    for r in h:
        if r["e"] > L["s"] - 3e6 and r["s"] < L["e"] + 1e6 and r is not L:
            print("    %-34s thread %s  start %+9.3f ms  dur %8.3f ms" % (r.get("Function", "?"), r.get("Thread_Id", "?"), (r["s"] - L["s"]) / 1e6, (r["e"] - r["s"]) / 1e6))
    ks = [r for r in k if r["e"] > L["s"] - 3e6 and r["s"] < L["e"] + 1e6]
    for r in ks[:3] + ks[-3:]:
        print("    kernel %-40s start %+9.3f ms  dur %8.3f ms" % (r["Kernel_Name"][:40], (r["s"] - L["s"]) / 1e6, (r["e"] - r["s"]) / 1e6))
