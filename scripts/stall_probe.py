#!/usr/bin/env python3
"""Workload of scripts/stall_trace.sh: warm 50 000-case shards of the generator, every graph replay timed on the host with a device sync
behind it (a stalled replay shows as 25-55 ms instead of ~2).  Prints the host time of each stall relative to the first replay, so that the
kernel / copy / HIP-API trace of the same process can be searched at that offset.  usage: stall_probe.py [shards] [cases] [poll_every]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from openpystruct_amd import sizing
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
pe = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cfg = sizing.SizingConfig()
sizing.generate_dataset(n, cfg, "cuda", poll_every=pe)
orig = torch.cuda.CUDAGraph.replay
rt = []
def rp(self):
    t0 = time.perf_counter_ns(); orig(self); torch.cuda.synchronize(); rt.append((t0, time.perf_counter_ns()))
torch.cuda.CUDAGraph.replay = rp
tg = []
for i in range(shards):
    t0 = time.perf_counter(); sizing.generate_dataset(n, cfg, "cuda", poll_every=pe); torch.cuda.synchronize(); tg.append(time.perf_counter() - t0)
torch.cuda.CUDAGraph.replay = orig
d = [(b - a) / 1e6 for a, b in rt]
med = sorted(d)[len(d) // 2]
print("shards", shards, "cases", n, "poll_every", pe, "replays", len(d), "median replay ms %.3f" % med, "shard s", " ".join("%.4f" % t for t in tg))
for i, ((a, b), t) in enumerate(zip(rt, d)):
    if t > 10 + 2 * med:
        print("STALL replay %d (shard %d, replay %d of it): %.1f ms, host clock %d .. %d ns (CLOCK_MONOTONIC)" % (i, i // (len(d) // shards), i % (len(d) // shards), t, a, b))
import resource, threading
ru = resource.getrusage(resource.RUSAGE_SELF)
print("process CPU time: user %.2f s  sys %.2f s   threads now: %d" % (ru.ru_utime, ru.ru_stime, len(os.listdir("/proc/self/task"))))
