#!/usr/bin/env python3
"""Which replays of the generator's cached 25-epoch graph stall (one replay of ~2 ms taking 25-55 ms), by shard size and poll interval:
each replay timed with a device sync (profiles/r03_notes.md section 5)."""
import time, torch, sys, os
sys.path.insert(0, ".")
from openpystruct_amd import sizing
cfg = sizing.SizingConfig()
orig_replay = torch.cuda.CUDAGraph.replay
for n, pe in ((50000, 25), (50000, 50), (50000, 10), (10000, 25), (200000, 25)):
    sizing._EPOCH_GRAPHS.clear()
    sizing.generate_dataset(n, cfg, "cuda", poll_every=pe)
    rt = []
    def rp(self):
        t0 = time.perf_counter(); orig_replay(self); torch.cuda.synchronize(); rt.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.CUDAGraph.replay = rp
    for i in range(8):
        sizing.generate_dataset(n, cfg, "cuda", poll_every=pe)
    torch.cuda.CUDAGraph.replay = orig_replay
    med = sorted(rt)[len(rt) // 2]
    stalls = [(i, round(t, 1)) for i, t in enumerate(rt) if t > 15 + 2 * med]
    print(n, pe, "replays", len(rt), "stalls at", stalls)
