#!/usr/bin/env python3
"""The 10^4-beam launch from a cold chip: us per launch of consecutive blocks of 100 graph-captured launches (one buffer set, then 16
rotating sets): does the contract batch see the warm-up transient the 2^20-beam launch shows (scripts/sat_series.py)?"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, openpystruct_amd as oa
dev = torch.device("cuda")
B = 10000
for nsets in (1, 16):
    base = bench.synth_inputs(B, 0, dev, "trajectory")
    sets = [base] + [dict(base, I=base["I"].roll(k, 0).contiguous(), Fy=base["Fy"].roll(k, 0).contiguous()) for k in range(1, nsets)]
    outs = [oa.beam_solve(**st) for st in sets]
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(96):
                oa.beam_solve(**sets[i % nsets], out=outs[i % nsets])
        import time; time.sleep(1.0)          # let the chip idle
        nb = 80
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(nb + 1)]
        ev[0].record(s)
        for k in range(nb):
            g.replay(); ev[k + 1].record(s)
        s.synchronize()
    d = np.array([ev[k].elapsed_time(ev[k + 1]) * 1e3 / 96 for k in range(nb)])
    print(f"{nsets} buffer set(s): us per launch over consecutive blocks of 96 launches (block = {d[-1] * 96 / 1e3:.2f} ms):")
    print(" ".join("%.2f" % v for v in d))
    print(f"  blocks 1..5 mean {d[:5].mean():.2f}   6..20 mean {d[5:20].mean():.2f}   21..80 mean {d[20:].mean():.2f} sigma {d[20:].std():.2f}")
