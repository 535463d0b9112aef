#!/usr/bin/env python3
"""The 2^20-beam launch, one HIP-event pair per launch: the duration SERIES of 120 back-to-back launches (r03: sigma 119 us on 1 044, min 861,
max 1 435 over 51 profiled calls -- a +-14 % spread nobody explained), the chip's clocks and power before / during / after, and the same
with a 2 ms pause between launches.  usage: sat_series.py [launches]"""
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, openpystruct_amd as oa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda")
B = 1 << 20


def smi(tag):
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power (W)", "Temperature (Sensor junction)", "Temperature (Sensor memory)"))]
        print(f"[{tag}] " + " | ".join(keep), flush=True)
    except Exception as e:
        print(f"[{tag}] rocm-smi unavailable: {e!r}")


inp = bench.synth_inputs(B, 0, dev, "trajectory")
out = oa.beam_solve(**inp)
torch.cuda.synchronize()
smi("idle, before")
for pause in (0.0, 0.002):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); oa.beam_solve(**inp, out=out); b.record()
        if pause:
            torch.cuda.synchronize(); time.sleep(pause)
    torch.cuda.synchronize()
    d = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
    print(f"pause {pause * 1e3:.0f} ms between launches: us per launch, launches 1..{n}:")
    print(" ".join("%.0f" % v for v in d))
    for lo, hi in ((0, 10), (10, 30), (30, n)):
        s = d[lo:hi]
        print(f"  launches {lo + 1}..{hi}: mean {s.mean():.1f} sigma {s.std():.1f} ({100 * s.std() / s.mean():.1f} %) min {s.min():.0f} max {s.max():.0f}  frac of 8 TB/s {4925.0 * B / s.mean() / 1e3 / 8000:.3f}")
    if pause == 0.0:
        smi("right after 120 back-to-back launches")
print("kernel", oa.kernel_name(B, 100))
