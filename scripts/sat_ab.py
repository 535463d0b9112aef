"""A/B of tilings x store policy at one batch size, HIP-event timed eager launches (large batches: launch gaps are noise).
    python scripts/sat_ab.py 1048576 "8 520 528 16" """
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, openpystruct_amd as oa
B = int(sys.argv[1]); tilings = [int(t) for t in sys.argv[2].split()]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda')
inp = bench.synth_inputs(B, 0, dev, 'trajectory')
import time
warm = oa.beam_solve(**inp)
t0 = time.time()
while time.time() - t0 < 0.3:      # clocks, TLBs: the first configuration of a process must not pay for them
    for _ in range(5): oa.beam_solve(**inp, out=warm)
    torch.cuda.synchronize()
del warm
for til in tilings:
    for so in (False, True):
        out = oa.beam_solve(**inp, tiling=til, stream_out=so)
        for _ in range(3): oa.beam_solve(**inp, tiling=til, out=out, stream_out=so)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): oa.beam_solve(**inp, tiling=til, out=out, stream_out=so)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"B {B} tiling {til} stream_out {int(so)} {oa.kernel_name(B, 100, til)}: {us:.1f} us, frac {4925.0 * B / us / 1e3 / 8000:.3f}", flush=True)
        del out
