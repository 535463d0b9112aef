"""Are the row-staged 16-lane kernel's results bit-identical to beam_solve.hip's 16-lane kernel? (same beam_math.hpp statements)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench, openpystruct_amd as oa
inp = bench.synth_inputs(10007, 0, torch.device('cuda'), 'trajectory')
a = oa.beam_solve(**inp, tiling=16)
b = oa.beam_solve(**inp, tiling=16 | 0x200)
torch.cuda.synchronize()
for name, p, q in zip(("v", "theta", "V", "M"), a, b):
    d = (p != q).sum().item()
    print(name, "differing entries", d, "max rel", float(((p - q).abs() / p.abs().clamp_min(1e-300)).max()))
