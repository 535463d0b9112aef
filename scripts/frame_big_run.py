#!/usr/bin/env python3
"""A few launches of the HBM-workspace frame path (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import frames
bays, stories, B = (int(a) for a in sys.argv[1:4])
topo = frames.grid_frame(bays, stories)
I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda")
sol = frames.frame_solve(topo, I)
for _ in range(10):
    frames.frame_solve(topo, I, out=sol)
torch.cuda.synchronize()
