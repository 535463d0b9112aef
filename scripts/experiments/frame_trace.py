#!/usr/bin/env python3
"""Per-step cycle stamps of the frame factorisation (diagnostic build: OPS_AMD_EXTRA_HIPCC_FLAGS=-DOPS_AMD_FRAME_TRACE,
library selected with OPS_AMD_LIB).  Stamps of workgroup 0: wave 0 and the look-ahead wave, first 40 block steps."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(2 * 40 * 4, dtype=torch.int64, device="cuda")
os.environ["OPS_AMD_FRAME_TRACE_PTR"] = str(buf.data_ptr())
from openpystruct_amd import frames
bays, stories, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
topo = frames.grid_frame(bays, stories)
I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda")
sol = frames.frame_solve(topo, I)
for _ in range(3):
    frames.frame_solve(topo, I, out=sol)
torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
frames.frame_solve(topo, I, out=sol); torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(2, 40, 4)
for w, name in enumerate(["wave0", "lookahead"]):
    a = t[w]; ok = a[:, 0] > 0
    a = a[ok]
    print(name, "steps", ok.sum())
    print("  load pivot   med", np.median(a[:, 1] - a[:, 0]))
    print("  own work     med", np.median(a[:, 2] - a[:, 1]))
    print("  barrier wait med", np.median(a[:, 3] - a[:, 2]))
    print("  step total   med", np.median(a[1:, 0] - a[:-1, 0]), "clock64 ticks")
