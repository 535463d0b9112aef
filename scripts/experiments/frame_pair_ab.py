"""A/B of the two-columns-per-step elimination (OPS_AMD_FRAME_PAIR=1) against the default one-column steps: bit equality of every output and
ms per launch at the bench's batches.  The switch is read per launch, so one process measures both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openpystruct_amd import frames

for (bays, stories, B) in [(15, 16, 12288), (10, 10, 16384), (5, 5, 32768), (3, 3, 65536), (10, 2, 65536), (7, 9, 8192)]:
    topo = frames.grid_frame(bays, stories)
    g = torch.Generator(device="cuda").manual_seed(1)
    I = torch.exp(torch.empty((B, topo.Ne), dtype=torch.float64, device="cuda").uniform_(np.log(1e-4), np.log(5e-3), generator=g))
    res = {}
    for mode in ("0", "1"):
        os.environ["OPS_AMD_FRAME_PAIR"] = mode
        sol = frames.frame_solve(topo, I)
        torch.cuda.synchronize()
        for _ in range(3):
            frames.frame_solve(topo, I, out=sol)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            frames.frame_solve(topo, I, out=sol)
        e1.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / 10, sol.disp.clone(), sol.forces.clone(), int(sol.status.abs().sum()))
    same = torch.equal(res["0"][1], res["1"][1]) and torch.equal(res["0"][2], res["1"][2])
    rel = float((res["0"][1] - res["1"][1]).abs().max() / res["0"][1].abs().max())
    print(f"{bays}x{stories} kd {topo.kd} B {B}: one column {res['0'][0]:.3f} ms, pairs {res['1'][0]:.3f} ms, bit-equal {same} (max rel diff {rel:.1e}), status {res['0'][3]} {res['1'][3]}", flush=True)
