import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from openpystruct_amd import frames
for (b, s) in ((10, 10), (5, 5), (3, 3), (15, 16)):
    topo = frames.grid_frame(b, s)
    for B in (1, 4, 16, 64, 256, 1024):
        I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda")
        sol = frames.frame_solve(topo, I); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            frames.frame_solve(topo, I, out=sol); torch.cuda.synchronize()
        print(os.environ.get("OPS_AMD_FRAME_LEGACY", "0"), f"{b}x{s} kd {topo.kd} B {B}: {1e6 * (time.perf_counter() - t0) / 50:.1f} us")
