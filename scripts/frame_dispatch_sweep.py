#!/usr/bin/env python3
"""Is the batch threshold between the workgroup-per-frame and the wave-per-frame kernels where the two meet?  us per launch, forced
wave / forced workgroup / the library's choice, over frame sizes and batches."""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from openpystruct_amd import _cabi, frames
    _cabi.set_option("frame_latency_batch", {"wave": 0, "workgroup": 1000000, "auto": -1}[sys.argv[1]])
    for (b, s) in ((1, 1), (2, 2), (3, 3), (5, 5), (10, 2), (7, 5), (8, 8), (10, 10), (15, 16)):
        topo = frames.grid_frame(b, s)
        for B in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
            I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda")
            sol = frames.frame_solve(topo, I); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                frames.frame_solve(topo, I, out=sol)
            torch.cuda.synchronize()
            print(f"{sys.argv[1]} {b}x{s} {topo.n_eq} {topo.kd} {B} {1e6 * (time.perf_counter() - t0) / 20:.1f}", flush=True)
    sys.exit(0)
rows = {}
for mode in ("wave", "workgroup", "auto"):      # (library option frame_latency_batch: 0 / 10^6 / the library's own model)
    out = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True).stdout
    for l in out.splitlines():
        p = l.split()
        if len(p) == 6 and p[0] == mode:
            rows.setdefault((p[1], int(p[2]), int(p[3]), int(p[4])), {})[mode] = float(p[5])
print("frame n_eq kd B   wave  workgroup  auto   auto/best")
for k in sorted(rows, key=lambda k: (k[1], k[3])):
    r = rows[k]
    if len(r) == 3:
        print(k[0], k[1], k[2], k[3], r["wave"], r["workgroup"], r["auto"], round(r["auto"] / min(r["wave"], r["workgroup"]), 2))
