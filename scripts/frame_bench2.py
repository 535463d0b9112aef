#!/usr/bin/env python3
"""Throughput of the batched frame solve at several batch sizes (wave-per-frame kernel: needs >= 12 frames per CU in flight)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import frames
from openpystruct_amd import _cabi
_cabi.set_option("frame_latency_batch", int(os.environ.get("FRAME_BENCH_LATENCY_BATCH", "-1")))
cases = [(10, 10, 4096), (10, 10, 16384), (15, 16, 1024), (15, 16, 4096), (15, 16, 12288), (5, 5, 32768), (3, 3, 65536)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for bays, stories, B in cases:
    topo = frames.grid_frame(bays, stories)
    I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda") * (1 + 0.1 * torch.rand(B, topo.Ne, dtype=torch.float64, device="cuda"))
    sol = frames.frame_solve(topo, I)
    torch.cuda.synchronize()
    assert os.environ.get("FRAME_BENCH_NOCHECK") or int(sol.status.abs().sum()) == 0   # (phase-ablation builds give wrong results)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        frames.frame_solve(topo, I, out=sol)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    wsb = int(_cabi.load().ops_frame_workspace_bytes(4097, topo.n_eq, topo.kd)) - int(_cabi.load().ops_frame_workspace_bytes(4096, topo.n_eq, topo.kd))
    print(json.dumps({"frame": f"{bays}x{stories}", "elements": topo.Ne, "n_eq": topo.n_eq, "half_bandwidth": topo.kd, "B": B,
                      "ms_per_launch": round(ms, 4), "frame_solves_per_s": B / ms * 1e3, "workspace_bytes_per_frame": wsb,
                      "hbm_GBs_at_2x_workspace": 2 * wsb * B / ms / 1e6}))
