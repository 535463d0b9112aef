#!/usr/bin/env python3
"""Throughput of the batched frame solve (BASELINE config 5 flavour): B frames of a bays x stories grid."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import frames
for bays, stories, B in [(10, 10, 4096), (5, 5, 8192), (3, 3, 16384)]:
    topo = frames.grid_frame(bays, stories)
    I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda") * (1 + 0.1 * torch.rand(B, topo.Ne, dtype=torch.float64, device="cuda"))
    sol = frames.frame_solve(topo, I)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        frames.frame_solve(topo, I, out=sol)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = topo.n_eq * topo.kd * topo.kd  # ~ n kd^2 (LDL^T, multiply-adds counted as 2)
    print(json.dumps({"frame": f"{bays}x{stories}", "elements": topo.Ne, "n_eq": topo.n_eq, "half_bandwidth": topo.kd, "B": B,
                      "ms_per_launch": ms, "frame_solves_per_s": B / ms * 1e3, "factor_GFLOPs": B * flops / ms / 1e6,
                      "lds_bytes_per_frame": topo.lds_bytes()}))
for bays, stories, B in [(15, 16, 1024)]:
    topo = frames.grid_frame(bays, stories)
    I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda")
    sol = frames.frame_solve(topo, I)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        frames.frame_solve(topo, I, out=sol)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"frame": f"{bays}x{stories}", "elements": topo.Ne, "n_eq": topo.n_eq, "half_bandwidth": topo.kd, "B": B,
                      "ms_per_launch": ms, "frame_solves_per_s": B / ms * 1e3, "band": "HBM workspace + LDS window"}))
