#!/usr/bin/env python3
"""Where the four-waves-per-frame kernel (small batches) and the wave-per-frame / packed kernels (large batches) meet: time per call over the batch size,
both families forced through the library option frame_latency_batch (huge: small-batch family for every B; 0: tuned kernels for every B).

    python scripts/frame_coop_sweep.py 15x16 12x12 10x10
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import _cabi, frames  # noqa: E402


def timed(topo, I, sol):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(5):
            frames.frame_solve(topo, I, out=sol)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    return best


def main():
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(15, 16)]
    lib = _cabi.load()
    for bays, stories in shapes:
        topo = frames.grid_frame(bays, stories)
        for B in [int(v) for v in os.environ.get("BATCHES", "64,256,512,1024,2048,4096,8192").split(",")]:
            I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda") * (1 + 0.1 * torch.rand(B, topo.Ne, dtype=torch.float64, device="cuda"))
            rec = {"frame": f"{bays}x{stories}", "half_bandwidth": topo.kd, "B": B}
            for name, lat, coop in (("small_coop", 1 << 30, 2), ("small_workgroup", 1 << 30, 0), ("tuned", 0, 1)):
                _cabi.set_option("frame_latency_batch", lat)
                _cabi.set_option("frame_coop", coop)
                topo.__dict__.pop("_ws", None)
                try:
                    sol = frames.frame_solve(topo, I)
                    torch.cuda.synchronize()
                    rec[name + "_family"] = int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd)) >> 24
                    rec[name + "_us"] = round(timed(topo, I, sol), 1)
                except Exception as e:      # (a family that cannot serve the shape)
                    rec[name + "_us"] = repr(e)[:60]
            _cabi.set_option("frame_latency_batch", -1)
            _cabi.set_option("frame_coop", 1)
            rec["default_family"] = int(lib.ops_frame_plan_signature(B, topo.n_eq, topo.kd)) >> 24
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
