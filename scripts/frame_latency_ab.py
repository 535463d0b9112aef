#!/usr/bin/env python3
"""Small batches of frames (the reference's own use: ONE frame per epoch, FR:178-183): time per call of the four-waves-per-frame kernel
(csrc/frame_coop.hpp, library option frame_coop = 2: for every small batch) against the r01 workgroup-per-frame kernels (frame_coop = 0), answers compared.

    python scripts/frame_latency_ab.py [bays x stories ...]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import _cabi, frames  # noqa: E402


def main():
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2, 2), (3, 3), (5, 5), (7, 7), (10, 10), (15, 16)]
    for bays, stories in shapes:
        topo = frames.grid_frame(bays, stories)
        for B in (1, 16, 256):
            I = torch.full((B, topo.Ne), 5e-4, dtype=torch.float64, device="cuda") * (1 + 0.1 * torch.rand(B, topo.Ne, dtype=torch.float64, device="cuda"))
            rec = {"frame": f"{bays}x{stories}", "n_eq": topo.n_eq, "half_bandwidth": topo.kd, "B": B}
            outs = {}
            for name, val in (("workgroup", 0), ("coop", 2)):
                _cabi.set_option("frame_coop", val)
                topo.__dict__.pop("_ws", None)
                sol = frames.frame_solve(topo, I)
                torch.cuda.synchronize()
                assert int(sol.status.abs().sum()) == 0
                outs[name] = sol.disp.clone()
                g = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    frames.frame_solve(topo, I, out=sol)
                    side.synchronize()
                    with torch.cuda.graph(g, stream=side):
                        for _ in range(20):
                            frames.frame_solve(topo, I, out=sol)
                torch.cuda.current_stream().wait_stream(side)
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best = 1e9
                for _ in range(5):
                    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
                rec[name + "_us"] = round(best, 1)
            _cabi.set_option("frame_coop", 1)
            rec["speedup"] = round(rec["workgroup_us"] / rec["coop_us"], 2)
            rec["max_rel_diff"] = float((outs["workgroup"] - outs["coop"]).abs().max() / outs["workgroup"].abs().max())
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
