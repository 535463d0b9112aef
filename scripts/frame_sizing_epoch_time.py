#!/usr/bin/env python3
"""Time per epoch of the frame sizing loop (FR:163-206 through frames.optimize_frames) at the reference's own batch -- ONE frame -- and at a few more.

    python scripts/frame_sizing_epoch_time.py [bays x stories ...]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import frames  # noqa: E402


def main():
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(3, 3), (5, 5), (10, 10), (15, 16)]
    for bays, stories in shapes:
        topo = frames.grid_frame(bays, stories)
        for B in (1, 64):
            frames.optimize_frames(topo, B, max_epochs=50)              # warm: library, plan, kernels
            torch.cuda.synchronize()
            n = 400
            t0 = time.perf_counter()
            I, sol, ep = frames.optimize_frames(topo, B, max_epochs=n, poll_every=10 ** 9)     # no early stop inside the timed run
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"frame": f"{bays}x{stories}", "B": B, "epochs": n, "us_per_epoch": round(dt / n * 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
