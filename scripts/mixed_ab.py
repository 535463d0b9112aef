#!/usr/bin/env python3
"""The two-wave-size launch against the default tiling: us per launch over batch sizes, one buffer set, graphs of 50 launches behind
60 ms of the launch itself (bench.chip_warm)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import openpystruct_amd as oa
from openpystruct_amd import runtime
runtime.configure()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
ROWS = 0x200
res = {}
for B in ([int(a) for a in sys.argv[1:]] or [2560, 5120, 7680, 10000, 10240, 20000, 50000]):
    st = bench.synth_inputs(B, 0, dev, "trajectory")
    row = {}
    for name, til in (("default", 0), ("mixed", 40 | ROWS), ("default_again", 0), ("mixed_again", 40 | ROWS)):
        out = oa.beam_solve(**st, tiling=til)
        torch.cuda.synchronize()
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            for _ in range(3):
                oa.beam_solve(**st, tiling=til, out=out)
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                for _ in range(50):
                    oa.beam_solve(**st, tiling=til, out=out)
            g.replay(); stream.synchronize()
            bench.chip_warm(g.replay, stream, fn_ms=60.0)
            v = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream); g.replay(); e1.record(stream); stream.synchronize()
                v.append(e0.elapsed_time(e1) / 50 * 1e3)
        row[name] = sorted(v)[2]
        pass
    res[B] = row
    print(B, {k: round(v, 2) for k, v in row.items()}, "mixed/default", round(min(row["mixed"], row["mixed_again"]) / min(row["default"], row["default_again"]), 3), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/mixed_ab.json", "w"), indent=1)
