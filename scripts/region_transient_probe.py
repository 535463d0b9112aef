#!/usr/bin/env python3
"""Why does a 20-launch timed region run ~1.2 us per launch slower than a 500-launch one on the same box?
(bench.py at the driver's flags, 10 000 beams x 100 elements.)  Series of 20-launch replays after different untimed preludes."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import openpystruct_amd as oa
from openpystruct_amd import runtime
runtime.configure()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
B = 10000
st = bench.synth_inputs(B, 0, dev, "trajectory")
out = oa.beam_solve(**st, tiling=0)
torch.cuda.synchronize()
stream = torch.cuda.Stream(device=dev)
graphs = {}
with torch.cuda.stream(stream):
    for _ in range(5):
        oa.beam_solve(**st, tiling=0, out=out)
    stream.synchronize()
    for K in (20, 500):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for i in range(K):
                oa.beam_solve(**st, tiling=0, out=out)
        g.replay(); stream.synchronize()
        graphs[K] = g

def region(g, K):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        e0.record(stream); g.replay(); e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3

res = {}
# 1. the bench's own prelude (40 ms of copy kernel, the graph twice, drain), then the region -- ten times each
for K in (20, 500):
    v = []
    for _ in range(10):
        with torch.cuda.stream(stream):
            bench.chip_warm(graphs[K].replay, stream)
        v.append(region(graphs[K], K))
    res[f"bench_prelude_K{K}"] = v
# 2. prelude + fn_ms of the kernel itself (graph replays) in front of the 20-launch region
for fn_ms in (2.0, 5.0, 10.0, 20.0):
    v = []
    for _ in range(10):
        with torch.cuda.stream(stream):
            bench.chip_warm(graphs[20].replay, stream, fn_ms=fn_ms)
        v.append(region(graphs[20], 20))
    res[f"prelude_plus_{fn_ms:g}ms_of_the_kernel_K20"] = v
# 3. no copy kernel at all: only the kernel itself for 40 ms, then the region
v = []
for _ in range(10):
    with torch.cuda.stream(stream):
        bench.chip_warm(graphs[20].replay, stream, ms=0.0, fn_ms=40.0)
    v.append(region(graphs[20], 20))
res["only_the_kernel_40ms_K20"] = v
# 4. series: after the bench's prelude, 60 replays of the 20-launch graph back to back, an event after each
with torch.cuda.stream(stream):
    bench.chip_warm(graphs[20].replay, stream)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
with torch.cuda.stream(stream):
    ev[0].record(stream)
    for i in range(60):
        graphs[20].replay(); ev[i + 1].record(stream)
torch.cuda.synchronize()
res["series_60_replays_of_K20_us_per_launch"] = [ev[i].elapsed_time(ev[i + 1]) / 20 * 1e3 for i in range(60)]
# 5. idle sensitivity: prelude with 10 ms of the kernel, then sleep, then the region
for sl in (0.0, 0.001, 0.01, 0.1):
    v = []
    for _ in range(6):
        with torch.cuda.stream(stream):
            bench.chip_warm(graphs[20].replay, stream, fn_ms=10.0)
        time.sleep(sl)
        v.append(region(graphs[20], 20))
    res[f"kernel_10ms_then_sleep_{sl:g}s_K20"] = v
for k, v in res.items():
    a = np.array(v)
    print(f"{k:55s} median {np.median(a):7.3f}  min {a.min():7.3f}  max {a.max():7.3f}" if "series" not in k else f"{k}: " + " ".join(f"{x:.2f}" for x in a))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/region_transient_probe.json", "w"), indent=1)
