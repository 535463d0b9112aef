#!/usr/bin/env python3
"""The grouped split-row weight-gradient launch of one Transformer-Diffusion step (twelve products, 3584 / 3072 / 512 rows) on its
own: us per launch (HIP events over a captured graph of 50 launches) + correctness against float64."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from openpystruct_amd import _cabi
lib = _cabi.load()
dev = "cuda"
shapes = [(3584, 360, 120), (3584, 120, 120), (3584, 256, 120), (3584, 120, 256)] * 2 + [(3072, 256, 120), (3072, 120, 256), (512, 256, 120), (512, 100, 256)]
g = torch.Generator().manual_seed(1)
ops, arr = [], (_cabi.WgradProblem * len(shapes))()
for e, (T, N, K) in zip(arr, shapes):
    dY = torch.randn(T, N, generator=g).to(torch.bfloat16).to(dev); X = torch.randn(T, K, generator=g).to(torch.bfloat16).to(dev)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.append((dY, X, dW, db))
    e.T, e.N, e.K, e.dY, e.X, e.dW, e.dbias = T, N, K, dY.data_ptr(), X.data_ptr(), dW.data_ptr(), db.data_ptr()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    assert lib.ops_linear_wgrad_accumulate_group(len(shapes), arr, s.cuda_stream) == 0
    s.synchronize()
    worst = 0.0
    for dY, X, dW, db in ops:
        worst = max(worst, float((dW.double() - dY.double().t() @ X.double()).norm() / (dY.double().t() @ X.double()).norm()),
                    float((db.double() - dY.double().sum(0)).norm() / dY.double().sum(0).norm()))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(50):
            lib.ops_linear_wgrad_accumulate_group(len(shapes), arr, s.cuda_stream)
    gr.replay(); s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); gr.replay(); e1.record(s); s.synchronize()
print("grouped weight gradients: %.2f us per launch, worst relative error %.2e" % (e0.elapsed_time(e1) * 1e3 / 50, worst))
