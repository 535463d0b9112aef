#!/usr/bin/env python3
"""The optimiser step (gradient norm + clip + Adam + bf16 shadow + tiled weight copies) on the PINN's and the TFD's parameter sets:
us per step over a captured graph of 50 steps."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from openpystruct_amd import _cabi, train
ru = lambda v, m: (v + m - 1) // m * m
sets = {"pinn": [(350, 684), (350,), (350,), (350,)] + [(175, 350), (175,), (350, 175), (350,), (1, 1, 3), (1,), (1,), (1,), (350,), (350,)] * 2 + [(302, 350), (302,)],
        "tfd": [(256, 120), (256,), (120, 256), (120,), (1, 1, 120)] + [(360, 120), (360,), (120, 120), (120,), (256, 120), (256,), (120, 256), (120,), (120,), (120,), (120,), (120,)] * 2 + [(256, 120), (256,), (256,), (256,), (100, 256), (100,)]}
for name, shapes in sets.items():
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.1) for s in shapes]
    flat = torch.zeros(sum(p.numel() for p in ps), device="cuda")
    off = 0
    for p in ps:
        p.grad = flat[off:off + p.numel()].view_as(p); off += p.numel()
    opt = train.FlatClipAdam(ps, flat, 1e-3)
    opt.enable_shadow()
    mats = [p for p in ps if p.dim() == 2][:_cabi.MLP_MAX_REPACK]
    tiles = [(torch.zeros(ru(N, 16), ru(K, 32), dtype=torch.bfloat16, device="cuda"), torch.zeros(ru(K, 16), ru(N, 32), dtype=torch.bfloat16, device="cuda")) for N, K in (m.shape for m in mats)]
    ent = (_cabi.MlpRepackEntry * len(mats))()
    for e, w, (wp, wtp) in zip(ent, mats, tiles):
        e.W, e.N, e.K, e.Wp, e.ldw, e.Wtp, e.ldwt = w.data_ptr(), w.shape[0], w.shape[1], wp.data_ptr(), wp.shape[1], wtp.data_ptr(), wtp.shape[1]
    for mode in ("repack", "plain"):
        opt.repack = ent if mode == "repack" else None
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            flat.normal_(); opt.step(); s.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(50):
                    opt.step()
            gr.replay(); s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); gr.replay(); e1.record(s); s.synchronize()
        print(f"{name} ({flat.numel()} parameters, {len(mats)} tiled matrices) {mode}: {e0.elapsed_time(e1) * 1e3 / 50:.2f} us per step (two launches)")
