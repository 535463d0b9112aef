#!/usr/bin/env python3
"""Which formulation of the PINN's small bf16 products does the BLAS library run fastest? (graph-replayed, per-call us)"""
import torch
dev = "cuda"
def bench(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
bf = torch.bfloat16
for (M, N, K) in ((128, 175, 350), (128, 350, 175), (128, 350, 684), (128, 302, 350)):
    x = torch.randn(M, K, device=dev, dtype=bf); w = torch.randn(N, K, device=dev, dtype=bf); b = torch.randn(N, device=dev, dtype=bf)
    g = torch.randn(M, N, device=dev, dtype=bf)
    wT = w.t().contiguous()
    Np = (N + 63) // 64 * 64
    wp = torch.zeros(Np, K, device=dev, dtype=bf); wp[:N] = w; bp = torch.zeros(Np, device=dev, dtype=bf); bp[:N] = b
    gp = torch.zeros(M, Np, device=dev, dtype=bf); gp[:, :N] = g
    out = {}
    out["fwd addmm(b, x, w.t())"] = bench(lambda: torch.addmm(b, x, w.t()))
    out["fwd mm(x, w.t())"] = bench(lambda: torch.mm(x, w.t()))
    out["fwd mm(x, wT)"] = bench(lambda: torch.mm(x, wT))
    out["fwd (w @ x.t())"] = bench(lambda: torch.mm(w, x.t()))
    out["fwd addmm padded N"] = bench(lambda: torch.addmm(bp, x, wp.t()))
    out["dX mm(g, w)"] = bench(lambda: torch.mm(g, w))
    out["dX (w.t() @ g.t())"] = bench(lambda: torch.mm(w.t(), g.t()))
    out["dX mm(gp, wp)"] = bench(lambda: torch.mm(gp, wp))
    out["dW mm(g.t(), x)"] = bench(lambda: torch.mm(g.t(), x))
    out["dW (x.t() @ g).t()"] = bench(lambda: torch.mm(x.t(), g))
    print((M, N, K), {k: round(v, 1) for k, v in out.items()})
