#!/usr/bin/env python3
"""Pure PyTorch: which single op of the bf16 step goes wrong under `replay, eager pass, replay`?  Each candidate is captured alone in a HIP
graph (side stream, after eager warm-up), then 40 x (refill inputs, replay, compare with the eager result, run an unrelated eager bf16
training pass of nn.TransformerEncoderLayer on the default stream).  Prints how many replays gave a non-finite / wrong output."""
import torch, torch.nn as nn
dev = torch.device("cuda")
torch.manual_seed(0)
T, N, K = 3584, 360, 120
x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.1
b = torch.randn(N, device=dev, dtype=torch.bfloat16)
g = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
g3 = g.view(512, 7, N).transpose(0, 1)          # non-contiguous view like the attention's packed projection gradient

cands = {
    "sum(0) of bf16 [3584,360]": lambda: g.sum(0),
    "sum((0,1)) of a transposed bf16 view": lambda: g3.sum((0, 1)),
    "addmm(bias, x, w^T) bf16": lambda: torch.addmm(b, x, w.t()),
    "linear(x, w, b) bf16": lambda: torch.nn.functional.linear(x, w, b),
    "g^T @ x bf16": lambda: g.t() @ x,
    "g @ w bf16": lambda: g @ w,
    "sum(0).float()": lambda: g.sum(0).float(),
    "fp32 sum(0)": lambda: g.float().sum(0),
}
layer = nn.TransformerEncoderLayer(120, 8, 256, 0.1, batch_first=True).to(dev)
xe = torch.randn(288, 7, 120, device=dev)


def eager_pass():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        layer(xe).float().square().mean().backward()


for name, fn in cands.items():
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            out = fn()
    torch.cuda.current_stream().wait_stream(side)
    for interleave in (True, False):
        bad_nf = bad_val = 0
        worst = (0.0, None)
        for it in range(40):
            g.copy_(torch.randn(T, N, device=dev)); x.copy_(torch.randn(T, K, device=dev))
            ref = fn().float().clone()
            torch.cuda.synchronize()
            gr.replay(); torch.cuda.synchronize()
            o = out.float()
            if not bool(torch.isfinite(o).all()):
                bad_nf += 1
            elif float((o - ref).abs().max()) > 1e-2 * float(ref.abs().max()) + 1e-3:
                bad_val += 1
                e = float((o - ref).abs().max()) / float(ref.abs().max())
                if e > worst[0]:
                    worst = (e, (it, o.flatten()[:4].tolist(), ref.flatten()[:4].tolist()))
            if interleave:
                eager_pass(); torch.cuda.synchronize()
        print(f"{name:45s} {'with' if interleave else 'no  '} eager pass between replays: non-finite {bad_nf:2d} / 40   wrong {bad_val:2d} / 40   worst {worst}", flush=True)
    del gr
