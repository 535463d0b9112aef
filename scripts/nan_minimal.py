#!/usr/bin/env python3
"""Pure PyTorch (nothing of openpystruct_amd is imported): a bf16-autocast training step of nn.TransformerEncoder captured in a HIP graph,
replays interleaved with an EAGER step of another batch size -- the pattern in which the r03 NaNs appeared (profiles/r04_notes.md).
Several "runs" per process (a new model, optimiser, side stream and graph each, the previous ones destroyed), as train_surrogate does.
Prints, per run, the first replay after which a parameter gradient is non-finite.  usage: nan_minimal.py [runs] [epochs]"""
import gc, sys
import torch, torch.nn as nn
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flags = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else set()
# flags: detach (the step returns loss.detach(): no autograd graph outlives it), same (eager step with the captured batch size), noeager (no eager
# training step), noeval (no eager evaluation pass), fp32 (no autocast), sgd (hand-written update instead of capturable Adam), side (eager
# step issued on the capture stream), nobias (Linear / attention biases frozen), setnone (gradients not pre-allocated: p.grad = None start)
dev = torch.device("cuda")
B, S, D = 512, 7, 120


def one_run(r):
    torch.manual_seed(r)
    layer = nn.TransformerEncoderLayer(D, 8, 256, 0.1, batch_first=True)
    model = nn.Sequential(nn.TransformerEncoder(layer, 2)).to(dev)
    head = nn.Linear(D, 100).to(dev)
    params = list(model.parameters()) + list(head.parameters())
    flat = torch.zeros(sum(p.numel() for p in params), device=dev)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p); off += p.numel()
    opt = torch.optim.Adam(params, lr=3e-3, capturable=True) if "sgd" not in flags else None
    import contextlib
    ac = (lambda: contextlib.nullcontext()) if "fp32" in flags else (lambda: torch.autocast("cuda", dtype=torch.bfloat16))
    X = torch.randn(800, S, D, device=dev); Y = torch.randn(800, 100, device=dev)

    def step(x, y):
        flat.zero_()
        with ac():
            out = head(model(x)[:, 0, :])
            loss = (out.float() - y).square().mean()
        loss.backward()
        if opt is not None:
            opt.step()
        else:
            with torch.no_grad():
                torch._foreach_add_(params, [p.grad for p in params], alpha=-1e-3)
        return loss.detach() if "detach" in flags else loss

    sx, sy = X[:B].clone(), Y[:B].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step(sx, sy)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            sloss = step(sx, sy)
    torch.cuda.current_stream().wait_stream(side)
    bad = None
    names = [n for n, _ in list(model.named_parameters()) + list(head.named_parameters())]
    for ep in range(epochs):
        sx.copy_(X[:B]); sy.copy_(Y[:B])
        g.replay()
        torch.cuda.synchronize()
        nf = [n for n, p in zip(names, params) if not bool(torch.isfinite(p.grad).all()) or not bool(torch.isfinite(p).all())]
        if nf and bad is None:
            bad = (ep, nf[:4])
        if "noeager" not in flags:
            if "side" in flags:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step(X[:B] if "same" in flags else X[B:], Y[:B] if "same" in flags else Y[B:])
                torch.cuda.current_stream().wait_stream(side)
            else:
                step(X[:B] if "same" in flags else X[B:], Y[:B] if "same" in flags else Y[B:])     # eager tail batch: 288 rows
        if "noeval" not in flags:
            with torch.no_grad(), ac():
                model.eval(); head(model(X[:200])[:, 0, :]); model.train()
    del g
    gc.collect(); torch.cuda.synchronize()
    clear = getattr(torch._C, "_cuda_clearCublasWorkspaces", None)
    if clear is not None:
        clear()
    return bad, float(sloss)


print("flags", sorted(flags))
for r in range(runs):
    bad, l = one_run(r)
    print("run", r, "NAN at epoch %d in %s" % bad if bad else "OK ", "last loss %.4f" % l, flush=True)
