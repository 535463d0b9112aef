#!/usr/bin/env python3
"""Pure PyTorch: which framework ops give wrong results from their SECOND HIP-graph replay on (the captured-hipMemsetAsync defect of this
runtime, scripts/graph_reduce_probe.py)?  Every op is captured alone (after eager warm-up), replayed four times with fresh inputs and
compared with its eager result.  Shapes: the ones the surrogate training steps of this repo contain on their framework paths."""
import torch, torch.nn as nn, torch.nn.functional as F
dev = torch.device("cuda")
torch.manual_seed(0)


def run(name, make_inputs, fn, tol=2e-2):
    ins = make_inputs()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(*ins); side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            out = fn(*ins)
    torch.cuda.current_stream().wait_stream(side)
    errs = []
    for it in range(4):
        fresh = make_inputs()
        for a, b in zip(ins, fresh):
            a.data.copy_(b.data)
        ref = fn(*ins)
        ref = [r.float().clone() for r in (ref if isinstance(ref, (tuple, list)) else [ref])]
        torch.cuda.synchronize()
        gr.replay(); torch.cuda.synchronize()
        o = [t.float() for t in (out if isinstance(out, (tuple, list)) else [out])]
        e = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) if bool(torch.isfinite(a).all()) else float("inf") for a, b in zip(o, ref))
        errs.append(e)
    ok = all(e < tol for e in errs)
    print(f"{'ok    ' if ok else 'WRONG '} {name:58s} " + " ".join("%.2g" % e for e in errs), flush=True)


bf = torch.bfloat16
for rows in (16, 128, 256, 448, 512, 2016, 3584):
    run(f"sum(0) bf16 [{rows},360]", lambda rows=rows: [torch.randn(rows, 360, device=dev, dtype=bf)], lambda g: g.sum(0))
run("sum(0) f32 [128,350]", lambda: [torch.randn(128, 350, device=dev)], lambda g: g.sum(0))
run("sum(0) f32 [512,256]", lambda: [torch.randn(512, 256, device=dev)], lambda g: g.sum(0))
run("mean() f64 [512,101]", lambda: [torch.randn(512, 101, device=dev, dtype=torch.float64)], lambda g: (g ** 2).mean())
run("mean() f64 [128,101]", lambda: [torch.randn(128, 101, device=dev, dtype=torch.float64)], lambda g: (g ** 2).mean())
run("mean() f32 [128,302]", lambda: [torch.randn(128, 302, device=dev)], lambda g: g.abs().mean())
run("mean() f32 [512,100]", lambda: [torch.randn(512, 100, device=dev)], lambda g: g.abs().mean())
run("norm() f32 [593914]", lambda: [torch.randn(593914, device=dev)], lambda g: g.norm())
run("sum(dim=1) f32 [512,6,120]", lambda: [torch.randn(512, 6, 120, device=dev)], lambda g: g.sum(dim=1))
run("amax f32 [512,100]", lambda: [torch.randn(512, 100, device=dev)], lambda g: g.amax())
run("any() u8 [50000]", lambda: [(torch.rand(50000, device=dev) > 0.5).to(torch.uint8)], lambda g: g.any().to(torch.uint8))
run("softmax f32 [4096,7,7]", lambda: [torch.randn(4096, 7, 7, device=dev)], lambda g: torch.softmax(g, -1))
run("cumsum f32 [512,100]", lambda: [torch.randn(512, 100, device=dev)], lambda g: g.cumsum(1))
run("index_select f32 [6666,684] -> 128", lambda: [torch.randn(6666, 684, device=dev)], lambda g: g.index_select(0, torch.arange(0, 6400, 50, device=dev)))
bn = nn.BatchNorm1d(350).to(dev)
def bn_step(x):
    x = x.detach().requires_grad_()
    y = bn(x); y.square().mean().backward()
    return y.detach(), x.grad, bn.weight.grad.clone()
run("BatchNorm1d fwd+bwd f32 [128,350] (+ mean())", lambda: [torch.randn(128, 350, device=dev)], bn_step)
ln = nn.LayerNorm(120).to(dev)
def ln_step(x):
    ln.weight.grad = None; ln.bias.grad = None
    x = x.detach().requires_grad_()
    y = ln(x); (y * y).sum(-1).sum().backward()
    return y.detach(), x.grad, ln.weight.grad.clone(), ln.bias.grad.clone()
run("LayerNorm fwd+bwd f32 [3584,120]", lambda: [torch.randn(3584, 120, device=dev)], ln_step)
lin = nn.Linear(120, 360).to(dev)
def lin_step(x):
    lin.weight.grad = None; lin.bias.grad = None
    with torch.autocast("cuda", dtype=bf):
        y = lin(x)
    (y.float() * y.float()).sum().backward()
    return lin.weight.grad.clone(), lin.bias.grad.clone()
run("Linear(120,360) bf16 autocast fwd+bwd, 3584 rows: W.grad, b.grad", lambda: [torch.randn(3584, 120, device=dev)], lin_step, tol=5e-2)
def lin_step128(x):
    return lin_step(x)
run("Linear(120,360) bf16 autocast fwd+bwd, 128 rows", lambda: [torch.randn(128, 120, device=dev)], lin_step128, tol=5e-2)
mha = nn.MultiheadAttention(120, 8, dropout=0.0, batch_first=True).to(dev)
def mha_step(x):
    for p in mha.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=bf):
        y, _ = mha(x, x, x, need_weights=False)
    (y.float() ** 2).sum().backward()
    return mha.in_proj_weight.grad.clone(), mha.in_proj_bias.grad.clone(), mha.out_proj.bias.grad.clone()
run("MultiheadAttention bf16 fwd+bwd [512,7,120]: grads", lambda: [torch.randn(512, 7, 120, device=dev)], mha_step, tol=8e-2)
params = [torch.randn(n, device=dev, requires_grad=True) for n in (239400, 350, 122500, 105700)]
def clip(g0):
    for p in params:
        p.grad = g0[: p.numel()].clone()
    return torch.nn.utils.clip_grad_norm_(params, 1.0, foreach=True)
run("clip_grad_norm_ foreach (4 tensors)", lambda: [torch.randn(239400, device=dev)], clip)
print("torch", torch.__version__, "hip", torch.version.hip)
