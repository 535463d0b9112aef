#!/usr/bin/env python3
"""Host enqueue time vs. total time of the PINN training step, graph replay and eager (env: AC=0 disables autocast, CONV=0 drops the conv/BN branch)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import surrogates
torch.manual_seed(0)
dev = torch.device("cuda")
model = surrogates.FNNWithResidual(684, 350, 2, 302, use_conv=os.environ.get("CONV","1")=="1").to(dev)
crit = surrogates.CompositeLoss(100, 101, 101, 0.5, 0.1, torch.tensor(-2.0, device=dev), torch.tensor(2.0, device=dev)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-3, device=dev), capturable=True, fused=True)
X = torch.randn(128, 684, device=dev); Y = torch.randn(128, 302, device=dev)
def step():
    opt.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=os.environ.get("AC","1")=="1"):
        loss = crit(model(X).float(), Y)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3): step()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side): step()
torch.cuda.synchronize()
for name, fn in (("graph", g.replay), ("eager", step)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3*(t1-t0)/100:.3f} ms/step, total {1e3*(t2-t0)/100:.3f} ms/step")
