"""Is the PINN training epoch GPU-bound or host-bound?  The captured step (graph replay) behind one eager batch-assembly launch, as the
training loop issues it: host time to QUEUE 200 steps against the time until the GPU has finished them; and the same with two graph
replays per step (twice the GPU work, the same host calls + one)."""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openpystruct_amd import train
from openpystruct_amd.pinn_fused import PinnFusedStep
from openpystruct_amd.surrogates import CompositeLoss, FNNWithResidual

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = FNNWithResidual(684, 350, 2, 302, 0.5).to(dev)
crit = CompositeLoss(100, 101, 101, 0.5, 0.1, 1e-3, 0.7, 1.5e-6).to(dev)
params = list(model.parameters())
flat = torch.zeros(sum(q.numel() for q in params), device=dev)
off = 0
for q in params:
    q.grad = flat[off:off + q.numel()].view_as(q); off += q.numel()
opt = train.FlatClipAdam(params, flat, 5e-4, weight_decay=1e-3)
eng = PinnFusedStep(model, crit, seed=1)
opt.repack = eng._repack
fold = os.environ.get("OPS_AMD_PINN_NORM_FOLD", "1") == "1"
if fold:
    opt.norm_ready_parts = eng.enable_norm(flat, opt.ws, opt.step_count, opt.betas)
X = torch.randn(6666, 684, device=dev); Y = torch.randn(6666, 302, device=dev)
sig = torch.tensor(0.01, device=dev)
idx = torch.randperm(6666, device=dev)
model.train()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        eng.gather(X, Y, idx[:128], sig, 1); eng.fwd_bwd(128); opt.step()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        eng.fwd_bwd(128); opt.step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for reps in (1, 2):
    for trial in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(200):
            eng.gather(X, Y, idx[128 * (s % 50):128 * (s % 50) + 128], sig, 1)
            for _ in range(reps):
                g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"norm fold {fold}: {reps} replay(s) per step: host queues a step in {1e6 * (t1 - t0) / 200:.1f} us, GPU finishes a step in {1e6 * (t2 - t0) / 200:.1f} us")
