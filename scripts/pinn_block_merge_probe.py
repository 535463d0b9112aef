"""Times the merged residual-block forward launch (ops_mlp_block_fwd_launch) against the two launches it replaces, in isolation (HIP
events over 200 launches each), with the debug skip mask of the fused kernel (b.loss_C of the fc1 argument block: bit 0 = no products
of phase A, 1 = no tail / LDS writes of h, 2 = no copy-out of h, 3 = no boundary terms) to see where its time goes."""
import ctypes, os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
eng = PinnFusedStep(model, crit, seed=1)
model.train()
eng.set_batch(torch.randn(128, 684, device=dev), torch.randn(128, 302, device=dev))
eng.fwd_bwd(128)
torch.cuda.synchronize()
lib, s = eng.lib, torch.cuda.current_stream().cuda_stream
f1m, f2m = eng._fwd_merged[1][1], eng._fwd_merged[1][2]
f1, f2 = eng._fwd[1], eng._fwd[2]
for a in (f1m, f2m, f1, f2):
    a.B = 128


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def two():
    lib.ops_mlp_strip_launch(ctypes.byref(f1), s); lib.ops_mlp_strip_launch(ctypes.byref(f2), s)


print(f"two launches (eager, back to back): {timeit(two):.1f} us per pair")
for mask in (0, 1, 2, 3, 4, 8, 15):
    f1m.loss_C = mask
    t = timeit(lambda: lib.ops_mlp_block_fwd_launch(ctypes.byref(f1m), ctypes.byref(f2m), s))
    print(f"merged launch, skip mask {mask:2d}: {t:.1f} us")
f1m.loss_C = 0
