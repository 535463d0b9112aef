#!/usr/bin/env python3
"""Per-launch time of every strip launch of the PINN step (a HIP graph of 40 identical launches each), for the product library and
for stand-alone experiment builds of csrc/mlp_block.hip:

    cd scratch && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-fast-math -ffp-contract=off -DMB_EXP=1 \
        ../openpystruct_amd/csrc/mlp_block.hip -o libmlp_exp1.so          # MB_EXP=1: no product, 2: one reduction step
    python scripts/mlp_launch_bench.py exp1

(the table of profiles/r02_notes.md section 7)."""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from openpystruct_amd import _cabi
from openpystruct_amd.pinn_fused import PinnFusedStep
sys.path.insert(0, 'tests')
import test_gpu_pinn_fused as T
dev = torch.device('cuda:0')
model, crit = T._make(0, 0.5); model, crit = model.to(dev), crit.to(dev); T._attach_flat(model)
eng = PinnFusedStep(model, crit, seed=1)
x = torch.randn(128, 684, device=dev); y = torch.randn(128, 302, device=dev)
model.train(); eng.set_batch(x, y); eng.fwd_bwd(128); torch.cuda.synchronize()
names = ['F0 in(684->350)+bn+act', 'F1 fc1(350->175)+side', 'F2 fc2(175->350)+stencil+bn', 'F1b', 'F2b', 'Fo out(350->302)+loss',
         'Gout dX(302->350)+bnbwd', 'Ga dX(350->175)+side', 'Gb dX(175->350)+stencil+bnbwd', 'Ga1', 'Gb1 (+act bwd)']
launches = eng._fwd + eng._bwd
libs = {'prod': _cabi.load()}
for tag in sys.argv[1:]:
    libs[tag] = ctypes.CDLL(os.path.abspath(f'scratch/libmlp_{tag}.so'))
    libs[tag].ops_mlp_strip_launch.restype = ctypes.c_int
    libs[tag].ops_mlp_strip_launch.argtypes = [ctypes.POINTER(_cabi.MlpStripArgs), ctypes.c_void_p]
    libs[tag].ops_mlp_wgrad_group.restype = ctypes.c_int
    libs[tag].ops_mlp_wgrad_group.argtypes = [ctypes.c_int, ctypes.POINTER(_cabi.MlpWgradProblem), ctypes.c_void_p]
def timeit(fn, n=40):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(side.cuda_stream); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
print(f"{'launch':34s} " + ' '.join(f'{t:>8s}' for t in libs))
for name, a in zip(names, launches):
    a.B = 128
    row = []
    for tag, lib in libs.items():
        row.append(timeit(lambda s, lib=lib, a=a: lib.ops_mlp_strip_launch(ctypes.byref(a), s)))
    print(f'{name:34s} ' + ' '.join(f'{v:8.2f}' for v in row))
row = [timeit(lambda s, lib=lib: lib.ops_mlp_wgrad_group(len(eng._wgrad), eng._wgrad, s)) for lib in libs.values()]
print(f"{'wgrad group':34s} " + ' '.join(f'{v:8.2f}' for v in row))
