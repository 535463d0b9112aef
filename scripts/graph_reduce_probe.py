#!/usr/bin/env python3
"""Pure PyTorch + one ctypes call into libamdhip64: why does a captured `g.sum(0)` followed by another op stop updating after its first replay?
PyTorch's multi-block reduction zeroes its semaphores with cudaMemsetAsync at every launch (ATen/native/cuda/Reduce.cuh); if that memset is not
re-executed by a HIP-graph replay, the kernel's "am I the last block of my column" test only works while the semaphore memory still holds the
zeros it started with -- i.e. until a later allocation of the same captured step reuses the block.
  A  graph { t = g.sum(0); o = t.float() }                         (o may take the block the semaphores just gave back)
  B  graph { t = g.sum(0); pad = empty(64 KB); o = t.float() }      (padding allocation in between: o lands elsewhere)
  C  graph { hipMemsetAsync(buf, 0); buf += 1 }                     (is a captured memset replayed at all?)
  D  graph { t = small.sum(0); o = t.float() }                      (single-block reduction: no semaphores)"""
import ctypes
import torch
dev = torch.device("cuda")
hip = ctypes.CDLL("libamdhip64.so")
g = torch.randn(3584, 360, device=dev, dtype=torch.bfloat16)
small = torch.randn(16, 360, device=dev, dtype=torch.bfloat16)


def capture(fn):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            out = fn()
    torch.cuda.current_stream().wait_stream(side)
    return gr, out


def check(name, fn, src):
    gr, out = capture(fn)
    res = []
    for it in range(6):
        src.copy_(torch.randn_like(src, dtype=torch.float32))
        ref = src.float().sum(0)
        gr.replay(); torch.cuda.synchronize()
        o = out[0] if isinstance(out, tuple) else out
        res.append(float((o.float() - ref).abs().max() / ref.abs().max()))
    print(f"{name}: relative error of replays 1..6: " + " ".join("%.3g" % r for r in res), flush=True)


check("A  sum(0) then .float()              ", lambda: g.sum(0).float(), g)
check("B  sum(0), 64 KB allocation, .float()", lambda: (lambda t: (torch.empty(65536, dtype=torch.uint8, device=dev), t.float())[1])(g.sum(0)), g)
check("A' sum(0) alone                      ", lambda: g.sum(0), g)
check("D  single-block sum(0) then .float() ", lambda: small.sum(0).float(), small)

buf = torch.zeros(256, dtype=torch.int32, device=dev)
def memset_then_add():
    rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(1024), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    buf.add_(1)
    return buf
gr, _ = capture(memset_then_add)
vals = []
for it in range(4):
    gr.replay(); torch.cuda.synchronize(); vals.append(int(buf[0]))
print("C  captured hipMemsetAsync + add_(1): buf[0] after replays 1..4 =", vals, "(1 1 1 1: the memset is replayed; 1 2 3 4 or so: it is not)", flush=True)
print("torch", torch.__version__, "hip", torch.version.hip)
