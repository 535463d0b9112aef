#!/usr/bin/env python3
"""Pure PyTorch (nothing of openpystruct_amd is imported): is the periodic stall of short graph replays a property of the box?  A graph of
25 elementwise kernels (~2 ms per replay, like the generator's 25-epoch graph at 50 000 cases), replayed N times with a device sync after
each; every replay slower than 10 ms + twice the median is printed with the host clock at its end modulo 100 ms.  Then the same as eager
launches, and as one ~2 ms kernel per iteration.  usage: stall_minimal.py [iterations]"""
import sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda")
x = torch.randn(48 << 20, device=dev)       # 192 MB: one pass ~ 80 us
y = torch.empty_like(x)


def body(k):
    for _ in range(k):
        torch.mul(x, 1.0001, out=y)


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    body(3); side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        body(25)
torch.cuda.synchronize()


def loop(what, fn):
    d, ends = [], []
    for _ in range(n):
        t0 = time.perf_counter_ns(); fn(); torch.cuda.synchronize(); t1 = time.perf_counter_ns()
        d.append((t1 - t0) / 1e6); ends.append(t1)
    med = sorted(d)[len(d) // 2]
    st = [(i, round(t, 1), round((e % 100_000_000) / 1e6, 1)) for i, (t, e) in enumerate(zip(d, ends)) if t > 10 + 2 * med]
    print(f"{what}: {n} iterations, median {med:.3f} ms, total {sum(d):.0f} ms, stalls (index, ms, end mod 100 ms): {st}", flush=True)


loop("graph of 25 kernels", g.replay)
# the generator's poll: the answer of "any case still active?" travels through a pinned flag behind an event (sizing.optimize_cases)
flags = torch.zeros(2, dtype=torch.uint8).pin_memory()
events = [torch.cuda.Event(), torch.cuda.Event()]
act = torch.ones(50000, dtype=torch.uint8, device=dev)
state = {"k": 0}
def replay_and_poll():
    k = state["k"]
    g.replay()
    flags[k & 1: (k & 1) + 1].copy_(act.any().to(torch.uint8).reshape(1), non_blocking=True)
    events[k & 1].record()
    if k >= 1:
        events[(k - 1) & 1].synchronize()
    state["k"] = k + 1
loop("graph + any() + pinned D2H flag + event (the generator's poll)", replay_and_poll)
def replay_and_any():
    g.replay(); act.any()
loop("graph + any() (no copy)", replay_and_any)
def replay_and_copy():
    g.replay(); flags[0:1].copy_(act[0:1], non_blocking=True)
loop("graph + pinned D2H copy of one byte", replay_and_copy)
loop("25 eager launches", lambda: body(25))
big = torch.randn(1 << 30, device=dev // 1 if False else dev)
loop("one long kernel", lambda: torch.mul(big, 1.0001, out=big))
