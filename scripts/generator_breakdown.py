#!/usr/bin/env python3
"""Where the warm `generate_dataset` call spends its wall time (host timers with a device sync at each boundary)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import sizing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
cfg = sizing.SizingConfig()
sizing.generate_dataset(n, cfg, "cuda")          # cold call: library load, allocator, graph pools
torch.cuda.synchronize()
T = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
sizing.make_cases = timed("make_cases", sizing.make_cases)
sizing.optimize_cases = timed("optimize_cases", sizing.optimize_cases)
_init = sizing.SizingState.__init__
sizing.SizingState.__init__ = timed("  SizingState.__init__", _init)
sizing.SizingState.finalize = timed("  finalize", sizing.SizingState.finalize)
_graph_enter = torch.cuda.graph.__enter__; _graph_exit = torch.cuda.graph.__exit__
def ge(self):
    self._t0 = time.perf_counter(); return _graph_enter(self)
def gx(self, *a):
    r = _graph_exit(self, *a); T["  graph capture+instantiate"] = T.get("  graph capture+instantiate", 0.0) + time.perf_counter() - self._t0; return r
torch.cuda.graph.__enter__ = ge; torch.cuda.graph.__exit__ = gx
_replay = torch.cuda.CUDAGraph.replay
cnt = [0]
def rp(self):
    cnt[0] += 1; return _replay(self)
torch.cuda.CUDAGraph.replay = rp
for rep in range(3):
    T.clear(); cnt[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rec = sizing.generate_dataset(n, cfg, "cuda")
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"total {tot*1e3:.2f} ms, replays {cnt[0]}: " + ", ".join(f"{k.strip()} {v*1e3:.2f}" for k, v in T.items()))
